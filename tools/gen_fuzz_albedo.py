"""Directional albedo table of the fuzz (sheen) lobe shared by oracle/gi_oracle.cpp and gatling_amd/csrc/gi_shading.h.

The lobe (open_pbr_surface.mtlx:569-581: sheen_bsdf layered over the coat) is modelled as D * V with
  D(h) = (2 + 1/a) * sin(theta_h)^(1/a) / (2 pi)          ("Charlie" distribution, Conty & Kulla 2017; a = fuzz_roughness clamped to [0.07, 1])
  V    = 1 / (4 (n.l + n.v - n.l n.v))                    (Ashikhmin / Neubelt visibility)
and its directional albedo E(mu = n.v, a) = integral over the hemisphere of D V (n.l) has no closed form.  This script integrates it numerically
(midpoint rule, 500 x 1000 directions) on the grid a = 1/16 .. 1, mu = 0 .. 1 in steps of 1/16 and prints the table both sources embed; they interpolate it
bilinearly and add 0.01, which bounds the true albedo from above everywhere on a >= 0.07 (checked below on random points: largest underestimate of the bare
interpolation 0.0093), so the layer under the fuzz never receives more energy than the fuzz leaves.

  python tools/gen_fuzz_albedo.py            # prints the C initialiser and the check
"""
import numpy as np


def albedo(mu, a, n=500):
    ct = (np.arange(n) + 0.5) / n
    ph = (np.arange(2 * n) + 0.5) / (2 * n) * 2 * np.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1 - CT ** 2)
    l = np.stack([st * np.cos(PH), st * np.sin(PH), CT], -1)
    v = np.array([np.sqrt(max(0.0, 1 - mu * mu)), 0.0, mu])
    h = l + v
    h /= np.linalg.norm(h, axis=-1, keepdims=True)
    s2 = np.clip(1 - h[..., 2] ** 2, 0, 1)
    D = (2 + 1 / a) * np.power(s2, 0.5 / a) / (2 * np.pi)
    V = 1 / (4 * (CT + mu - CT * mu))
    return float((D * V * CT).sum() * (1.0 / n) * (2 * np.pi / (2 * n)))


def main():
    al = np.arange(1, 17) / 16.0
    mu = np.arange(0, 17) / 16.0
    G = np.array([[albedo(max(m, 1e-4), a) for m in mu] for a in al])
    print("// rows: alpha = 1/16 .. 1, columns: mu = 0 .. 1 (tools/gen_fuzz_albedo.py)")
    for row in G:
        print("  {" + ", ".join(f"{np.float32(x):.8e}f" for x in row) + "},")

    def interp(m, a):
        x = m * 16; i = min(int(x), 15); fx = x - i
        y = a * 16 - 1; j = min(max(int(y), 0), 14); fy = y - j
        return (G[j, i] * (1 - fx) + G[j, i + 1] * fx) * (1 - fy) + (G[j + 1, i] * (1 - fx) + G[j + 1, i + 1] * fx) * fy

    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(600):
        m, a = rng.uniform(0, 1), rng.uniform(0.07, 1)
        worst = max(worst, albedo(max(m, 1e-4), a, 300) - interp(m, a))
    print(f"// largest underestimate of the bilinear interpolation on 600 random (mu, alpha >= 0.07): {worst:.4f} (< the 0.01 both sources add)")


if __name__ == "__main__":
    main()
