// bvh_quality.cpp -- CPU count of what a ray costs in the BVH8 the product builds: node visits and triangle tests per ray, with the kernels'
// visiting rule (octant-ordered depth-first walk over hit masks computed at node-test time, gi_traversal.h), for builder experiments that
// need no GPU.  Results (hits) do not depend on the tree (DESIGN.md "Traversal contract"), only the cost does.
//
//   g++ -O2 -std=c++17 -pthread -Igatling_amd/csrc -Iinclude tools/bvh_quality.cpp gatling_amd/csrc/bvh8.cpp -o tools/build/bvh_quality
//   tools/build/bvh_quality soup 1000000        (C3's generator: centres uniform in [-1,1]^3, vertex offsets N(0, 0.01^2))
//   tools/build/bvh_quality spheres 1024 4      (C4-like: a grid of icospheres)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <vector>

#include "bvh8.h"

using namespace gi;

struct Ray { float o[3], d[3], tMin, tMax; };
struct RayStats { double nodes = 0, tris = 0, rays = 0, hits = 0; };

static inline float expScale(uint8_t e) { uint32_t u = (uint32_t)e << 23; float f; memcpy(&f, &u, 4); return f; }

static bool triTest(const Ray& r, const TriRec& T, float& t, float& u, float& v)
{
  const float* e1 = T.e1; const float* e2 = T.e2;
  float pv[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
  float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
  float inv = 1.0f / det;
  float tv[3] = {r.o[0] - T.v0[0], r.o[1] - T.v0[1], r.o[2] - T.v0[2]};
  u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
  float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
  v = (r.d[0] * qv[0] + r.d[1] * qv[1] + r.d[2] * qv[2]) * inv;
  t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
  return det != 0.0f && u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > r.tMin;
}

// what an ideal visiting order would cost: children sorted by entry distance, culled again against the current tBest when popped
static int traceSorted(const Bvh8& B, const Ray& r, float& tBest, RayStats& c, bool anyHit)
{
  struct Item { uint32_t node; float tn; };
  Item stack[512]; int sp = 0;
  const float id[3] = {1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2]};
  tBest = r.tMax; int best = -1;
  stack[sp++] = {0, r.tMin};
  c.rays++;
  while (sp > 0) {
    Item it = stack[--sp];
    static const bool noCull = getenv("BVHQ_SORTED_NOCULL") != nullptr; if (!noCull && it.tn > tBest) continue;
    const Node8& n = B.nodes[it.node]; c.nodes++;
    float s[3] = {expScale(n.e[0]), expScale(n.e[1]), expScale(n.e[2])};
    Item kids[8]; int nk = 0;
    for (int k = 0; k < 8; k++) {
      if (n.meta[k] == 0) continue;
      float tn = r.tMin, tf = tBest * 1.00001f;
      for (int a = 0; a < 3; a++) {
        float lo = n.p[a] + (float)n.qlo[a][k] * s[a], hi = n.p[a] + (float)n.qhi[a][k] * s[a];
        float t0 = (lo - r.o[a]) * id[a], t1 = (hi - r.o[a]) * id[a];
        if (t0 > t1) { float x = t0; t0 = t1; t1 = x; }
        tn = std::fmax(tn, t0); tf = std::fmin(tf, t1 * 1.00001f);
      }
      if (!(tn <= tf)) continue;
      if (n.imask & (1u << k)) kids[nk++] = {n.childBase + (uint32_t)__builtin_popcount(n.imask & ((1u << k) - 1u)), tn};
      else {
        uint32_t cnt = (uint32_t)__builtin_popcount(n.meta[k] >> 5), off = n.meta[k] & 31u;
        for (uint32_t j = 0; j < cnt; j++) {
          c.tris++;
          float t, u, v;
          if (triTest(r, B.tris[n.triBase + off + j], t, u, v) && t < tBest) { tBest = t; best = (int)(n.triBase + off + j); if (anyHit) { c.hits++; return best; } }
        }
      }
    }
    for (int i = 1; i < nk; i++) { Item x = kids[i]; int j = i - 1; while (j >= 0 && kids[j].tn < x.tn) { kids[j + 1] = kids[j]; j--; } kids[j + 1] = x; } // farthest first
    for (int i = 0; i < nk; i++) stack[sp++] = kids[i];
  }
  if (best >= 0) c.hits++;
  return best;
}

// closest hit; returns the triangle index in tree order or -1
static std::vector<uint32_t>* g_visited = nullptr; // packet mode: the nodes a walk visits, in order
static int trace(const Bvh8& B, const Ray& r, float& tBest, RayStats& c, bool anyHit = false);
static int trace(const Bvh8& B, const Ray& r, float& tBest, RayStats& c, bool anyHit)
{
  static const int cull = getenv("BVHQ_CULL") ? atoi(getenv("BVHQ_CULL")) : 0;
  struct Group { uint32_t childBase; uint32_t mask; uint8_t imask; float tn[8]; float tmin; }; // mask bit (slot ^ octinv): hit internal child in `slot`
  static const bool sorted = getenv("BVHQ_SORTED") != nullptr;
  if (sorted) return traceSorted(B, r, tBest, c, anyHit);
  Group stack[256]; int sp = 0;
  const float id[3] = {1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2]};
  const uint32_t octinv = (r.d[0] >= 0.0f ? 1u : 0u) | (r.d[1] >= 0.0f ? 2u : 0u) | (r.d[2] >= 0.0f ? 4u : 0u);
  tBest = r.tMax; int best = -1;
  uint32_t nodeIdx = 0; bool have = true;
  Group cur{0, 0, 0, {0}, 0};
  c.rays++;
  while (true) {
    if (have) {
      const Node8& n = B.nodes[nodeIdx]; c.nodes++;
      if (g_visited) g_visited->push_back(nodeIdx);
      float s[3] = {expScale(n.e[0]), expScale(n.e[1]), expScale(n.e[2])};
      uint32_t imaskHit = 0; float tnk[8] = {0}; float gmin = 3.0e38f;
      for (int k = 0; k < 8; k++) {
        if (n.meta[k] == 0) continue;
        float tn = r.tMin, tf = tBest * 1.00001f;
        for (int a = 0; a < 3; a++) {
          float lo = n.p[a] + (float)n.qlo[a][k] * s[a], hi = n.p[a] + (float)n.qhi[a][k] * s[a];
          float t0 = (lo - r.o[a]) * id[a], t1 = (hi - r.o[a]) * id[a];
          if (t0 > t1) { float x = t0; t0 = t1; t1 = x; }
          tn = std::fmax(tn, t0); tf = std::fmin(tf, t1 * 1.00001f);
        }
        if (!(tn <= tf)) continue;
        if (n.imask & (1u << k)) { imaskHit |= 1u << ((uint32_t)k ^ octinv); tnk[k] = tn; gmin = std::fmin(gmin, tn); }
        else {
          uint32_t cnt = (uint32_t)__builtin_popcount(n.meta[k] >> 5), off = n.meta[k] & 31u;
          for (uint32_t j = 0; j < cnt; j++) {
            c.tris++;
            float t, u, v;
            if (triTest(r, B.tris[n.triBase + off + j], t, u, v) && t < tBest) { tBest = t; best = (int)(n.triBase + off + j); if (anyHit) { c.hits++; return best; } }
          }
        }
      }
      if (cur.mask) stack[sp++] = cur;
      if (cull == 3 && imaskHit) { // min over the hit children but the one visited first
        uint32_t fb = 31u - (uint32_t)__builtin_clz(imaskHit), fslot = fb ^ octinv; gmin = 3.0e38f;
        for (int k = 0; k < 8; k++) if ((imaskHit >> ((uint32_t)k ^ octinv)) & 1u) if ((uint32_t)k != fslot) gmin = std::fmin(gmin, tnk[k]);
      }
      cur = Group{n.childBase, imaskHit, n.imask, {0}, gmin};
      memcpy(cur.tn, tnk, sizeof(tnk));
      have = false;
    }
    if (!cur.mask) { if (sp == 0) break; cur = stack[--sp]; if (cull >= 2 && cur.tmin > tBest) { cur.mask = 0; continue; } }
    if (cur.mask) {
      uint32_t bit = 31u - (uint32_t)__builtin_clz(cur.mask); cur.mask &= ~(1u << bit);
      uint32_t slot = bit ^ octinv;
      if (cull == 1 && cur.tn[slot] > tBest) continue;
      nodeIdx = cur.childBase + (uint32_t)__builtin_popcount(cur.imask & ((1u << slot) - 1u));
      have = true;
    }
  }
  if (best >= 0) c.hits++;
  return best;
}

static void addTri(std::vector<TriRec>& tris, const float* a, const float* b, const float* cc)
{
  TriRec t; memset(&t, 0, sizeof(t));
  for (int k = 0; k < 3; k++) { t.v0[k] = a[k]; t.e1[k] = b[k] - a[k]; t.e2[k] = cc[k] - a[k]; }
  t.origId = (uint32_t)tris.size(); tris.push_back(t);
}

int main(int argc, char** argv)
{
  const char* kind = argc > 1 ? argv[1] : "soup";
  std::mt19937 rng(1234); std::uniform_real_distribution<float> U(-1.0f, 1.0f); std::normal_distribution<float> N(0.0f, 0.01f);
  std::vector<TriRec> tris;
  float camPos[3] = {0, -4, 0}; float vfov = 40.0f * 3.14159265f / 180.0f;
  if (!strcmp(kind, "soup")) {
    size_t n = argc > 2 ? (size_t)atol(argv[2]) : 1000000;
    for (size_t i = 0; i < n; i++) {
      float c[3] = {U(rng), U(rng), U(rng)}, p[3][3];
      for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) p[v][k] = c[k] + N(rng);
      addTri(tris, p[0], p[1], p[2]);
    }
  } else { // a grid^2 field of UV spheres (C4 / C5-like local density: smooth closed meshes)
    int count = argc > 2 ? atoi(argv[2]) : 1024, sub = argc > 3 ? atoi(argv[3]) : 4;
    int side = (int)std::ceil(std::sqrt((double)count)), seg = 8 << sub, rings = 4 << sub;
    for (int s = 0; s < count; s++) {
      float cx = ((s % side) + 0.5f) / side * 2.0f - 1.0f, cz = ((s / side) + 0.5f) / side * 2.0f - 1.0f, rad = 0.8f / side;
      auto P = [&](int i, int j, float* o) { float th = 3.14159265f * j / rings, ph = 6.2831853f * i / seg; o[0] = cx + rad * std::sin(th) * std::cos(ph); o[1] = rad * std::sin(th) * std::sin(ph); o[2] = cz + rad * std::cos(th); };
      for (int j = 0; j < rings; j++) for (int i = 0; i < seg; i++) {
        float a[3], b[3], c2[3], d[3]; P(i, j, a); P(i + 1, j, b); P(i + 1, j + 1, c2); P(i, j + 1, d);
        if (j > 0) addTri(tris, a, b, c2);
        if (j < rings - 1) addTri(tris, a, c2, d);
      }
    }
  }
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  Bvh8 B; buildBvh8(tris, B);
  double t1 = now();
  size_t slots = 0, leafSlots = 0, leafTris = 0;
  for (const Node8& n : B.nodes) for (int k = 0; k < 8; k++) if (n.meta[k]) { slots++; if (!(n.imask & (1u << k))) { leafSlots++; leafTris += (size_t)__builtin_popcount(n.meta[k] >> 5); } }
  printf("%zu triangles, %zu nodes, depth %u, build %.0f ms, slot fill %.3f, triangles per leaf slot %.2f\n", tris.size(), B.nodes.size(), B.maxDepth, t1 - t0,
         (double)slots / (8.0 * B.nodes.size()), (double)leafTris / (double)leafSlots);
  if (argc > 2 && !strcmp(argv[argc - 1], "packet")) {
    // VERDICT r05 next #5, measured before built: in the pixel-major work order a wave's 64 camera rays are samples of ONE pixel (Gaussian pixel filter, sigma 0.375 px:
    // gi_device_math.h gi_fis_gauss).  A wave-uniform walk would visit the UNION of the nodes its rays visit, one node per step, each step with the lanes whose own
    // walk contains that node.  Per pixel: sum of the rays' visits (what the per-lane walks cost in lane-steps), size of the union (steps of the shared walk), and for
    // every union node how many of the 64 rays visit it.  1920x1080 frame, every 30th pixel in x and y.
    const int W = 1920, H = 1080; float th = std::tan(vfov / 2);
    std::normal_distribution<float> G(0.0f, 0.375f);
    double sumVisits = 0, sumUnion = 0, pixels = 0, hist[65] = {0}, sharedLaneSteps = 0;
    double atLeast48 = 0;
    std::vector<uint32_t> visited; std::vector<std::pair<uint32_t, uint32_t>> all;
    for (int y = 15; y < H; y += 30) for (int x = 15; x < W; x += 30) {
      all.clear();
      RayStats dummy;
      for (int sIdx = 0; sIdx < 64; sIdx++) {
        Ray r; r.tMin = 0; r.tMax = 3.0e38f; memcpy(r.o, camPos, 12);
        float px = x + 0.5f + G(rng), py = y + 0.5f + G(rng);
        float dx = (px / W * 2 - 1) * th * W / H, dz = (py / H * 2 - 1) * th;
        float d[3] = {dx, 1.0f, dz}, l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int k = 0; k < 3; k++) r.d[k] = d[k] / l;
        visited.clear(); g_visited = &visited; float t; trace(B, r, t, dummy); g_visited = nullptr;
        for (uint32_t n : visited) all.push_back({n, (uint32_t)sIdx});
      }
      std::sort(all.begin(), all.end());
      sumVisits += (double)all.size(); pixels++;
      for (size_t i = 0; i < all.size();) { size_t j = i; while (j < all.size() && all[j].first == all[i].first) j++; const size_t lanes = j - i; hist[lanes]++; sumUnion++; sharedLaneSteps += (double)lanes; if (lanes >= 48) atLeast48++; i = j; }
    }
    printf("packet walk of one pixel's 64 camera rays (%0.f pixels): per-lane walks %.1f node visits per pixel (%.2f per ray); the union is %.1f nodes per pixel = steps of a shared walk;\n"
           "  lanes per shared step: mean %.1f of 64; steps with >= 48 lanes: %.1f %%;  per-lane walks at 0.79 lane utilisation need %.1f wave steps per pixel\n",
           pixels, sumVisits / pixels, sumVisits / pixels / 64.0, sumUnion / pixels, sharedLaneSteps / sumUnion, 100.0 * atLeast48 / sumUnion, sumVisits / pixels / (64.0 * 0.79));
    printf("  histogram (lanes: share of the shared walk's steps):");
    for (int lo = 1; lo <= 64; lo += 8) { double a = 0; for (int k = lo; k < lo + 8 && k <= 64; k++) a += hist[k]; printf("  %d-%d: %.1f %%", lo, std::min(lo + 7, 64), 100.0 * a / sumUnion); }
    printf("\n");
    return 0;
  }
  // rays: camera rays, then two diffuse bounces and a shadow ray towards (0,0,1.5) from every hit
  const int W = 480, H = 270; RayStats cam, sec, shd;
  std::uniform_real_distribution<float> U01(0.0f, 1.0f);
  float th = std::tan(vfov / 2);
  for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
    Ray r; r.tMin = 0; r.tMax = 3.0e38f; memcpy(r.o, camPos, 12);
    float dx = ((x + 0.5f) / W * 2 - 1) * th * W / H, dz = ((y + 0.5f) / H * 2 - 1) * th;
    float d[3] = {dx, 1.0f, dz}, l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; k++) r.d[k] = d[k] / l;
    float t; int hit = trace(B, r, t, cam);
    for (int bounce = 0; bounce < 2 && hit >= 0; bounce++) {
      const TriRec& T = B.tris[hit];
      float n[3] = {T.e1[1] * T.e2[2] - T.e1[2] * T.e2[1], T.e1[2] * T.e2[0] - T.e1[0] * T.e2[2], T.e1[0] * T.e2[1] - T.e1[1] * T.e2[0]};
      float nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); if (nl == 0) break;
      float sgn = (n[0] * r.d[0] + n[1] * r.d[1] + n[2] * r.d[2]) > 0 ? -1.0f : 1.0f;
      for (int k = 0; k < 3; k++) n[k] *= sgn / nl;
      Ray s; s.tMin = 0; for (int k = 0; k < 3; k++) s.o[k] = r.o[k] + r.d[k] * t + n[k] * 1e-4f;
      // shadow ray
      { Ray q = s; float L[3] = {0.0f - q.o[0], 0.0f - q.o[1], 1.5f - q.o[2]}; float ll = std::sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
        for (int k = 0; k < 3; k++) q.d[k] = L[k] / ll; q.tMax = ll; float tt;
        if (q.d[0] * n[0] + q.d[1] * n[1] + q.d[2] * n[2] > 0) trace(B, q, tt, shd, true); }
      // uniform direction in the hemisphere of n
      float v[3], vl;
      do { for (int k = 0; k < 3; k++) v[k] = U(rng); vl = v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; } while (vl > 1.0f || vl < 1e-6f);
      vl = std::sqrt(vl); float dn = (v[0] * n[0] + v[1] * n[1] + v[2] * n[2]) / vl; float f = dn < 0 ? -1.0f : 1.0f;
      for (int k = 0; k < 3; k++) s.d[k] = f * v[k] / vl;
      s.tMax = 3.0e38f; r = s; hit = trace(B, r, t, sec);
    }
  }
  auto rep = [](const char* name, const RayStats& c) { printf("  %-9s %8.0f rays  hit %.3f  nodes/ray %6.2f  triangles/ray %6.2f\n", name, c.rays, c.hits / c.rays, c.nodes / c.rays, c.tris / c.rays); };
  rep("camera", cam); rep("secondary", sec); rep("shadow", shd);
  double nr = cam.rays + sec.rays + shd.rays;
  printf("  all: nodes/ray %.2f triangles/ray %.2f  cost(214 n + 110 t) %.0f\n", (cam.nodes + sec.nodes + shd.nodes) / nr, (cam.tris + sec.tris + shd.tris) / nr,
         (214.0 * (cam.nodes + sec.nodes + shd.nodes) + 110.0 * (cam.tris + sec.tris + shd.tris)) / nr);
  return 0;
}
