#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native gi render loop.

Metric (BASELINE.json): Msamples/s = width*height*spp / t_render / 1e6 at 8 bounces, 1920x1080.
One "step" = one giCRender pass of the hot path over the whole workload (BASELINE config C2 by default:
cornell, 1920x1080, spp 1024, max-bounces 8, UsdPreviewSurface model); scene ingest, BVH build and upload happen
before the timed region (the scene is resident in HBM), the timed region contains every bounce stage, the
per-pixel accumulation, for N>1 the RCCL tile gather, and the final D2H of the colour AOV.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|c4|c5] [--spp S]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N>1 is strong scaling: the rows of the same frame are dealt round-robin to the N ranks (rank r renders rows r, r+N, ...:
every share samples the whole image; one rank per GPU, scene replicated) and gathered to rank 0 with
one RCCL gather.  Rank 0 prints ONE JSON line.

roofline (what the line means): `frac` = HBM bytes the dominant kernel really moved (rocprofv3 PMC FETCH_SIZE / WRITE_SIZE, collected by this same run in
separate `--pmc` passes of this same workload, calibrated with tools/pmc_calib.hip) / its mean launch time (HIP events on the library's stream, live) / 8 TB/s.
`algorithmic_frac` is the SURVEY 8d canonical-bytes figure (it counts node / triangle bytes whether they come from HBM, L2 or LDS, so it can exceed 1).  The VALU
side is reported in wall-clock terms against ceilings MEASURED by tools/valu_calib.hip (profiles/valu_issue_calibration.json): `valu_instr_per_simd_per_ns`,
`valu_frac` = that / the dual-issue ceiling (alternating instruction classes, ~1.04), `valu_frac_fp32_only` = that / the single-class ceiling (~0.59).  `bound` names
the nearer of the two rooflines as THIS run measured them; `bound_note` is set when neither is near; `prior_experiment_note` names what earlier calibration runs found for the
kernel family (the big-scene traversal followed its instruction count one to one, profiles/r04k_valu_sensitivity.txt) -- a citation, not a result of the run.

At N = 1 the default line carries, under `also`, compact objects for configs C3 and C4 -- the wavefront pipeline (k_raygen / k_trace_dyn / k_route / k_shade /
k_trace_dyn<any>) -- and for `c5share`, one rank's share of C5's 8-GPU partition at full spp (`projected_8gpu` = 8 x its rate), each with its own roofline fractions;
at N >= 8 config C5 itself, tiled across the ranks.  Raw counters and every kernel's time share / VALU rate / lanes / L2 hit go to profiles/bench_last_pmc.json
(and gpurun_out/ when it exists), not into the line.  The big-scene legs are timed with the per-launch HIP events OFF (hundreds of stage launches per step: the event pairs
cost 1 - 2 % of a C3 / C4 step) and take their stage breakdown from one more, untimed step with the events on (`events_in_timed_region`: false); the main line keeps its
events inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def make_workload(name, spp_override=None):
    from gatling_amd.scene import MAT_DIFFUSE, RenderSettings
    from gatling_amd.scenes import cornell_box, interior_scene, random_triangle_soup, sphere_grid
    if name == "c2":
        desc, rs, w, h = cornell_box(), RenderSettings(spp=1024, max_bounces=8), 1920, 1080
        label = "C2: cornell.usda 1920x1080 spp=1024 max-bounces=8 UsdPreviewSurface (diffuse + GGX specular), NEE off"
    elif name == "c1":
        desc, rs, w, h = cornell_box(MAT_DIFFUSE), RenderSettings(spp=64, max_bounces=4), 512, 512
        label = "C1: cornell.usda 512x512 spp=64 max-bounces=4 diffuse-only"
    elif name == "c3":
        desc, rs, w, h = random_triangle_soup(1_000_000), RenderSettings(spp=256, max_bounces=8, next_event_estimation=True), 1920, 1080
        label = "C3: 1M-triangle soup, 1 OpenPBR material, rect light, NEE on, 1920x1080 spp=256 max-bounces=8"
    elif name == "c4":
        desc, rs, w, h = sphere_grid(32, 4, 32), RenderSettings(spp=256, max_bounces=8), 1920, 1080
        label = "C4: 32x32 instanced icospheres (5120 tris each), 32 OpenPBR/UsdPreviewSurface materials, 1920x1080 spp=256 max-bounces=8"
    elif name == "c5":
        desc, rs, w, h = interior_scene(), RenderSettings(spp=1024, max_bounces=8, next_event_estimation=True), 3840, 2160
        label = "C5: interior, 10.24M instanced triangles, 51 materials, 4 rect lights, NEE on, 3840x2160 spp=1024 max-bounces=8"
    elif name == "c5share":  # rank 3's share of C5's 8-GPU partition (rows 3, 11, 19, ...): the per-GPU work of the configuration C5 is specified on, on ONE GPU
        desc, rs, w, h = interior_scene(), RenderSettings(spp=1024, max_bounces=8, next_event_estimation=True), 3840, 2160
        label = "C5 share 3/8: interior, 10.24M instanced triangles, NEE on, 3840x2160 spp=1024 max-bounces=8, rows 3::8 (270 of 2160)"
    else:
        raise SystemExit(f"unknown workload {name}")
    if spp_override:
        rs.spp = spp_override
        label += f" [spp overridden to {spp_override}]"
    return desc, rs, w, h, label


def cpu_baseline(desc, rs, w, h, budget_s=7.0):
    """The CPU oracle (kind "port") on this box's host cores, on a bounded sample of the same workload: the full
    frame at a reduced spp chosen so the run takes ~budget_s (throughput is per sample, so it scales linearly)."""
    import copy
    from oracle import orc
    cores = os.cpu_count() or 1
    probe = copy.copy(rs); probe.spp = 2
    orc.render(desc, probe, max(8, w // 8), max(8, h // 8), threads=cores)  # warm the library / thread pool
    t0 = time.perf_counter(); orc.render(desc, probe, w, h, threads=cores); t1 = (time.perf_counter() - t0) / probe.spp
    spp = int(max(1, min(rs.spp, budget_s / max(t1, 1e-4))))
    run = copy.copy(rs); run.spp = spp
    t0 = time.perf_counter(); _, cnt = orc.render(desc, run, w, h, threads=cores); dt = time.perf_counter() - t0
    return {"value": round(w * h * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"full {w}x{h} frame at spp={spp} of {rs.spp} (same seeds as the first {spp} samples), {dt:.1f} s wall, "
                      f"{cnt['segments'] / cnt['samples']:.3f} segments/sample"}


VALU_CYCLES_PER_INST_DEFAULT = 4.0  # used only when profiles/valu_issue_calibration.json is absent: a wave64 VALU instruction occupies a SIMD's 16 fp32
                                    # lanes for 4 cycles (157.3 TFLOP/s = 1024 SIMDs x 16 lanes x 2 (fma) x 2 (packed) x 2.4 GHz)
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md)
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"),
              ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_THREAD_CYCLES_VALU"))


def fetch_calibration():
    """FETCH_SIZE / WRITE_SIZE scale factors (reported KiB -> bytes), calibrated with tools/pmc_calib on known byte counts
    (profiles/pmc_calibration.json; MI355X_MICROARCH.md 'HBM': wide coalesced reads report exactly half)."""
    cal = {"fetch": 2.0, "fetch_scattered": 1.0, "write": 1.0, "source": "MI355X_MICROARCH.md (wide coalesced reads x2), write uncalibrated"}
    p = os.path.join(ROOT, "profiles", "pmc_calibration.json")
    if os.path.exists(p):
        try:
            j = json.load(open(p))
            cal.update({k: j[k] for k in ("fetch", "fetch_scattered", "write", "source") if k in j})
        except Exception:
            pass
    return cal


def valu_calibration():
    """VALU issue ceilings MEASURED by tools/valu_calib.hip (profiles/valu_issue_calibration.json, written by tools/valu_calib_report.py): wave64 instructions
    per SIMD and NANOSECOND at 8 waves/SIMD, chip-wide -- in wall time, because the shader clock sags under a full-chip VALU load (1.4-1.5 GHz measured, not
    the 2.4 GHz maximum), so `cycles x 2.4 GHz` is not a rate.  "fp32": a stream of one fp32 class (v_fma_f32 / v_pk_fma_f32 / v_cvt / v_max3: ~0.53-0.59);
    "mixed": alternating classes dual-issue (cvt + fma 1:1, int adds: ~1.04).  A real kernel's ceiling lies between the two."""
    p = os.path.join(ROOT, "profiles", "valu_issue_calibration.json")
    out = {"fp32": CLOCK_HZ / VALU_CYCLES_PER_INST_DEFAULT / 1e9, "mixed": CLOCK_HZ / VALU_CYCLES_PER_INST_DEFAULT / 1e9, "source": "default (no calibration file): 4 cycles at 2.4 GHz"}
    try:
        j = json.load(open(p))
        rate = {r["op"]: float(r["wall_instr_per_simd_per_ns"]) for r in j["rows"] if int(r["waves_per_simd"]) == 8}
        fp, mixed = rate["v_fma_f32"], max(rate.values())
        if 0.05 < fp <= mixed < 4.0:
            out = {"fp32": fp, "mixed": mixed, "source": j.get("source", "profiles/valu_issue_calibration.json")}
    except Exception:
        pass
    return out


def short_kernel(name):
    return name.replace("gi::", "").replace("void ", "").split("(")[0]


def pmc_live(workload, spp, timeout_s=240):
    """Runs this workload once more per counter group under `rocprofv3 --pmc` (no tracing options: counters only) and returns
    {kernel: {counter: sum, "dispatches": n, "pmc_us": mean duration under counter collection}} or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="gatling_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    notes = []
    try:
        for i, group in enumerate(PMC_PASSES):
            d = os.path.join(tmp, f"p{i}")
            cmd = [exe, "--pmc", *group, "-d", d, "-o", "probe", "--", sys.executable, os.path.abspath(__file__), "--probe", "--workload", workload]
            if spp:
                cmd += ["--spp", str(spp)]
            try:
                r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                notes.append(f"pass {group[0]}: timeout"); continue
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                notes.append(f"pass {group[0]}: rc {r.returncode}, {len(dbs)} db"); continue
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value), sum(duration) * 1.0 / count(*) "
                               "from counters_collection group by kernel_name, counter_name").fetchall()
            for k, c, n, v, dur in rows:
                e = out.setdefault(short_kernel(k), {})
                e[c] = float(v); e["dispatches"] = int(n); e["pmc_us"] = float(dur) / 1e3
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not out:
        return None, "; ".join(notes) or "no counters collected"
    return out, "; ".join(notes)


def reference_probe():
    """the real reference next to the GPU path, when a build of it and a Vulkan ray-tracing device exist on
    this box.  It is timed the way src/gatling/main.cpp:197-207 times itself ("Rendering finished (%.3fs)"), second run."""
    import shutil
    import subprocess
    g, v = shutil.which("gatling"), shutil.which("vulkaninfo")
    if not g or not v:
        return {"status": "absent", "gatling": bool(g), "vulkaninfo": bool(v),
                "note": "no reference build / Vulkan RT device on this box (BASELINE.md section 2); cpu_baseline.kind stays 'port'"}
    scene = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gatling_bench_cornell.usda")
    try:  # the C2 scene as .usda (same generator as the GPU run)
        from gatling_amd.scenes import cornell_box
        from gatling_amd.usda_writer import write_usda
        write_usda(scene, cornell_box())
    except Exception as e:  # noqa: BLE001
        return {"status": f"could not write the scene file: {e!r}"[:200], "gatling": True, "vulkaninfo": True}
    times = []
    for _ in range(2):  # first run pays MDL -> GLSL -> SPIR-V compilation and the BLAS / TLAS build
        try:
            r = subprocess.run([g, scene, "/tmp/gatling_ref.png", "--image-width", "1920", "--image-height", "1080", "--spp", "16", "--max-bounces", "8"],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
        except Exception as e:  # noqa: BLE001
            return {"status": f"failed: {e}"}
        import re as _re
        m = _re.search(r"Rendering finished \(([0-9.]+)s\)", r.stdout or "")
        if r.returncode != 0 or not m:
            return {"status": f"failed: rc {r.returncode}"}
        times.append(float(m.group(1)))
    out = {"status": "timed", "value": round(1920 * 1080 * 16 / times[1] / 1e6, 4), "unit": "Msamples/s", "sample": "cornell 1920x1080 spp=16, second run", "cores": os.cpu_count()}
    out["image_comparison"] = reference_image_comparison("/tmp/gatling_ref.png", 16)
    return out


def reference_image_comparison(ref_png, spp):
    """SURVEY 8d(ii) on the image the reference just wrote: our C2 frame at the same spp, twice with disjoint sample offsets (the second through progressive
    accumulation: B = 2 x accumulated - A), against the reference's sRGB8 file -- RMSE vs 2 x our Monte-Carlo standard error, mean luminance within 0.5 %, and the
    reference's own differing-byte count (tools/compare_reference.py)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import compare_reference as cr
        from gatling_amd import capi
        from gatling_amd.scene import RenderSettings
        from gatling_amd.scenes import cornell_box
        ref, is8 = cr.load_image(ref_png)
        scene = capi.Scene(cornell_box())
        rs = RenderSettings(spp=spp, max_bounces=8)  # progressive accumulation on: the second call renders samples [spp, 2 spp) and blends
        a = scene.render(rs, 1920, 1080).copy()
        acc = scene.render(rs, 1920, 1080).copy()
        scene.close()
        r = cr.compare(a, 2.0 * acc - a, ref, is8)
        return {k: r[k] for k in ("pass", "rmse", "standard_error", "rmse_over_standard_error", "mean_luminance_rel_error", "srgb8_differing_bytes", "srgb8_bytes", "tolerance")}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timers", action="store_true", help="do not record per-stage HIP events (roofline fields become 0)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes (roofline.traffic / frac become null)")
    ap.add_argument("--in-process", action="store_true", help="ONE process driving --gpus N devices inside the library (giCInitializeDevices: rows dealt to the devices per "
                    "giCRender, shares copied into place on device 0) instead of one rank per GPU + RCCL gather; for comparing the two multi-GPU forms")
    ap.add_argument("--probe", action="store_true", help=argparse.SUPPRESS)  # internal: one untimed step, no output (the --pmc passes run this)
    args = ap.parse_args()
    if args.probe:
        args.steps, args.warmup, args.no_timers, args.no_cpu_baseline, args.no_pmc = 1, 0, True, True, True

    import torch  # plumbing: device sync, torch.distributed (RCCL).  Imported first so one HIP runtime is shared.
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and not args.in_process:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU), or pass --in-process")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    if os.environ.get("GATLING_BENCH_SHARE_GPU"):  # tests: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or bool(os.environ.get("GATLING_BENCH_FORCE_DIST"))  # the env var drives the N>1 code path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from gatling_amd import capi
    from gatling_amd.dist import RowGather, interleaved_rows
    if args.in_process and world == 1 and args.gpus > 1:
        capi.initialize(devices=list(range(args.gpus)))  # before the first Scene: one giCInitialize per process

    def timed_run(workload, spp, steps, warmup, no_timers, main_leg=True):
        """Scene resident in HBM, `warmup` untimed steps, then exactly `steps` steps between barrier + synchronize; time = max over ranks."""
        desc, rs, w, h, label = make_workload(workload, spp or None)
        rs.progressive_accumulation = False  # every step renders the same frame from sample 0 (no cross-step accumulation)
        scene = capi.Scene(desc, device=local_rank)
        r0, r1, rstride = interleaved_rows(h, world, rank)  # rows rank::world: every rank's share costs the same (dist.py)
        if workload == "c5share" and world == 1:
            r0, r1, rstride = interleaved_rows(h, 8, 3)
        nrows = len(range(r0, r1, rstride))
        dev_ptr = scene.device_pointer(w, h)

        class _Tile:  # zero-copy view of the library's device render buffer (rows r0..r1) for the RCCL gather
            def __init__(self):
                self.__cuda_array_interface__ = {"shape": (nrows, w, 4), "typestr": "<f4", "data": (dev_ptr + r0 * w * 16, False), "version": 2,
                                                 "strides": (rstride * w * 16, 16, 4)}
        tile = torch.as_tensor(_Tile(), device=f"cuda:{local_rank}") if use_dist else None
        host_full = torch.empty((h, w, 4), dtype=torch.float32).pin_memory() if (use_dist and rank == 0) else None
        gather = RowGather(h, w, torch.float32, torch.device("cuda", local_rank), interleaved=True) if use_dist else None  # buffers allocated once, outside the timed region

        last = {}
        # rank 0 hands the assembled frame to pinned host memory on a copy stream: the D2H of frame i (33 MB at 1080p, 133 MB at 4K) runs on the
        # SDMA engines while the ranks render frame i + 1; the next gather waits for it before it overwrites the device frame, and the closing
        # synchronize() of the timed region covers the last one
        copy_stream = torch.cuda.Stream(device=local_rank) if (use_dist and rank == 0) else None
        copied = torch.cuda.Event() if copy_stream is not None else None

        def step():
            if not use_dist:
                # blocks; the colour AOV is complete in the library's host memory on return (reference semantics: hdGatling reads giGetRenderBufferMem in place,
                # renderBuffer.cpp:53-154) -- a view of it, not a second host copy through numpy (33 MB, ~4 ms per C2 step until r03)
                last["img"] = scene.render(rs, w, h, copy=False) if rstride == 1 else scene.render(rs, w, h, rows=(r0, r1), row_stride=rstride, copy=False)
            else:
                gather.wait_packed()  # frame i - 1's pack copy has read the render buffer (it runs on torch's stream, the library renders on its own)
                scene.render(rs, w, h, rows=(r0, r1), device_only=True, row_stride=rstride)
                if copy_stream is not None and "img" in last:
                    torch.cuda.current_stream().wait_event(copied)  # frame i - 1 has left the device frame buffer
                full = gather(tile)
                if rank == 0:
                    copy_stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(copy_stream):
                        host_full.copy_(full, non_blocking=True)
                        copied.record(copy_stream)
                    last["img"] = host_full

        def sync():
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
                torch.cuda.synchronize()

        # HIP events around the stage launches, on the library's own stream: every 8th iteration for the LDS-resident scenes (~1 000 iterations of ~0.2 ms launches when the
        # stage kernels run them; events on every launch cost ~16 % there), every iteration for the big scenes (~30-130 iterations of ms-long launches whose cost varies
        # tenfold between the first and the last -- sampling every 8th mis-scaled the stage totals by 8-12 %, the "unowned" time of VERDICT r02 weak #4)
        timer_stride = 0 if no_timers else (8 if workload in ("c1", "c2") else 1)
        # The further legs of the big scenes (hundreds of stage launches per step, an event pair around each: 1 - 2 % of a C3 / C4 step) are TIMED with the events off
        # and get their stage breakdown from one more, untimed step with them on; the main line keeps its events inside the timed region (one pair per fused launch).
        two_phase = (not main_leg) and timer_stride == 1
        scene.set_option(capi.OPTION_KERNEL_TIMERS, 0 if two_phase else timer_stride)
        # ingest / BVH build / upload are "reported separately": the first untimed step is the one that builds and uploads the scene (giCRender syncs the
        # geometry lazily, like giRender's dirty handling) -- its wall time and the library's own build / upload clocks go into the line as first_frame_ms / build_ms / upload_ms
        first = {"build_ms": None, "upload_ms": None, "first_frame_ms": None}
        for i in range(warmup):
            t_f = time.perf_counter()
            step()
            if i == 0:
                torch.cuda.synchronize()
                s0 = scene.stats()
                first = {"build_ms": round(s0["bvhBuildMs"], 2), "upload_ms": round(s0["uploadMs"], 2), "first_frame_ms": round((time.perf_counter() - t_f) * 1e3, 2)}
        if not main_leg:  # further legs start on a GPU that idled through the CPU baseline and the counter passes: a 2 ms step (C1) needs more than one to bring the clocks back
            t_w = time.perf_counter()  # up, and a big scene's second and third frames are still 0.2 - 1 % slower than its tenth (profiles/r05fh_clock_power_under_load.txt)
            while time.perf_counter() - t_w < 1.0:
                step()
        sync()
        t0 = time.perf_counter()
        stats = []
        for _ in range(steps):
            step()
            stats.append(scene.stats())
        sync()
        dt = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())

        timer_stats = stats
        if two_phase:
            scene.set_option(capi.OPTION_KERNEL_TIMERS, timer_stride)
            step()
            sync()
            timer_stats = [scene.stats()]

        return {"desc": desc, "rs": rs, "w": w, "h": h, "label": label, "scene": scene, "rows": (r0, r1, rstride), "dt": dt, "stats": stats, "timer_stats": timer_stats, "last": last, "first": first}

    def measure(workload, spp, steps, warmup, no_timers, no_pmc, main_line):
        """One workload end to end: timed steps, one counting step, the roofline object (live --pmc passes at N = 1).  Every rank takes part;
        rank 0 gets the JSON object, the others None."""
        R = timed_run(workload, spp, steps, warmup, no_timers, main_line)
        desc, rs, w, h, label, scene, (r0, r1, rstride), dt, stats, last = (R[k] for k in ("desc", "rs", "w", "h", "label", "scene", "rows", "dt", "stats", "last"))
        tstats = R["timer_stats"]  # the steps the HIP events were recorded in (the timed steps themselves, or one untimed step after them: timed_run)
        if args.probe:
            scene.close()
            return None, R
        # --- roofline inputs: one extra (untimed) step with the traversal counters on
        scene.set_option(capi.OPTION_COUNT_TRAVERSAL, 1)
        scene.render(rs, w, h, rows=(r0, r1), device_only=True, row_stride=rstride)
        cst = scene.stats()
        scene.set_option(capi.OPTION_COUNT_TRAVERSAL, 0)
        out = None
        if rank == 0:
            samples_per_step = (w * h if (world > 1 or rstride == 1) else len(range(r0, r1, rstride)) * w) * rs.spp  # (a share leg on one GPU counts the share's samples)
            value = samples_per_step * steps / dt / 1e6
            # dominant traversal kernel k_trace<closest>: algorithmic bytes per launch: ray 32 + hit 20 per ray,
            # 80 B per BVH8 node visited, 48 B per triangle tested; divided by its mean launch time (HIP events).
            launches = sum(s["traceLaunches"] for s in tstats)  # traceMs is the sampled total scaled to all launches
            trace_ms = sum(s["traceMs"] for s in tstats)
            rays = sum(s["segments"] for s in tstats)
            nodes_per_ray = cst["nodesVisited"] / max(1, cst["segments"])
            tris_per_ray = cst["trisTested"] / max(1, cst["segments"])
            bytes_total = rays * (52.0 + 80.0 * nodes_per_ray + 48.0 * tris_per_ray)
            achieved = bytes_total / max(trace_ms * 1e-3, 1e-12) / 1e9
            seg_per_sample = rays / max(1, sum(s["samples"] for s in tstats))
            stream_only = (samples_per_step * steps / dt) * seg_per_sample * 192.0 / 1e9  # whole-pipeline stream floor
            if stats[-1]["fusedPath"]:
                kernel, prefixes = "k_path / k_path_bw (fused persistent path kernel: raygen + closest hit + shade per path, wave-local wavefront when NEE is off)", ("k_path",)
            elif cst["triangleCount"] <= 128:
                kernel, prefixes = "k_trace<closest>", ("k_trace<false",)
            else:
                kernel, prefixes = "k_trace_dyn<closest> + k_route", ("k_trace_dyn<false", "k_route")
            avg_launch_s = trace_ms * 1e-3 / max(1, launches)
            stage = {k: round(sum(s[k] for s in tstats) / len(tstats), 3) for k in ("raygenMs", "traceMs", "shadeMs", "shadowMs")}
            # what the four stage timers do not own: k_init / k_accumulate / the queue-size polls / the D2H of the colour AOV (render_ms is the library's own
            # wall clock around the bounce loop + D2H; ms_per_step adds the host side of capi.Scene.render)
            render_ms = sum(s["renderMs"] for s in tstats) / len(tstats)
            stage["renderMs"] = round(render_ms, 3)
            stage["otherMs"] = round(render_ms - sum(stage[k] for k in ("raygenMs", "traceMs", "shadeMs", "shadowMs")), 3)
            raw = {}
            roofline = {"bound": "hbm", "kernel": kernel, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "algorithmic_GBps": round(achieved, 2), "algorithmic_frac": round(achieved / HBM_PEAK_GBS, 5),
                        "bytes_per_launch": round(bytes_total / max(1, launches), 1), "avg_launch_us": round(avg_launch_s * 1e6, 3),
                        "nodes_per_ray": round(nodes_per_ray, 3), "tris_per_ray": round(tris_per_ray, 3),
                        "stage_ms_per_step": stage,
                        "pipeline_stream_only_GBps": round(stream_only, 2), "pipeline_stream_only_frac": round(stream_only / HBM_PEAK_GBS, 5)}
            if world == 1 and not no_pmc:
                scene.close(); scene = None  # the probe processes need the device memory (C5: 17 GB of queues per process)
                pmc, note = pmc_live(workload, spp)
                cal = fetch_calibration()
                vcal = valu_calibration()
                roofline["pmc_note"] = note or "ok"
                if pmc:
                    dom = [v for k, v in pmc.items() if k.startswith(prefixes)]
                    # the closest-hit traversal launches: of the kernels that match, every one belongs to the traversal stage (k_route runs once per k_trace_dyn)
                    n = max([v.get("dispatches", 0) for k, v in pmc.items() if k.startswith(prefixes[0])] or [0])
                    if n:
                        # k_trace_dyn's reads are per-lane 80-B node / 48-B triangle fetches = 64-B sector requests, which FETCH_SIZE reports at their
                        # size; the coalesced x2 applies to streaming kernels only (profiles/pmc_calibration.json).  `traffic_upper` = everything x2.
                        scattered = prefixes[0].startswith("k_trace_dyn")
                        raw_fetch = sum(v.get("FETCH_SIZE", 0.0) for v in dom) * 1024.0 / n
                        fetch = raw_fetch * (cal["fetch_scattered"] if scattered else cal["fetch"])
                        write = sum(v.get("WRITE_SIZE", 0.0) for v in dom) * 1024.0 * cal["write"] / n
                        have = any("FETCH_SIZE" in v for v in dom) and any("WRITE_SIZE" in v for v in dom)
                        if have:
                            roofline["traffic"] = round(fetch + write, 1)
                            roofline["traffic_fetch"] = round(fetch, 1); roofline["traffic_write"] = round(write, 1)
                            roofline["traffic_upper"] = round(raw_fetch * cal["fetch"] + write, 1)
                            roofline["achieved"] = round((fetch + write) / max(avg_launch_s, 1e-12) / 1e9, 2)
                            roofline["frac"] = round(roofline["achieved"] / HBM_PEAK_GBS, 5)
                            roofline["calibration"] = cal
                        main_k = [v for k, v in pmc.items() if k.startswith(prefixes[0])]
                        valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in main_k) / n
                        if valu:
                            # VALU issue rate achieved (wave64 instructions per SIMD and ns over the launch) against the two measured ceilings
                            rate = valu / (SIMDS * avg_launch_s * 1e9)
                            roofline["valu_instr_per_simd_per_ns"] = round(rate, 5)
                            roofline["valu_frac"] = round(rate / vcal["mixed"], 5)        # vs the dual-issue ceiling (alternating instruction classes)
                            roofline["valu_frac_fp32_only"] = round(rate / vcal["fp32"], 5)  # vs a stream of fp32-class instructions only
                            roofline["valu_calibration"] = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in vcal.items()}
                            wc = sum(v.get("SQ_WAVE_CYCLES", 0.0) for v in main_k)
                            if wc:
                                roofline["wave_cycles_not_valu_frac"] = round(1.0 - sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in main_k) / wc, 5)
                                roofline["wait_inst_any_frac"] = round(sum(v.get("SQ_WAIT_INST_ANY", 0.0) for v in main_k) / wc, 5)
                        tc, av = sum(v.get("SQ_THREAD_CYCLES_VALU", 0.0) for v in main_k), sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in main_k)
                        if tc and av:
                            roofline["valu_lane_utilisation"] = round(tc / (64.0 * av), 5)  # active lanes per issued VALU instruction / 64
                        hit, miss = sum(v.get("TCC_HIT_sum", 0.0) for v in main_k), sum(v.get("TCC_MISS_sum", 0.0) for v in main_k)
                        if hit + miss > 0:
                            roofline["l2_hit_rate"] = round(hit / (hit + miss), 5)
                        if roofline.get("valu_frac") is not None and roofline["frac"] is not None:
                            # the larger of the two fractions names the nearer roofline; when both are far (< 0.6) the kernel is bound by neither -- latency / the
                            # vector-memory request path (tools/ta_calib.hip) -- and `bound` still names the nearer one, with the note saying so
                            roofline["bound"] = "valu" if roofline["valu_frac"] > roofline["frac"] else "hbm"  # derived from THIS run's two fractions, nothing else
                            if max(roofline["valu_frac"], roofline["frac"]) < 0.6:
                                roofline["bound_note"] = "neither measured fraction is near 1 (valu_frac: vs the dual-issue ceiling; valu_frac_fp32_only: vs a one-class stream)"
                            if scattered and main_line:
                                # what calibration experiments found for this kernel family (a citation, not a result of this run): its time follows its instruction COUNT --
                                # +64 instructions per node test = +11 % (r04k), more waves or cheaper instruction classes change nothing (r05r, r05s); DESIGN.md section 4
                                roofline["prior_experiment_note"] = "time follows instruction count: profiles/r04k_valu_sensitivity.txt, r05r_six_waves_variants.txt, r05s_node_test_perm_variants.txt"
                        # raw counters and the per-kernel table go to a FILE (the driver keeps only the last 8 KB of output: r03's line lost C3's value to them)
                        raw["pmc_kernels"] = {k: {c: (round(x, 1) if isinstance(x, float) else x) for c, x in v.items()} for k, v in pmc.items() if k.startswith(prefixes)}
                        if True:  # every kernel of the frame, compactly: time share under counters, VALU issue, lanes, L2 hit (the per-kernel picture of the wavefront pipeline)
                            tot = sum(v.get("pmc_us", 0.0) * v.get("dispatches", 0) for v in pmc.values()) or 1.0
                            raw["all_kernels"] = {
                                k: {"dispatches": v.get("dispatches", 0), "time_share": round(v.get("pmc_us", 0.0) * v.get("dispatches", 0) / tot, 4),
                                    "valu_frac": round(v.get("SQ_INSTS_VALU", 0.0) / (SIMDS * max(v.get("pmc_us", 0.0) * v.get("dispatches", 1) * 1e3, 1e-9)) / vcal["mixed"], 4),
                                    "lanes": round(v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4) if v.get("SQ_ACTIVE_INST_VALU") else None,
                                    "l2_hit": round(v.get("TCC_HIT_sum", 0.0) / (v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0)), 4) if (v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0)) else None,
                                    "fetch_GB": round(v.get("FETCH_SIZE", 0.0) * 1024.0 / 1e9, 3), "write_GB": round(v.get("WRITE_SIZE", 0.0) * 1024.0 / 1e9, 3)}
                                for k, v in sorted(pmc.items(), key=lambda kv: -kv[1].get("pmc_us", 0.0) * kv[1].get("dispatches", 0))[:10]}
            out = {"metric": "Msamples/s (spp x pixels / s) at 8 bounces, 1920x1080", "value": round(value, 2), "unit": "Msamples/s",
                   "n_gpus": args.gpus if args.in_process else world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3 / steps, 3),
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": label, "width": w, "height": h, "spp": rs.spp, "max_bounces": rs.max_bounces,
                              "parallelism": f"rows-interleaved{world}" if world > 1 else (f"in-process rows-interleaved{args.gpus}" if args.in_process and args.gpus > 1 else "single"), "segments_per_sample": round(seg_per_sample, 4),
                              "triangles": cst["triangleCount"], "bvh8_nodes": cst["nodeCount"], "iterations_per_step": stats[-1]["iterations"]},
                   "roofline": roofline}
            out.update(R["first"])  # build_ms / upload_ms / first_frame_ms (outside the timed region)
            out["_raw"] = raw
            if os.environ.get("GATLING_BENCH_CHECKSUM"):  # tests: the frame rank 0 ends up with (host memory), as a checksum
                import hashlib
                img = last["img"]
                out["image_checksum"] = hashlib.sha256((img.numpy() if hasattr(img, "numpy") else img).tobytes()).hexdigest()
        if scene is not None:
            scene.close()
        R["scene"] = None
        return out, R

    out, R = measure(args.workload, args.spp, args.steps, args.warmup, args.no_timers, args.no_pmc, True)
    if args.probe:
        return
    raws = {}
    if out is not None:
        raws[args.workload] = out.pop("_raw", {})
        for k in ("calibration", "valu_calibration"):  # constants of the calibration files: kept in the side file, not in the line
            if k in out["roofline"]:
                raws[args.workload][k] = out["roofline"].pop(k)
    if out is not None and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(R["desc"], R["rs"], R["w"], R["h"])
        out["cpu_baseline"]["reference"] = reference_probe()
    # Further measurements in the same line ("also"), so the driver's record covers every BASELINE config a single GPU can run:
    #   N = 1 (the headline C2 runs in the fused k_path): C3 and C4 -- the wavefront pipeline k_raygen / k_trace_dyn / k_route / k_shade / k_trace_dyn<any> -- and
    #         "c5share": rank 3's interleaved share of C5's 8-GPU partition at full spp, i.e. the per-GPU work of the configuration C5 is specified on
    #         (`projected_8gpu` = 8 x the share's rate; DESIGN.md section 7 says what the projection leaves out);
    #   N = 8: the tiled 4K interior itself.
    # Each leg is a COMPACT object (value, ms_per_step, stage times, roofline fractions); raw counters and per-kernel tables go to profiles/bench_last_pmc.json.
    # Every rank takes part (collectives inside); a failure is reported in the line instead of losing the headline number.
    also = os.environ.get("GATLING_BENCH_ALSO", ("c1,c3,c4,c5share,c5@64" if args.workload == "c2" and not args.spp else "") if world == 1 else ("c5" if world >= 8 else ""))
    extras = []
    for wl_spec in [x for x in also.split(",") if x and x != args.workload]:
        wl, _, wl_spp = wl_spec.partition("@")  # "c5@64": the workload at a reduced spp (the whole 4K frame of C5 on ONE GPU is 8.5 G samples at its own spp)
        extra = {"workload": wl_spec}
        try:
            e_steps = int(os.environ.get("GATLING_BENCH_ALSO_STEPS", ("2" if wl in ("c5share", "c5") else "3") if world == 1 else "2"))
            light = wl == "c1" or bool(wl_spp)  # no counter passes for the small / reduced legs: their kernels are the ones the full legs already characterise
            E, _ = measure(wl, int(wl_spp or os.environ.get("GATLING_BENCH_ALSO_SPP", "0")), e_steps, 1, world > 1, args.no_pmc or world > 1 or light, False)
            if E is not None:
                raws[wl] = E.pop("_raw", {})
                r = E["roofline"]
                extra.update({"value": E["value"], "unit": "Msamples/s", "ms_per_step": E["ms_per_step"], "steps": e_steps, "n_gpus": world,
                              "events_in_timed_region": bool(wl in ("c1", "c2")),  # big-scene legs: stage_ms / avg_launch_us come from one untimed step after the timed ones (timed_run)
                              "segments_per_sample": E["config"]["segments_per_sample"], "iterations_per_step": E["config"]["iterations_per_step"],
                              "build_ms": E.get("build_ms"), "upload_ms": E.get("upload_ms"), "first_frame_ms": E.get("first_frame_ms"),
                              "stage_ms": r.get("stage_ms_per_step"),
                              "roofline": {k: r.get(k) for k in ("bound", "kernel", "achieved", "frac", "traffic", "avg_launch_us", "algorithmic_frac", "valu_frac", "valu_lane_utilisation",
                                                                  "l2_hit_rate", "wait_inst_any_frac", "nodes_per_ray", "tris_per_ray", "valu_frac_fp32_only", "bound_note") if r.get(k) is not None}})  # (GB/s against 8 000; labels of the legs: make_workload / DESIGN.md section 4)
                if r.get("pmc_note") not in (None, "ok"):
                    extra["roofline"]["pmc_note"] = r["pmc_note"]
                if wl == "c5share":
                    extra["projected_8gpu"] = round(8.0 * E["value"], 1)
                    extra["projection_note"] = "8 x one rank's share (rows 3::8, full spp) on ONE GPU; no multi-GPU run exists (DESIGN.md section 7)"
        except Exception as e:  # noqa: BLE001
            extra["error"] = repr(e)[:300]
        extras.append(extra)
    # The delegate's own default workload (hdGatling: ONE sample per pixel and giRender call, 13 bounces, progressive accumulation, renderDelegate.cpp:93-110):
    # milliseconds of one blocking giCRender call incl. the D2H of the colour AOV, bounce-loop iterations per call -- compact legs c3@spp1 / c4@spp1 (tools/lowspp.py)
    if out is not None and world == 1 and args.workload == "c2" and not args.spp and not os.environ.get("GATLING_BENCH_ALSO"):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import lowspp
            for wl in ("c3", "c4"):
                for row in lowspp.measure(wl, [1], 20, quiet=True):
                    extras.append({"workload": f"{wl}@spp1", "what": "one giCRender per frame: spp 1, 13 bounces, progressive, D2H included", "unit": "ms per call",
                                   "ms_per_call": row["ms_per_call_mean"], "ms_per_call_min": row["ms_per_call_min"], "iterations_per_call": row["iterations"],
                                   "value": row["Msamples_per_s"], "value_unit": "Msamples/s", "calls": row["calls"], "stage_ms": row["stage_ms"]})
        except Exception as e:  # noqa: BLE001
            extras.append({"workload": "lowspp", "error": repr(e)[:300]})
    if out is not None and extras:
        out["also"] = extras
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        import ctypes
        for d in (os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")):  # (gpurun_out/ is what travels back from a GPU box)
            try:
                if os.path.isdir(d):
                    json.dump(raws, open(os.path.join(d, "bench_last_pmc.json"), "w"), indent=1)
            except Exception:  # noqa: BLE001
                pass
        sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)  # flush C stdio (RCCL prints a version banner there) so the JSON is the LAST line
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
