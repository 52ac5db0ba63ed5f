"""Quick parameter sweeps on the GPU (env-var knobs of gi_c.cpp)."""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(env, spp=256):
    extra = env.pop("ARGS", "").split() if "ARGS" in env else []
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--spp", str(spp), "--no-cpu-baseline"] + extra, env=e, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(env, "FAILED", out.stderr[-500:]); return
    j = json.loads(line[-1])
    print(env, "value", j["value"], "ms", j["ms_per_step"], j["roofline"]["stage_ms_per_step"], "trace_us", j["roofline"]["avg_launch_us"], flush=True)
for a in sys.argv[1:]:
    kv = dict(x.split("=") for x in a.split(",")) if a != "-" else {}
    run(kv)
