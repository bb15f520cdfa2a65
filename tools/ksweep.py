#!/usr/bin/env python3
"""tools/ksweep.py  -- rollout-kernel time over K for one network / horizon / variant (HIP events, separate pass).
   python3 tools/ksweep.py --layers 6-64-64-4 --T 150 --variant fused --K 1024,4096,8192,16384"""
import argparse, json, subprocess, sys, os
ap = argparse.ArgumentParser()
ap.add_argument("--layers", default=""); ap.add_argument("--T", type=int, default=100)
ap.add_argument("--variant", default="auto"); ap.add_argument("--K", default="4096")
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for K in a.K.split(","):
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--repeats", "3", "--latency-solves", "0",
           "--K", K, "--T", str(a.T), "--steps", str(a.steps), "--warmup", "5", "--variant", a.variant]
    if a.layers:
        cmd += ["--layers", a.layers]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print("K", K, "FAILED", r.stderr[-300:]); continue
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    print("K %6s %-30s rollout %.4f ms  noise %.4f  tail %.4f  step %.4f ms (min %.4f)  %.1f TF (%.1f%%)" % (
        K, d["config"]["rollout_variant"], d["stage_ms"]["rollout_ms"], d["stage_ms"]["noise_ms"], d["stage_ms"]["reduction_ms"],
        d["ms_per_step"], d["min_ms_per_step"], d["roofline"]["achieved"], 100 * d["roofline"]["frac"]), flush=True)
