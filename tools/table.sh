#!/bin/bash
# tools/table.sh: the measurement table of DESIGN.md / BASELINE.md -- bench.py on every BASELINE configuration and the
# other sizes quoted there, one line each; full JSON lines under gpurun_out/table/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/table
row() {  # tag, then bench.py arguments
  tag=$1; shift
  python3 bench.py "$@" > gpurun_out/table/$tag.json 2> gpurun_out/table/$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/table/$tag.err; return; }
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/table/$tag.json").read().splitlines() if l.startswith("{")][0])
r=d["roofline"]; c=d.get("cpu_baseline") or {}
print("%-14s %-32s %8.2f M/s  step %.4f ms (median %.4f, min %.4f)  solve-median %.4f | rollout %.4f noise %.4f tail %.4f ms | %5.1f TF %4.1f%% | cpu %s (1 thr %s)" % (
  "$tag", d["config"]["rollout_variant"], d["value"]/1e6, d["ms_per_step"], d["median_ms_per_step"], d["min_ms_per_step"],
  (d.get("per_solve_ms") or {}).get("median", float("nan")), d["stage_ms"]["rollout_ms"], d["stage_ms"]["noise_ms"], d["stage_ms"]["reduction_ms"],
  r["achieved"], 100*r["frac"], ("%.3f M/s x%d" % (c["value"]/1e6, c["cores"])) if c else "-", ("%.4f M/s" % (c["single_thread"]["value"]/1e6)) if c else "-"))
PY
}
row cfg1 --K 128 --T 50
row cfg2 --K 2048 --T 100
row cfg3
row cfg4 --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10
row cfg4_valu --K 16384 --T 150 --layers 6-64-64-4 --variant valu --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 --latency-solves 20 --event-solves 10 --prime-ms 50
row k8192 --K 8192 --no-cpu-baseline
row k16384 --K 16384 --no-cpu-baseline
row k65536 --K 65536 --no-cpu-baseline --steps 100
row k32768_h64 --K 32768 --T 150 --layers 6-64-64-4 --no-cpu-baseline --steps 50
row k4096_h64 --K 4096 --T 100 --layers 6-64-64-4 --no-cpu-baseline
row k1920_wd --K 1920 --T 100 --layers 6-64-64-64-64-4 --no-cpu-baseline
row k4096_l4 --K 4096 --T 100 --layers 6-32-32-32-32-4 --no-cpu-baseline
row bf2560 --dynamics basis --K 2560
