#!/bin/bash
# round 5, step w: randomised many-chunk solves (tools/stream_soak.py), then the headline's bench.py lines with the final PMC stamp in place
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 60 python3 tools/stream_soak.py 20 100000 || exit 1
timeout -k 10 600 python3 tools/stream_soak.py 420 0 > gpurun_out/r05_w_stream_soak.txt 2>&1; echo "stream soak rc=$?"; tail -3 gpurun_out/r05_w_stream_soak.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_w_bench_driver_args.json 2>/dev/null
python3 bench.py > gpurun_out/r05_w_bench_default.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("gpurun_out/r05_w_bench_driver_args.json", "gpurun_out/r05_w_bench_default.json"):
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][0])
    print(f, "%.2f M  %.4f ms  frac %.4f traffic %s" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
PY
