#!/bin/bash
# round 5, step y: FINAL sources (beta out of the rollout kernel, eta one hand-over from the chunk sums): the whole GPU suite, soak, rocprofv3 + PMC of headline / config 4 / K=16384, the table, the driver's line
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_y_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_y_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python3 tools/soak.py 20000 > gpurun_out/r05_y_soak.json 2>/dev/null; echo "soak rc=$?"
bash tools/prof_run.sh r05_y_headline 2>&1 | tail -1
bash tools/prof_run.sh r05_y_cfg4 --K 16384 --T 150 --layers 6-64-64-4 2>&1 | tail -1
bash tools/prof_run.sh r05_y_k16384 --K 16384 2>&1 | tail -1
for t in r05_y_headline r05_y_cfg4 r05_y_k16384; do echo "== $t"; head -5 gpurun_out/prof/$t/summary/kernel_stats.csv; done
# the counter summaries of THESE sources where bench.py looks for them (roofline.traffic is null without a summary stamped with the build's sources)
for t in headline cfg4 k16384; do cp gpurun_out/prof/r05_y_$t/summary/pmc.json profiles/r05_y_${t}_pmc.json; done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_y_bench_driver_args.json 2>/dev/null
python3 bench.py > gpurun_out/r05_y_bench_default.json 2>/dev/null
bash tools/table.sh 2>&1 | grep -v amdgpu.ids
bash tools/loop_time.sh 2>&1 | cut -c1-330
timeout -k 10 400 python3 tools/stream_soak.py 240 200000 > gpurun_out/r05_y_stream_soak.txt 2>&1; echo "stream soak rc=$?"; tail -1 gpurun_out/r05_y_stream_soak.txt
