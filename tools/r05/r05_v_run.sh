#!/bin/bash
# round 5, step v: FINAL sources (beta out of the rollout kernel for many-chunk solves): the whole GPU suite, soak, rocprofv3 + PMC of headline / config 4 / K=16384, the table, the driver's line
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_v_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_v_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python3 tools/soak.py 20000 > gpurun_out/r05_v_soak.json 2>/dev/null; echo "soak rc=$?"
bash tools/prof_run.sh r05_v_headline 2>&1 | tail -1
bash tools/prof_run.sh r05_v_cfg4 --K 16384 --T 150 --layers 6-64-64-4 2>&1 | tail -1
bash tools/prof_run.sh r05_v_k16384 --K 16384 2>&1 | tail -1
for t in r05_v_headline r05_v_cfg4 r05_v_k16384; do echo "== $t"; head -5 gpurun_out/prof/$t/summary/kernel_stats.csv; done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_v_bench_driver_args.json 2>/dev/null
python3 bench.py > gpurun_out/r05_v_bench_default.json 2>/dev/null
bash tools/table.sh 2>&1 | grep -v amdgpu.ids
bash tools/loop_time.sh 2>&1 | cut -c1-330
