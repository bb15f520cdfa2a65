#!/bin/bash
# round 5, step a: the whole GPU suite on the streaming tail, then rocprofv3 summaries of the three many-chunk workloads
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_a_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05_a_pytest.log
bash tools/prof_run.sh r05_a_k16384 --K 16384 2>&1 | tail -3
bash tools/prof_run.sh r05_a_cfg4 --K 16384 --T 150 --layers 6-64-64-4 2>&1 | tail -3
bash tools/prof_run.sh r05_a_k65536 --K 65536 2>&1 | tail -3
for t in r05_a_k16384 r05_a_cfg4 r05_a_k65536; do echo "== $t"; cat gpurun_out/prof/$t/summary/kernel_stats.csv; done
