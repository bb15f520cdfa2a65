#!/bin/bash
# round 5, step b: new tests (wait timeout, bench N>1 basis), then the nominal-margin sweep at the launch defaults
cd "$GRAFT_REPO_ROOT" || exit 1
nproc; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -m1 "model name" /proc/cpuinfo
timeout -k 10 400 python3 -m pytest tests/test_api_gpu.py tests/test_bench_multiprocess.py -m gpu -x -q -k "wait_timeout or stream_tail or bench" > gpurun_out/r05_b_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05_b_pytest.log
timeout -k 10 500 python3 tools/nominal_margin.py nn32 ${NN32:-5000} 500000 > gpurun_out/nominal_margin_nn32.txt 2> gpurun_out/nominal_margin_nn32.err; echo "nn32 rc=$?"; tail -3 gpurun_out/nominal_margin_nn32.err; cat gpurun_out/nominal_margin_nn32.txt
timeout -k 10 450 python3 tools/nominal_margin.py wd ${WD:-700} ${WDFIRST:-600000} > gpurun_out/nominal_margin_wd_${WDFIRST:-600000}.txt 2> gpurun_out/nominal_margin_wd.err; echo "wd rc=$?"; tail -3 gpurun_out/nominal_margin_wd.err; cat gpurun_out/nominal_margin_wd_${WDFIRST:-600000}.txt
