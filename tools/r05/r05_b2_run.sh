#!/bin/bash
# round 5, step b (continued): one chunk of the wd nominal-margin sweep (+ the wait-timeout test)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 200 python3 -m pytest tests/test_api_gpu.py -m gpu -x -q -k "wait_timeout" 2>&1 | tail -3
timeout -k 10 1000 python3 tools/nominal_margin.py wd ${WD:-2150} ${WDFIRST:-600700} > gpurun_out/nominal_margin_wd_${WDFIRST:-600700}.txt 2> gpurun_out/nominal_margin_wd.err; echo "wd rc=$?"; tail -2 gpurun_out/nominal_margin_wd.err; cat gpurun_out/nominal_margin_wd_${WDFIRST:-600700}.txt
