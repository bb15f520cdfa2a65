#!/bin/bash
# round 5, step j: the cancel path of chained ticks, the randomised many-chunk tail tests, then the whole suite once more
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 -m pytest tests/test_api_gpu.py tests/test_stream_tail_gpu.py -m gpu -x -q -k "chained or stream or two_iterations" 2>&1 | tail -5
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_j_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_j_pytest.log
