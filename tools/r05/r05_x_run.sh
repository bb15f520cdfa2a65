#!/bin/bash
# round 5, step x: eta one hand-over from the chunk sums (every weights workgroup stores its chunk's sum into all 32 replicas, a
# workgroup collects one replica) against the shipped form (exchange among the weights workgroups, then {eta} lines), beta out of
# the rollout kernel in both
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 python3 -m pytest tests/test_stream_tail_gpu.py tests/test_api_gpu.py -m gpu -x -q -k "stream or fault or beta or min_cost" 2>&1 | tail -3 || exit 1
bash tools/abn.sh r05_x_k16384 3 "tools/variants/twohop_eta.so tools/variants/onehop_eta.so" --K 16384 &&
bash tools/abn.sh r05_x_cfg4 3 "tools/variants/twohop_eta.so tools/variants/onehop_eta.so" --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10 &&
bash tools/abn.sh r05_x_k12352 2 "tools/variants/twohop_eta.so tools/variants/onehop_eta.so" --K 12352 &&
bash tools/abn.sh r05_x_k32768 2 "tools/variants/twohop_eta.so tools/variants/onehop_eta.so" --K 32768 --steps 100 &&
bash tools/abn.sh r05_x_k65536 2 "tools/variants/twohop_eta.so tools/variants/onehop_eta.so" --K 65536 --steps 100
