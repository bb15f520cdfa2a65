#!/bin/bash
# round 5, step r: chunk minima / sums of the many-chunk tail in ONE hop (every weights workgroup stores its chunk's value into all
# replicas; a workgroup collects one replica) against the two-hop form (exchange among weights workgroups, then {beta, eta} lines)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 python3 -m pytest tests/test_stream_tail_gpu.py tests/test_api_gpu.py -m gpu -x -q -k "stream or fault or timeout" 2>&1 | tail -3 || exit 1
bash tools/abn.sh r05_r_k16384 3 "tools/variants/tail2hop.so tools/variants/tail1hop.so" --K 16384 &&
bash tools/abn.sh r05_r_cfg4 3 "tools/variants/tail2hop.so tools/variants/tail1hop.so" --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10 &&
bash tools/abn.sh r05_r_k32768 2 "tools/variants/tail2hop.so tools/variants/tail1hop.so" --K 32768 --steps 100 &&
bash tools/abn.sh r05_r_k65536 2 "tools/variants/tail2hop.so tools/variants/tail1hop.so" --K 65536 --steps 100
