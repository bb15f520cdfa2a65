#!/bin/bash
# round 5, step aa: the streaming tail also for 4096 < K <= 8192 (two chunks per row; today: solve_tail_wide_kernel, every workgroup
# recomputes beta and eta from all K costs, the two chunks of a row meet at an arrival counter) -- variant build against the product
cd "$GRAFT_REPO_ROOT" || exit 1
MPPI_LIB_PATH=$PWD/tools/variants/stream_from_4096.so timeout -k 10 300 python3 -m pytest tests/test_parity_gpu.py tests/test_api_gpu.py -m gpu -x -q -k "8192 or 6400 or chained or sequence" 2>&1 | tail -3
bash tools/abn.sh r05_aa_k8192 3 "tools/variants/final.so tools/variants/stream_from_4096.so" --K 8192 &&
bash tools/abn.sh r05_aa_k6400 2 "tools/variants/final.so tools/variants/stream_from_4096.so" --K 6400 &&
bash tools/abn.sh r05_aa_k8192_h64 2 "tools/variants/final.so tools/variants/stream_from_4096.so" --K 8192 --layers 6-64-64-4
