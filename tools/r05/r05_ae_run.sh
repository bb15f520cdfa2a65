#!/bin/bash
# round 5, step ae: beta out of the rollout kernel also for the row / m44 forms where the tail is the streaming kernel (4096 < K <= 8192)
cd "$GRAFT_REPO_ROOT" || exit 1
MPPI_LIB_PATH=$PWD/tools/variants/rowpub.so timeout -k 10 300 python3 -m pytest tests/test_stream_tail_gpu.py tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -2
bash tools/abn.sh r05_ae_k8192 3 "tools/variants/final2.so tools/variants/rowpub.so" --K 8192 &&
bash tools/abn.sh r05_ae_cfg3 3 "tools/variants/final2.so tools/variants/rowpub.so" &&
bash tools/abn.sh r05_ae_k6400 2 "tools/variants/final2.so tools/variants/rowpub.so" --K 6400 &&
bash tools/abn.sh r05_ae_k8192_h64 2 "tools/variants/final2.so tools/variants/rowpub.so" --K 8192 --layers 6-64-64-4 &&
bash tools/abn.sh r05_ae_wd1920 2 "tools/variants/final2.so tools/variants/rowpub.so" --K 1920 --layers 6-64-64-64-64-4
