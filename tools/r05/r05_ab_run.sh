#!/bin/bash
# round 5, step ab: the streaming tail serves every K > 4096 (solve_tail_wide_kernel and its counter hand-over removed): the suite, then
# the product against the build before (tools/variants/final.so) on the rows the change can touch
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_ab_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r05_ab_pytest.log; [ $rc -eq 0 ] || exit 1
bash tools/abn.sh r05_ab_cfg3 3 "tools/variants/final.so tools/variants/nowide.so" &&
bash tools/abn.sh r05_ab_k8192 2 "tools/variants/final.so tools/variants/nowide.so" --K 8192 &&
bash tools/abn.sh r05_ab_cfg2 2 "tools/variants/final.so tools/variants/nowide.so" --K 2048 &&
bash tools/abn.sh r05_ab_wd1920 2 "tools/variants/final.so tools/variants/nowide.so" --K 1920 --layers 6-64-64-64-64-4 &&
bash tools/abn.sh r05_ab_k16384 2 "tools/variants/final.so tools/variants/nowide.so" --K 16384
