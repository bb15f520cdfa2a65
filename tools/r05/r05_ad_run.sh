#!/bin/bash
# round 5, step ad: the randomised whole-solve parity of tests/test_fuzz_gpu.py over further seeds on the FINAL sources (the latency forms
# gained gated instantiations this round, the multi forms the minimum-cost epilogue, K > 4096 the streaming tail): every form, then the three
# automatic re-associated forms against the NOMINAL oracle
cd "$GRAFT_REPO_ROOT" || exit 1
for f in all row_tree m44 multi4_tree_gen; do
  arg=$f; [ $f = all ] && arg=""
  FUZZ_SECONDS=${1:-200} timeout -k 10 600 python3 tools/fuzz_sweep.py 700000 760000 $arg > gpurun_out/r05_ad_fuzz_$f.txt 2> gpurun_out/r05_ad_fuzz_$f.err; echo "$f rc=$?"; tail -2 gpurun_out/r05_ad_fuzz_$f.txt | cut -c1-400
done
