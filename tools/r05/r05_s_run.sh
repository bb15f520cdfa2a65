#!/bin/bash
# round 5, step s: beta (the minimum cost) leaves the rollout kernel as a tagged 64-bit atomic minimum; the tail kernels read it
# instead of reducing the costs (MPPI_MIN_COST=0: the old way, same build)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_s_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r05_s_pytest.log; [ $rc -eq 0 ] || exit 1
row() { tag=$1; mc=$2; shift 2; MPPI_MIN_COST=$mc python3 bench.py --no-cpu-baseline --latency-solves 0 --sustained-s 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$tag min_cost=$mc: ms_per_step %.4f (min %.4f) value %.2f M | %s' % (d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6, {k: round(v,4) for k,v in d['stage_ms'].items() if k.endswith('_ms')}))"; }
for i in 1 2 3; do
  for mc in 0 1; do
    row cfg3 $mc
    row k16384 $mc --K 16384
    row cfg4 $mc --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10
  done
done
for mc in 0 1 0 1; do
  row cfg2 $mc --K 2048
  row k8192 $mc --K 8192
  row k65536 $mc --K 65536 --steps 100
  row wd1920 $mc --K 1920 --T 100 --layers 6-64-64-64-64-4
done
