#!/bin/bash
# round 5, step u: beta out of the rollout kernel for many-chunk solves (final form): the suite, then the rows it touches
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_u_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r05_u_pytest.log; [ $rc -eq 0 ] || exit 1
row() { tag=$1; mc=$2; shift 2; MPPI_MIN_COST=$mc python3 bench.py --no-cpu-baseline --latency-solves 0 --sustained-s 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('%-8s min_cost=$mc: ms_per_step %.4f (min %.4f) value %.2f M | rollout %.4f tail %.4f' % ('$tag', d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6, d['stage_ms']['rollout_ms'], d['stage_ms']['reduction_ms']))"; }
for i in 1 2 3; do
  for mc in 0 1; do
    row cfg3 $mc
    row k12352 $mc --K 12352
    row k16384 $mc --K 16384
    row cfg4 $mc --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10
    row k32768 $mc --K 32768 --steps 100
  done
done
