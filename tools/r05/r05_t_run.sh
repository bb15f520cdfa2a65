#!/bin/bash
# round 5, step t: many-chunk tail with (a) beta out of the rollout kernel (tagged atomic minimum) and (b) eta from a per-row exchange of
# chunk sums among the row's own workgroups -- against the shipped form of r05_n (tools/variants/tail2hop.so), same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 python3 -m pytest tests/test_stream_tail_gpu.py tests/test_api_gpu.py tests/test_multi_tree_gpu.py -m gpu -x -q 2>&1 | tail -3 || exit 1
row() { tag=$1; lib=$2; mc=$3; shift 3; MPPI_LIB_PATH=$PWD/tools/variants/$lib.so MPPI_MIN_COST=$mc python3 bench.py --no-cpu-baseline --latency-solves 0 --sustained-s 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('%-8s %-8s min_cost=$mc: ms_per_step %.4f (min %.4f) value %.2f M | rollout %.4f tail %.4f' % ('$tag', '$lib', d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6, d['stage_ms']['rollout_ms'], d['stage_ms']['reduction_ms']))"; }
for i in 1 2 3; do
  for v in "tail2hop 0" "rowsum 0" "rowsum 1"; do
    set -- $v
    row k16384 $1 $2 --K 16384
    row cfg4 $1 $2 --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10
  done
done
for i in 1 2; do
  for v in "tail2hop 0" "rowsum 0" "rowsum 1"; do
    set -- $v
    row k12352 $1 $2 --K 12352
    row k32768 $1 $2 --K 32768 --steps 100
    row k65536 $1 $2 --K 65536 --steps 100
  done
done
