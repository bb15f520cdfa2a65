#!/bin/bash
# round 5, step p: where does the generator kernel beside the rollout pay beyond one dynamics wave per SIMD?
cd "$GRAFT_REPO_ROOT" || exit 1
run() {  # tag, args...
  tag=$1; shift
  for b in 0 1; do
    e=""; [ $b = 1 ] && e="MPPI_GEN_BESIDE=1"
    env $e python3 bench.py --no-cpu-baseline --latency-solves 0 --sustained-s 0 --repeats 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('beside=$b $tag: ms_per_step %.4f value %.2f M | %s' % (d['ms_per_step'], d['value']/1e6, {k: round(v,4) for k,v in d['stage_ms'].items() if k.endswith('_ms')}))"
  done
}
run h64_k65536_T100 --K 65536 --T 100 --layers 6-64-64-4 --steps 30
run h64_k32768_T100 --K 32768 --T 100 --layers 6-64-64-4 --steps 50
run h64_k24576_T150 --K 24576 --T 150 --layers 6-64-64-4 --steps 50
run h32_k32768_T100 --K 32768 --T 100 --steps 100
run h32_k24576_T100 --K 24576 --T 100 --steps 100
run h32l4_k32768_T100 --K 32768 --T 100 --layers 6-32-32-32-32-4 --steps 50
