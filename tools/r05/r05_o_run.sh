#!/bin/bash
# round 5, step o: K = 65 536 / 32 768: the generator kernel beside the rollout (lowest stream priority) instead of behind it
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for b in 0 1; do
  e=""; [ $b = 1 ] && e="MPPI_GEN_BESIDE=1"
  env $e python3 bench.py --no-cpu-baseline --K 65536 --steps 100 --latency-solves 0 --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('beside=$b k65536: ms_per_step %.4f (min %.4f) value %.2f M | %s' % (d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6, {k: round(v,4) for k,v in d['stage_ms'].items() if k.endswith('_ms')}))"
  env $e python3 bench.py --no-cpu-baseline --K 32768 --T 150 --layers 6-64-64-4 --steps 50 --latency-solves 0 --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('beside=$b k32768_h64: ms_per_step %.4f (min %.4f) value %.2f M | %s' % (d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6, {k: round(v,4) for k,v in d['stage_ms'].items() if k.endswith('_ms')}))"
done
done
