#!/bin/bash
# round 5, step d: chained control ticks -- parity test, A/B of the headline with / without the chain, row_split ub form 7
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 -m pytest tests/test_api_gpu.py -m gpu -x -q -k "chained or device_resident or wait_timeout" 2>&1 | tail -5
timeout -k 10 200 python3 -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "device_resident or warm_start" 2>&1 | tail -3
echo "== row_split_ub"; timeout -k 5 60 tools/variants/row_split_ub 2>&1 | tail -12
for rep in 1 2 3; do
for chain in 1 0; do
  f=""; [ $chain = 0 ] && f="--no-chain"
  python3 bench.py $f --no-cpu-baseline --latency-solves 0 --sustained-s 0 --event-solves 0 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('chain $chain: ms_per_step %.4f (min %.4f max %.4f) first block %.4f cold %.4f  value %.2f M' % (d['ms_per_step'], d['min_ms_per_step'], d['max_ms_per_step'], d['first_block_ms_per_step'], d['cold']['ms_per_step'], d['value']/1e6))"
done
done
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('driver args, chain 1: ms_per_step %.4f first %.4f cold %.4f value %.2f M; per_solve median %.4f' % (d['ms_per_step'], d['first_block_ms_per_step'], d['cold']['ms_per_step'], d['value']/1e6, d['per_solve_ms']['median']))"
