#!/bin/bash
# round 5, step f: chained ticks v2 (U through the gate block, publish-only tails): parity, headline A/B
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 -m pytest tests/test_api_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "chained or device_resident or warm_start or two_iterations or slide" 2>&1 | tail -5
for rep in 1 2; do
for f in "" "--no-chain"; do
  python3 bench.py $f --no-cpu-baseline --latency-solves 0 --sustained-s 0 --event-solves 0 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('[$f] ms_per_step %.4f (min %.4f max %.4f) cold %.4f  value %.2f M' % (d['ms_per_step'], d['min_ms_per_step'], d['max_ms_per_step'], d['cold']['ms_per_step'], d['value']/1e6))"
done
done
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --sustained-s 0.5 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('driver args: ms_per_step %.4f first %.4f cold %.4f value %.2f M; per_solve median %.4f' % (d['ms_per_step'], d['first_block_ms_per_step'], d['cold']['ms_per_step'], d['value']/1e6, d['per_solve_ms']['median']))"
python3 bench.py --no-cpu-baseline --K 8192 --latency-solves 0 --sustained-s 0 --event-solves 0 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('K=8192: ms_per_step %.4f value %.2f M %s' % (d['ms_per_step'], d['value']/1e6, d['config']['rollout_variant']))"
python3 bench.py --no-cpu-baseline --K 2048 --latency-solves 0 --sustained-s 0 --event-solves 0 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('K=2048: ms_per_step %.4f value %.2f M %s' % (d['ms_per_step'], d['value']/1e6, d['config']['rollout_variant']))"
