#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 -m pytest tests/test_api_gpu.py -m gpu -x -q -k "chained" 2>&1 | tail -3
for f in "" "--no-chain"; do
  python3 bench.py $f --no-cpu-baseline --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10 --latency-solves 0 --sustained-s 0 --event-solves 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('gen at gate open: cfg4 [$f]: ms_per_step %.4f (min %.4f) value %.2f M' % (d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6))"
  python3 bench.py $f --no-cpu-baseline --K 16384 --latency-solves 0 --sustained-s 0 --event-solves 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('gen at gate open: k16384 [$f]: ms_per_step %.4f (min %.4f) value %.2f M' % (d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6))"
  python3 bench.py $f --no-cpu-baseline --K 65536 --steps 100 --latency-solves 0 --sustained-s 0 --event-solves 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('gen at gate open: k65536 [$f]: ms_per_step %.4f (min %.4f) value %.2f M' % (d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6))"
done
