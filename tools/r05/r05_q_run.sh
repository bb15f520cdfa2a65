#!/bin/bash
# round 5, step q: generator beside the rollout by rule (+ chained ticks there): parity, suite, the large-K rows
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 -m pytest tests/test_api_gpu.py tests/test_multi_tree_gpu.py -m gpu -x -q -k "chained or generator or multi4_tree" 2>&1 | tail -3
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_q_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_q_pytest.log
row() { tag=$1; shift; python3 bench.py --no-cpu-baseline --latency-solves 0 --sustained-s 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$tag: ms_per_step %.4f (min %.4f) value %.2f M | %s' % (d['ms_per_step'], d['min_ms_per_step'], d['value']/1e6, {k: round(v,4) for k,v in d['stage_ms'].items() if k.endswith('_ms')}))"; }
row k65536 --K 65536 --steps 100
row k32768 --K 32768 --steps 100
row k32768_nochain --K 32768 --steps 100 --no-chain
row k24576 --K 24576 --steps 100
row k32768_h64_T150 --K 32768 --T 150 --layers 6-64-64-4 --steps 50
row k32768_h64_T150_nochain --K 32768 --T 150 --layers 6-64-64-4 --steps 50 --no-chain
row cfg4 --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10
row k16384 --K 16384
