#!/bin/bash
# round 5, step e: granule hand-off to the smoothing workgroup (no acquire): the whole GPU suite, then the headline and the tick
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_e_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05_e_pytest.log
for rep in 1 2; do
for f in "" "--no-chain"; do
  python3 bench.py $f --no-cpu-baseline --latency-solves 0 --sustained-s 0 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('[$f] ms_per_step %.4f (min %.4f max %.4f) cold %.4f  value %.2f M | stage %s' % (d['ms_per_step'], d['min_ms_per_step'], d['max_ms_per_step'], d['cold']['ms_per_step'], d['value']/1e6, {k: round(v,4) for k,v in d['stage_ms'].items() if k.endswith('_ms')}))"
done
done
python3 bench.py --no-cpu-baseline --K 1920 --layers 6-64-64-64-64-4 --latency-solves 0 --sustained-s 0 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('wd K=1920: ms_per_step %.4f value %.2f M %s' % (d['ms_per_step'], d['value']/1e6, d['config']['rollout_variant']))"
python3 bench.py --no-cpu-baseline --K 8192 --latency-solves 0 --sustained-s 0 | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('K=8192: ms_per_step %.4f value %.2f M %s' % (d['ms_per_step'], d['value']/1e6, d['config']['rollout_variant']))"
bash tools/loop_time.sh 2>&1 | tail -8
