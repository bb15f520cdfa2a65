#!/bin/bash
# round 5, step g: the whole GPU suite on the final sources, soak of the chained ticks against the batched (unchained) ticks,
# the newest shipped model with and without the chain, rocprofv3 summaries of the headline and of the many-chunk workloads
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s > gpurun_out/r05_g_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_g_pytest.log; grep "nominal margin" gpurun_out/r05_g_pytest.log | sort -u | head -40
timeout -k 10 300 python3 tools/soak.py 20000 > gpurun_out/r05_g_soak.json 2> gpurun_out/r05_g_soak.err; echo "soak rc=$?"; cat gpurun_out/r05_g_soak.json
for f in "" "--no-chain"; do
  python3 bench.py $f --no-cpu-baseline --K 1920 --layers 6-64-64-64-64-4 --latency-solves 0 --sustained-s 0 --event-solves 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('wd K=1920 [$f]: ms_per_step %.4f value %.2f M %s' % (d['ms_per_step'], d['value']/1e6, d['config']['rollout_variant']))"
  python3 bench.py $f --no-cpu-baseline --K 4096 --layers 6-64-64-4 --latency-solves 0 --sustained-s 0 --event-solves 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('6-64-64-4 K=4096 [$f]: ms_per_step %.4f value %.2f M %s' % (d['ms_per_step'], d['value']/1e6, d['config']['rollout_variant']))"
done
bash tools/prof_run.sh r05_g_headline 2>&1 | tail -2
bash tools/prof_run.sh r05_g_cfg4 --K 16384 --T 150 --layers 6-64-64-4 2>&1 | tail -2
for t in r05_g_headline r05_g_cfg4; do echo "== $t"; cat gpurun_out/prof/$t/summary/kernel_stats.csv; done
