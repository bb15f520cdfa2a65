#!/bin/bash
# round 5, step aj: timing-only probe -- needs tools/variants/skipu.so, a build of abi_solve.hip in which write_gate skips the memcpy of U
# (MPPI_GATE_SKIP_U=1) or copies only its first 12 steps (=2); results are meaningless, the step time says what the host's U write costs
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do for m in 0 1 2; do
MPPI_GATE_SKIP_U=$m MPPI_LIB_PATH=$PWD/tools/variants/skipu.so python3 bench.py --no-cpu-baseline --repeats 3 --latency-solves 0 --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('skip_u=$m ms/step %.4f (min %.4f)' % (d['ms_per_step'], d['min_ms_per_step']))"
done; done
