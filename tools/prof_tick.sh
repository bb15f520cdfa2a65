#!/bin/bash
# tools/prof_tick.sh <tag>: the two-controller tick (K = 1920 each) under rocprofv3 --kernel-trace, once with each
# controller's solve on its own stream (round 2) and once batched into one launch; dispatch-timeline summaries and
# kernel stats under gpurun_out/prof/<tag>/summary/ -- copy into profiles/.
set -o pipefail
tag=${1:-tick}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof/$tag
rm -rf "$out" && mkdir -p "$out/summary"
for mode in streams batch one; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$mode" -- python3 tools/tick_time.py --modes $mode --ticks 100 --repeats 3 > "$out/$mode.json" 2> "$out/$mode.err" || { tail -5 "$out/$mode.err"; exit 1; }
  python3 tools/prof_summary.py stats "$out/$mode" "$out/summary/${mode}_kernel_stats.csv"
  python3 tools/prof_summary.py timeline "$out/$mode" "$out/summary/${mode}_timeline.json" "rocprofv3 --kernel-trace --stats -- python3 tools/tick_time.py --modes $mode --ticks 100 --repeats 3"
done
