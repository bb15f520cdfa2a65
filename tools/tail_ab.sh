#!/bin/bash
# tools/tail_ab.sh: the tail stage of many-chunk solves (K > 8192), builds against each other on one box, alternating, bench.py's own
# stage events and step time.  A form is a variant build tools/variants/<form>.so (tools/build_variant.sh, or a copy of an older
# libmppi_hip.so); a name without such a file runs the product build.  HISTORY: profiles/r05_a_tail_stream_probe.txt was made with
# FORMS="old new", "old" being the round-4 two-launch tail (weights_kernel + solve_tail_kernel<PRE>) that the library then still
# carried behind MPPI_TAIL_FORM=old; that switch was removed with the old kernels at the end of round 5 -- to
# repeat that A/B, build the library of commit aa93161 as tools/variants/old.so.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/tail_ab
row() {  # tag, form, then bench.py arguments
  tag=$1; form=$2; shift; shift
  lib=""; [ -f tools/variants/$form.so ] && lib=$PWD/tools/variants/$form.so  # a form that is a variant build (tools/build_variant.sh)
  MPPI_LIB_PATH=$lib MPPI_TAIL_FORM=$form python3 bench.py --no-cpu-baseline "$@" > gpurun_out/tail_ab/$tag.$form.json 2> gpurun_out/tail_ab/$tag.$form.err || { echo "$tag $form FAILED"; tail -3 gpurun_out/tail_ab/$tag.$form.err; return; }
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/tail_ab/$tag.$form.json").read().splitlines() if l.startswith("{")][0])
print("%-12s %-4s %-36s %8.2f M/s  step %.4f ms (min %.4f) | rollout %.4f noise %.4f tail %.4f ms" % (
  "$tag", "$form", d["config"]["rollout_variant"], d["value"]/1e6, d["ms_per_step"], d["min_ms_per_step"],
  d["stage_ms"]["rollout_ms"], d["stage_ms"]["noise_ms"], d["stage_ms"]["reduction_ms"]))
PY
}
for rep in 1 2; do
for form in ${FORMS:-new}; do
row k16384 $form --K 16384
row k65536 $form --K 65536 --steps 100
row cfg4 $form --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10
row k12288 $form --K 12288
done
done
