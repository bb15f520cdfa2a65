#!/bin/bash
# tools/ab.sh <out-tag> <libA.so> <libB.so> <rounds> <bench.py arguments...>
# Same-box A/B of two builds of libmppi_hip.so (MPPI_LIB_PATH override of autorally_amd/capi.py),
# alternating A, B, A, B ... so that clock drift of the box hits both arms alike.
tag=$1; A=$2; B=$3; n=$4; shift 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
for i in $(seq 1 $n); do
  for arm in A B; do
    lib=$A; [ $arm = B ] && lib=$B
    MPPI_LIB_PATH=$PWD/$lib python3 bench.py --no-cpu-baseline --repeats 3 --latency-solves 0 --sustained-s 0 "$@" > gpurun_out/$tag/$arm$i.json 2> gpurun_out/$tag/$arm$i.err || { echo "arm $arm failed"; tail -3 gpurun_out/$tag/$arm$i.err; exit 1; }
    python3 - <<PY
import json
d=json.load(open("gpurun_out/$tag/$arm$i.json"))
print("$tag $arm$i %-28s value %.3fM  ms/step %.4f (min %.4f)  rollout %.4f ms  tail %.4f ms" % (d["config"]["rollout_variant"], d["value"]/1e6, d["ms_per_step"], d["min_ms_per_step"], d["stage_ms"]["rollout_ms"], d["stage_ms"]["reduction_ms"]))
PY
  done
done
