#!/bin/bash
# tools/abn.sh <out-tag> <rounds> "<lib1.so> <lib2.so> ..." <bench.py arguments...>: same-box comparison of several
# builds of libmppi_hip.so, interleaved round-robin (see tools/ab.sh)
tag=$1; n=$2; libs=$3; shift 3
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
for i in $(seq 1 $n); do
  for lib in $libs; do
    name=$(basename $lib .so)
    MPPI_LIB_PATH=$PWD/$lib python3 bench.py --no-cpu-baseline --repeats 3 --latency-solves 0 --sustained-s 0 "$@" > gpurun_out/$tag/$name.$i.json 2> gpurun_out/$tag/$name.$i.err || { echo "$name failed"; tail -3 gpurun_out/$tag/$name.$i.err; continue; }
    python3 - <<PY
import json
d=json.load(open("gpurun_out/$tag/$name.$i.json"))
print("$tag %-26s %d %-30s ms/step %.4f (min %.4f)  rollout %.4f ms  tail %.4f ms" % ("$name", $i, d["config"]["rollout_variant"], d["ms_per_step"], d["min_ms_per_step"], d["stage_ms"]["rollout_ms"], d["stage_ms"]["reduction_ms"]))
PY
  done
done
