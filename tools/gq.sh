#!/bin/bash
# tools/gq.sh <timeout s> <command...>: gpurun, waiting (not retrying a run) while no GPU slot is free (exit code 3: nothing ran)
t=$1; shift
# the snapshot ships the in-tree build: make sure it is the build of the sources as they are
(cd "$(dirname "$0")/.." && python3 -c "from autorally_amd import build as B; B.build()" > /dev/null) || { echo "build failed"; exit 1; }
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
