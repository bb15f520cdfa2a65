#!/bin/bash
# tools/stream_probe.sh: stamps of the streaming tail at two sizes, then the A/B of tools/tail_ab.sh (round 5, step a: FORMS="old new late"
# -- see that script's header for what became of "old"; "late" was a variant build that published beta together with eta)
cd "$GRAFT_REPO_ROOT" || exit 1
for k in 16384 65536; do
  echo "== stamps K=$k"; python3 tools/stream_stamps.py tools/variants/stamps.so $k 100
done
python3 -m pytest tests/test_api_gpu.py -x -q -k "stream_tail" 2>&1 | tail -3
python3 -m pytest tests/test_parity_gpu.py -x -q -k "many_chunk" 2>&1 | tail -3
bash tools/tail_ab.sh
