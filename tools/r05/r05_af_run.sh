#!/bin/bash
# round 5, step af: long runs on the final build -- random many-chunk shapes three ways, then 40 000 ticks per latency form chained against batched
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 700 python3 tools/stream_soak.py 360 400000 > gpurun_out/r05_af_stream_soak.txt 2> gpurun_out/r05_af_stream_soak.err; echo "stream soak rc=$?"; tail -1 gpurun_out/r05_af_stream_soak.txt
(while sleep 60; do echo "# soak running"; done) & ticker=$!
timeout -k 10 450 python3 tools/soak.py 40000 > gpurun_out/r05_af_soak.json 2>/dev/null; rc=$?; kill $ticker; echo "soak rc=$rc"; cut -c1-400 gpurun_out/r05_af_soak.json
