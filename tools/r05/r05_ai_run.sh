#!/bin/bash
# round 5, step ai: the gate block's replica line carries the head of the nominal sequence (state + U[0..11] in ONE load once the gate is seen
# open) -- one memory round trip less at the start of every chained solve
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python3 -m pytest tests/test_api_gpu.py -m gpu -x -q -k "chained or sequence or ahead" 2>&1 | tail -3 || exit 1
bash tools/abn.sh r05_ai_cfg3 3 "tools/variants/before_head.so tools/variants/gate_head.so" &&
bash tools/abn.sh r05_ai_cfg3_driver 2 "tools/variants/before_head.so tools/variants/gate_head.so" --steps 20 --warmup 5 &&
bash tools/abn.sh r05_ai_wd1920 2 "tools/variants/before_head.so tools/variants/gate_head.so" --K 1920 --layers 6-64-64-64-64-4 &&
bash tools/abn.sh r05_ai_k16384 2 "tools/variants/before_head.so tools/variants/gate_head.so" --K 16384 &&
bash tools/abn.sh r05_ai_cfg4 2 "tools/variants/before_head.so tools/variants/gate_head.so" --K 16384 --T 150 --layers 6-64-64-4 --steps 100 --warmup 10
