#!/bin/bash
# tools/form_selection.sh: the selection table against the clock (tests marked `timing`: never part of -m gpu), on the GPU box
cd "$GRAFT_REPO_ROOT" || exit 1
python3 -m pytest tests/test_form_selection_gpu.py -m timing -q -s 2>&1 | grep -E "form selection|passed|failed|Error" 
