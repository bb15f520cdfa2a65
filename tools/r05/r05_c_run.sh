#!/bin/bash
# round 5, step c: gate microbenchmark, the whole GPU suite on the pruned library, the last chunk of the wd margin sweep
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== gate_ub"; timeout -k 5 120 tools/variants/gate_ub 34 3.5 2.5 3000 2>&1 | tail -12
echo "== gate_ub (short rollout: 10 us)"; timeout -k 5 120 tools/variants/gate_ub 10 3.5 2.5 3000 2>&1 | tail -6
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_c_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05_c_pytest.log
timeout -k 10 700 python3 tools/nominal_margin.py wd ${WD:-2150} ${WDFIRST:-602850} > gpurun_out/nominal_margin_wd_${WDFIRST:-602850}.txt 2> gpurun_out/nominal_margin_wd.err; echo "wd rc=$?"; tail -2 gpurun_out/nominal_margin_wd.err; cat gpurun_out/nominal_margin_wd_${WDFIRST:-602850}.txt
