#!/bin/bash
# round 5, step i: final check of everything on the final sources: the whole GPU suite, smoke, co-residency probe, headline re-profiled
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_i_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 tools/coresidency_probe.py 2>/dev/null
bash tools/prof_run.sh r05_i_headline 2>&1 | tail -1
bash tools/prof_run.sh r05_i_cfg4 --K 16384 --T 150 --layers 6-64-64-4 2>&1 | tail -1
bash tools/prof_run.sh r05_i_k16384 --K 16384 2>&1 | tail -1
for t in r05_i_headline r05_i_cfg4 r05_i_k16384; do echo "== $t"; cat gpurun_out/prof/$t/summary/kernel_stats.csv | head -5; done
