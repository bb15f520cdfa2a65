#!/bin/bash
# round 5, step z: what re-association costs at the 1e-4 mark for the THIRD automatic form (multi4_tree, K > 8192): draws at the launch
# defaults with the shipped 6-32-32-4 weights at K = 12288 / 16384, exact multi4 and multi4_tree against the nominal oracle
cd "$GRAFT_REPO_ROOT" || exit 1
n=${1:-2500}; first=${2:-500000}
timeout -k 10 1100 python3 tools/nominal_margin.py nn32_big $n $first > gpurun_out/r05_z_nominal_margin_nn32_big_$first.txt 2> gpurun_out/r05_z_progress_$first.txt; echo "rc=$?"; tail -8 gpurun_out/r05_z_nominal_margin_nn32_big_$first.txt | cut -c1-260
