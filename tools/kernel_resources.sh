#!/bin/bash
# tools/kernel_resources.sh <csrc/file.hip> [name filter]: VGPRs / spills / LDS / occupancy of every kernel in
# the file (hipcc -Rpass-analysis=kernel-resource-usage, same flags as autorally_amd/build.py)
f=$1; pat=${2:-.}
x=""; case "$f" in *rollout_row.hip) x="-mllvm -amdgpu-sched-strategy=max-ilp";; esac  # build.py: EXTRA_FLAGS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form $x \
  --cuda-device-only -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|Spill|LDS Size|Occupancy|SGPRs:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' |
  awk '/Function Name/{if(n)print n; n=$0; next}{gsub(/^ +/,"");n=n" | "$0}END{print n}' | c++filt | grep -E "$pat" |
  sed -e 's/Function Name: //' -e 's/mppi:://g' -e 's/(mppi::RolloutArgs)//'
