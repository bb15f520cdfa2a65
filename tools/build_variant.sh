#!/bin/bash
# tools/build_variant.sh <name> <file.hip> <extra hipcc flags...>: a diagnostic / A-B build of libmppi_hip.so in which ONE source
# file is compiled with extra flags (-DMPPI_TAIL_STAMPS, ...), linked with the product build's other objects, as
# tools/variants/<name>.so (git-ignored; travels to the GPU box; loaded through MPPI_LIB_PATH)
name=$1; src=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
python3 -c "import sys; sys.path.insert(0, '$root'); from autorally_amd import build as B; B.build()" || exit 1
mkdir -p "$root/tools/variants" /tmp/bv_$$
X=""; case "$src" in rollout_row.hip|rollout_row64.hip) X="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
  -Xarch_host -mavx2 -Xarch_host -mfma $X "$@" -c "$root/autorally_amd/csrc/$src" -o /tmp/bv_$$/variant.o || exit 1
objs=$(ls "$root"/autorally_amd/build/*.o | grep -v "/${src%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/variants/$name.so" $objs /tmp/bv_$$/variant.o && rm -rf /tmp/bv_$$ && ls -la "$root/tools/variants/$name.so"
