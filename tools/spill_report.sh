#!/bin/bash
# Static register report of a variant build without a GPU: compiles kernels.hip with the given -D flags and prints VGPRs, spilled
# VGPRs and scratch bytes of the kernels the bench runs.   tools/spill_report.sh [-DTR_REMAT_WO -DTR_REMAT_BITAN ...]
set -e
cd "$(dirname "$0")/../tray_rust_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function "$@" -c hip/kernels.hip -o $T/k.o
F=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $T/k.o $T/f.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/f.bin --output=$F --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $F | awk -v all="$SPILL_ALL" '
  /\.name:/ {name=$2}
  /\.private_segment_fixed_size:/ {scr=$2}
  /\.sgpr_spill_count:/ {ss=$2}
  /\.vgpr_count:/ {v=$2}
  /\.vgpr_spill_count:/ {sp=$2; if (all != "" || name ~ /k_path_tilesILi0ELi[014]E|k_wf_queryILi1ELi5|k_wf_beginILi1/) printf "%-46s vgprs %3d  spilled %3d  scratch %4d B  sgprs spilled to lanes %3d\n", substr(name,1,46), v, sp, scr, ss}'
rm -rf $T
