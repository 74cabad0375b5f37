#!/bin/bash
# Static register report of a variant build without a GPU: compiles kernels.hip with the given -D flags and prints VGPRs, spilled
# VGPRs and scratch bytes of the kernels the bench runs.   tools/spill_report.sh [-DTR_REMAT_WO -DTR_REMAT_BITAN ...]
set -e
cd "$(dirname "$0")/../tray_rust_amd/csrc"
T=$(mktemp -d)
# (the kernel groups of hip/kernel_list.h the report looks at: the static tile kernels and the wavefront schedule's shading side for moving scenes;
#  SPILL_GROUPS="0 1 .. 9" for others)
for g in ${SPILL_GROUPS:-0 1 8}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function "$@" -DTR_INST_GROUP=$g -c hip/kernel_group.hip -o $T/k$g.o &
done
wait
for g in ${SPILL_GROUPS:-0 1 8}; do ../../tools/code_objects.sh $T/k$g.o $T/co$g; done | xargs -n1 /opt/rocm/lib/llvm/bin/llvm-readelf --notes | awk -v all="$SPILL_ALL" '
  /\.name:/ {name=$2}
  /\.private_segment_fixed_size:/ {scr=$2}
  /\.sgpr_spill_count:/ {ss=$2}
  /\.vgpr_count:/ {v=$2}
  /\.vgpr_spill_count:/ {sp=$2; if (all != "" || name ~ /k_path_tilesILi0ELi[014]E|k_wf_queryILi1ELi5|k_wf_beginILi1/) printf "%-46s vgprs %3d  spilled %3d  scratch %4d B  sgprs spilled to lanes %3d\n", substr(name,1,46), v, sp, scr, ss}'
rm -rf $T
