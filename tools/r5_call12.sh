#!/bin/bash
# round 5, GPU call 12: the moving-scene instantiations of the tile kernel at 4 waves per SIMD (-DTR_MIN_WAVES_ANIM=4: 128 VGPRs, ~110 spilled) against 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r5l libtrayhip.so libtrayhip_anim4.so libtrayhip.so libtrayhip_anim4.so
} 2>&1 | tee gpurun_out/r05_call12.txt
