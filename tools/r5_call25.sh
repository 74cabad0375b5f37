#!/bin/bash
# round 5, GPU call 25: the moving-scene instantiations of the tile kernel with / without the lanes-with-a-ray mask on the world-space reciprocal's guard
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
AB_WORKLOADS="moving_box:32 moving_box:128 cornell_box:64" bash tools/ab.sh r5y libtrayhip.so libtrayhip_b4.so libtrayhip_norcp.so libtrayhip.so libtrayhip_b4.so libtrayhip_norcp.so
} 2>&1 | tee gpurun_out/r05_call25.txt
