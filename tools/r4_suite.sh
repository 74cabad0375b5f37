#!/bin/bash
# Round 4: same-box A/B of the tile workloads and the C5 stand-in against round 3's library, then the whole GPU suite.
#   gpurun --timeout 1500 -- 'bash tools/r4_suite.sh <tag>'
TAG=${1:-a}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
AB_WORKLOADS="cornell_box:64 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r4$TAG libtrayhip_r3.so libtrayhip.so
bash tools/c5_libs.sh 32 libtrayhip_r3.so libtrayhip.so
} 2>&1 | tee gpurun_out/r04_${TAG}_ab.log
timeout 1300 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r04_${TAG}_gpu_suite.log | tail -40
