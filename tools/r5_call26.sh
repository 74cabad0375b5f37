#!/bin/bash
# round 5, GPU call 26: C5 and C4 of the final build against a build with cycle h's behaviour (-DTR_OWN_BOX_ALWAYS -DTR_COOP_WIDE_PAYLOAD -DTR_IEEE_RCP) on ONE box:
# the bench lines of cycles h and i (237.6 / 231.3, 707.5 / 703.2) come from different boxes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
C5_FRAME=64 timeout 100 bash tools/c5_libs.sh 128 libtrayhip.so libtrayhip_h.so 2>&1 | grep Msamples
AB_DIR=/tmp/mini_full MINI_DRAGON_GRID=660 AB_WORKLOADS="dragon:32" timeout 60 bash tools/ab.sh r5z libtrayhip.so libtrayhip_h.so libtrayhip.so libtrayhip_h.so
} 2>&1 | tee gpurun_out/r05_call26.txt
