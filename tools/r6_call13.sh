#!/bin/bash
# Round 6, GPU call 13: the AMDGPU machine scheduler's other strategies (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause / iterative-minreg / iterative-ilp)
# against the default (max occupancy) on the four workloads at full size. Scheduling moves no f32 operation: the bits are the same (RMSE column, parity tests below).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
export AB_DIR=/tmp/mini_full MINI_DRAGON_GRID=660
LIBS=${CALL13_LIBS:-"libtrayhip.so libtrayhip_ilp.so libtrayhip_mclause.so libtrayhip_iminreg.so libtrayhip_iilp.so"}
{
AB_WORKLOADS="cornell_box:64 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r06_sched $LIBS $LIBS
C5_FRAME=64 bash tools/c5_libs.sh 128 $LIBS $LIBS
} 2>&1 | grep -v "^Frame" | tee gpurun_out/r06_sched_strategy_ab.txt
