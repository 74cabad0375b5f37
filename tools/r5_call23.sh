#!/bin/bash
# round 5, GPU call 23: 1.0f / x as v_rcp_f32 + one Newton step where the wave's arguments allow it (dev_math.h: rcp_rn / rcp_rn3; exhaustive check: call 22)
# against the compiler's division everywhere (norcp_ = the same sources with -DTR_IEEE_RCP, built before the change) and cycle h's build: parity, time, counters
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
python tools/quick_gpu_check.py 2>&1 | tail -2
echo "== bit check, default build"; python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame"
echo "== A/B tile workloads (_ = reciprocal by Newton step, norcp_ = compiler's division, h_ = cycle h)"
AB_WORKLOADS="cornell_box:64 cornell_box:256 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r5w libtrayhip.so libtrayhip_norcp.so libtrayhip_h.so libtrayhip.so libtrayhip_norcp.so
echo "== counters"; rm -f gpurun_out/pmc_ab.txt
for w in cornell_box:64 smallpt:64 dragon:32; do PMC_SETS=1 python tools/pmc_ab.py $w libtrayhip.so libtrayhip_norcp.so > /dev/null; done; cat gpurun_out/pmc_ab.txt
echo "== C5 full detail, frame 64, 128 spp"; C5_FRAME=64 bash tools/c5_libs.sh 128 libtrayhip.so libtrayhip_norcp.so libtrayhip.so libtrayhip_norcp.so 2>&1 | grep Msamples
} 2>&1 | tee gpurun_out/r05_call23.txt
