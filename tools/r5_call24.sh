#!/bin/bash
# round 5, GPU call 24: the reciprocal by Newton step with the range guard restricted to the lanes that hold a ray at the world-space site only (build 3)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
echo "== bit check, default build"; python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame"
AB_WORKLOADS="cornell_box:64 cornell_box:256 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r5x libtrayhip.so libtrayhip_norcp.so libtrayhip.so libtrayhip_norcp.so
echo "== counters"; rm -f gpurun_out/pmc_ab.txt
for w in cornell_box:64 smallpt:64 dragon:32; do PMC_SETS=1 python tools/pmc_ab.py $w libtrayhip.so > /dev/null; done; cat gpurun_out/pmc_ab.txt
} 2>&1 | tee gpurun_out/r05_call24.txt
