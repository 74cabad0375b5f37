#!/bin/bash
# round 5, GPU call 21: the cooperative small-mesh test with the lanes per ray chosen by the number of staged rays (16 / 8 / 4; dev_geom.h: mesh_leaf_coop)
# against the build with four lanes per ray always (-DTR_COOP_QUADS_ONLY = rounds 2-5): parity per camera sample, time, VALU instructions and lanes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
python tools/quick_gpu_check.py 2>&1 | tail -4
echo "== bit check, default build (lanes per ray by staged rays)"; python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame"
echo "== A/B tile workloads"; AB_WORKLOADS="cornell_box:64 cornell_box:256 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r5u libtrayhip.so libtrayhip_quads.so libtrayhip.so libtrayhip_quads.so
echo "== counters, cornell_box 64 spp"; rm -f gpurun_out/pmc_ab.txt; PMC_SETS=1 python tools/pmc_ab.py cornell_box:64 libtrayhip.so libtrayhip_quads.so; cat gpurun_out/pmc_ab.txt
} 2>&1 | tee gpurun_out/r05_call21.txt
