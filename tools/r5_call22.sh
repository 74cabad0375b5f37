#!/bin/bash
# round 5, GPU call 22: (a) the own-box cull of occlusion rays skipped where the BVH<Instance> leaf holds one instance (its gate has said it all), (b) the
# cooperative test's lanes exchanging (t, k, c2) only -- each apart and together against cycle h's build; (c) tools/ubench_rcp: is 1.0f / x reproduced by
# v_rcp_f32 + Newton steps on every argument?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
python tools/quick_gpu_check.py 2>&1 | tail -2
echo "== bit check, default build"; python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame"
echo "== A/B tile workloads (_ = both, ownalways_ = lean exchange only, wide_ = own-box skip only, h_ = cycle h)"
AB_WORKLOADS="cornell_box:64 cornell_box:256 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r5v libtrayhip.so libtrayhip_h.so libtrayhip_ownalways.so libtrayhip_wide.so libtrayhip.so libtrayhip_h.so
echo "== counters"; rm -f gpurun_out/pmc_ab.txt
PMC_SETS=1 python tools/pmc_ab.py cornell_box:64 libtrayhip.so libtrayhip_h.so libtrayhip_ownalways.so libtrayhip_wide.so > /dev/null
PMC_SETS=1 python tools/pmc_ab.py smallpt:64 libtrayhip.so libtrayhip_h.so > /dev/null; cat gpurun_out/pmc_ab.txt
echo "== reciprocal"; ./tools/ubench_rcp
} 2>&1 | tee gpurun_out/r05_call22.txt
