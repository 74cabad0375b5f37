#!/bin/bash
# round 5, GPU call 2: bit check of the restructured libm, same-box A/B (glibc libm vs ocml; same-pixel pair hand-out), configs[4] per frame, C5 counters in the benched schedule
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
echo "== bit check, default build"; python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame"
echo "== bit check, -DTR_OCML_LIBM (rounds 1-4)"; TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_ocml.so python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame"
echo "== A/B tile workloads"; AB_WORKLOADS="cornell_box:64 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r5b libtrayhip.so libtrayhip_ocml.so libtrayhip_pairpix.so libtrayhip.so libtrayhip_ocml.so libtrayhip_pairpix.so
echo "== A/B C5 full detail, frame 330, 128 spp"; bash tools/c5_libs.sh 128 libtrayhip.so libtrayhip_ocml.so libtrayhip.so libtrayhip_ocml.so
echo "== configs[4] per frame"; python tools/r5_c5_frames.py 512
echo "== C5 counters (tr15_like:128, frame 64)"; timeout 600 python tools/pmc_workloads.py r05_b tr15_like:128 2>&1 | tail -3 | cut -c1-1500
} 2>&1 | tee gpurun_out/r05_call2.txt
