#!/bin/bash
# round 5, GPU call 1: parity of the glibc-libm build (bit-identical share per camera sample, against round 4's ocml form) and what it costs
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
python tools/quick_gpu_check.py 2>&1 | tail -4
echo "== bit check, default build (glibc libm restated)"; python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame"
echo "== bit check, -DTR_OCML_LIBM (rounds 1-4)"; TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_ocml.so python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame"
echo "== A/B tile workloads"; AB_WORKLOADS="cornell_box:64 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r5a libtrayhip.so libtrayhip_ocml.so libtrayhip_nolsv.so libtrayhip.so libtrayhip_ocml.so
echo "== A/B C5 full detail"; bash tools/c5_libs.sh 32 libtrayhip.so libtrayhip_ocml.so libtrayhip_nolsv.so libtrayhip.so
} 2>&1 | tee gpurun_out/r05_call1.txt
