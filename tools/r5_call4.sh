#!/bin/bash
# round 5, GPU call 4: 1 / wd kept for the return from a mesh, slerp's sin / cos from one reduction, the traversal kernel at 5 waves per SIMD
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
for fr in 64 127; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_rec.so libtrayhip.so libtrayhip_tw5.so libtrayhip_rec.so libtrayhip.so libtrayhip_tw5.so
done
echo "== moving_box (tile kernel, ANIM)"; AB_WORKLOADS="moving_box:32 cornell_box:64" bash tools/ab.sh r5d libtrayhip_rec.so libtrayhip.so libtrayhip_rec.so libtrayhip.so
echo "== bit check"; python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep "tr15\|moving"
} 2>&1 | tee gpurun_out/r05_call4.txt
