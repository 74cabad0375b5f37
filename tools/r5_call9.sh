#!/bin/bash
# round 5, GPU call 9: the flags word field-major again (k_wf_advance's scan), the hit record without it
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
for fr in 64 127; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_tab.so libtrayhip.so libtrayhip_tab.so libtrayhip.so
done
echo "== bit check"; python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep "tr15"
} 2>&1 | tee gpurun_out/r05_call9.txt
