#!/bin/bash
# round 5, GPU call 17: what a cache-resident transform table would give (-DTR_XF_KIDX_MASK=1023: every path reads one of 1024 time indices -- WRONG pictures,
# the timing ceiling of any scheme that orders samples by time)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
echo "== moving_box (tile kernel)"; AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r5q libtrayhip.so libtrayhip_kmask.so libtrayhip.so libtrayhip_kmask.so
for fr in 64 127; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip.so libtrayhip_kmask.so libtrayhip.so libtrayhip_kmask.so
done
} 2>&1 | tee gpurun_out/r05_call17.txt
