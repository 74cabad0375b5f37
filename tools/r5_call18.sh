#!/bin/bash
# round 5, GPU call 18: the same with 2^20 of the 2^24 time indices (-DTR_XF_KIDX_MASK=0xFFFFF: a sixteenth of the table -- 0.6 GB for moving_box, 1.4 GB for the
# tr15 stand-in --, far beyond the caches: what locality WITHOUT reuse gives -- TLB reach, DRAM pages)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
echo "== moving_box (tile kernel)"; AB_WORKLOADS="moving_box:128" bash tools/ab.sh r5r libtrayhip.so libtrayhip_kmask20.so libtrayhip_kmask.so libtrayhip.so libtrayhip_kmask20.so
echo "== C5 full detail, frame 127, 128 spp"; C5_FRAME=127 bash tools/c5_libs.sh 128 libtrayhip.so libtrayhip_kmask20.so libtrayhip.so libtrayhip_kmask20.so
} 2>&1 | tee gpurun_out/r05_call18.txt
