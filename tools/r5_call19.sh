#!/bin/bash
# round 5, GPU call 19: refill / node-phase thresholds of the OCCLUSION stage of the wavefront traversal on their own (its rays end at the first hit; lanes 0.25)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
for fr in 127 64; do
  echo "== C5 full detail, frame $fr, 128 spp (default: refill at 24 idle lanes, node phase while 16 lanes have node work)"
  C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip.so libtrayhip_rb12.so libtrayhip_rb40.so libtrayhip_nb8.so libtrayhip_nb28.so libtrayhip.so
done
} 2>&1 | tee gpurun_out/r05_call19.txt
