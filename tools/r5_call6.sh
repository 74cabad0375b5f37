#!/bin/bash
# round 5, GPU call 6: the frame's transform table (by shutter-time index) against the per-path cache
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
python tools/quick_gpu_check.py 2>&1 | tail -1
echo "== moving_box (tile kernel, ANIM), 1080p"
for t in 0 1 0 1; do TRAYHIP_XF_TABLE=$t python tools/mini_ab.py run /tmp/mini_ab "table$t" moving_box:32 moving_box:256 2>&1 | grep Msamples; done
for fr in 64 127; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_env.sh 128 "cache=TRAYHIP_XF_TABLE=0" "table=TRAYHIP_XF_TABLE=1" "cache=TRAYHIP_XF_TABLE=0" "table=TRAYHIP_XF_TABLE=1"
done
} 2>&1 | tee gpurun_out/r05_call6.txt
