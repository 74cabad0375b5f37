#!/bin/bash
# round 5, GPU call 7: the transform table -- tests, moving_box on the tile kernel (cache columns filled from the table), C5
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "transform_table or moving or tr15 or wavefront or frame_update or views" 2>&1 | tail -5
python tools/mini_ab.py prepare /tmp/mini_ab > /dev/null 2>&1
echo "== moving_box (tile kernel, ANIM), 1080p"
for t in 0 1 0 1; do TRAYHIP_XF_TABLE=$t python tools/mini_ab.py run /tmp/mini_ab "table$t" moving_box:256 2>&1 | grep Msamples; done
for fr in 64; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_env.sh 128 "cache=TRAYHIP_XF_TABLE=0" "table=TRAYHIP_XF_TABLE=1" "default=TRAYHIP_STATS=0"
done
} 2>&1 | tee gpurun_out/r05_call7.txt
