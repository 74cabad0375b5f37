#!/bin/bash
# round 5, GPU call 16: the instances that move within the frame behind their EXACT swept boxes (k_xf_table_bounds: the union over the transform table's 2^24
# records) instead of infinite ones (TRAYHIP_NO_SWEPT_BOXES=1), same library, same box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
D=/tmp/mini_ab; [ -f $D/cornell_box.json ] || python tools/mini_ab.py prepare $D > /dev/null 2>&1
{
echo "== moving_box (tile kernel), 32 / 128 spp"
for rep in 1 2; do
  TRAYHIP_NO_SWEPT_BOXES=1 python tools/mini_ab.py run $D "infinite_" moving_box:32 moving_box:128 2>&1 | grep Msamples
  python tools/mini_ab.py run $D "swept_" moving_box:32 moving_box:128 2>&1 | grep Msamples
done
for fr in 64 127; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_env.sh 128 "infinite=TRAYHIP_NO_SWEPT_BOXES=1" "swept=TRAYHIP_X=0" "infinite=TRAYHIP_NO_SWEPT_BOXES=1" "swept=TRAYHIP_X=0"
done
echo "== bit check with the table (and the boxes) forced on"; TRAYHIP_XF_TABLE=1 python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep "tr15\|moving"
TRAYHIP_XF_TABLE=1 TRAYHIP_MODE=wave python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep "moving" | sed 's/^/forced wavefront: /'
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "wavefront or tr15 or transform_table or update or frame or moving or sequence" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|rendering took" | tail -3
} 2>&1 | tee gpurun_out/r05_call16.txt
