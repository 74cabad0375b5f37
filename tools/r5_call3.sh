#!/bin/bash
# round 5, GPU call 3: C5 A/B of the pool layout (hit records as records, fields not stored) and of the shading kernels' occupancy
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
python tools/quick_gpu_check.py 2>&1 | tail -2
for fr in 64 330; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_base.so libtrayhip.so libtrayhip_base.so libtrayhip.so
done
echo "== bit check (wavefront scenes)"; python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep tr15
} 2>&1 | tee gpurun_out/r05_call3.txt
