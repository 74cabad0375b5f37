#!/bin/bash
# round 5, GPU call 20: the pool's ray and throughput groups side by side in one 64-byte record (a 32-byte record costs a 64-byte fetch; k_wf_begin and the query
# kernels read both) against two 32-byte records (-DWF_SPLIT_RAY_THRU = cycle f's layout), same box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
for fr in 64 127; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_f.so libtrayhip.so libtrayhip_f.so libtrayhip.so
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wavefront or tr15 or textured or views or pool or transform_table or update or frame" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|rendering took" | tail -3
} 2>&1 | tee gpurun_out/r05_call20.txt
