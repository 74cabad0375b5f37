#!/bin/bash
# round 5, GPU call 14: the wavefront traversal's leaf phase two triangles per pass in packed arithmetic (TriPair records) against cycle e's build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
for fr in 64 127; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_e.so libtrayhip.so libtrayhip_e.so libtrayhip.so
done
echo "== bit check"; python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep "tr15\|dragon"
TRAYHIP_MODE=wave python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep "dragon\|cornell" | sed 's/^/forced wavefront: /'
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wavefront or tr15 or textured or views or pool or transform_table or dragon" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|rendering took" | tail -3
} 2>&1 | tee gpurun_out/r05_call14.txt
