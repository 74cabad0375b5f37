#!/bin/bash
# round 5, GPU call 15: the instances of a BVH<Instance> leaf behind conservative boxes of their own in the quad records (host/gates.hpp: quad_tree's prim_box)
# against leaves as the reference has them (TRAYHIP_NO_INSTANCE_BOXES=1), same library, same box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
for fr in 64 127 0; do
  echo "== C5 full detail, frame $fr, 128 spp"; C5_FRAME=$fr bash tools/c5_env.sh 128 "leaves=TRAYHIP_NO_INSTANCE_BOXES=1" "boxes=TRAYHIP_X=0" "leaves=TRAYHIP_NO_INSTANCE_BOXES=1" "boxes=TRAYHIP_X=0"
done
echo "== bit check"; python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep "tr15"
TRAYHIP_MODE=wave python tools/r5_bitcheck.py /tmp/mini_ab 20000 2>&1 | grep "dragon\|cornell\|moving" | sed 's/^/forced wavefront: /'
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wavefront or tr15 or textured or views or pool or transform_table or dragon or update or frame" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|rendering took" | tail -3
} 2>&1 | tee gpurun_out/r05_call15.txt
