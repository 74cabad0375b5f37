#!/bin/bash
# round 5, GPU call 13: a frame update without the per-frame host work on the meshes (flatten's 0.1 s copy of the mesh arrays, scene_build's 0.08 s walk over
# the BVH<Triangle> nodes for the stack depth): the bench's C5 entry (two frames, the update inside the timed region)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "update or frame or sequence or moving" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|rendering took" | tail -3
timeout 600 python bench.py --workload tr15_like --frames 2 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C5 frames 63 - 64 at 512 spp:', b['value'], 'Msamples/s', b['ms_per_step'], 'ms per step; kernels', b['roofline'].get('kernel_ms'))"
} 2>&1 | tee gpurun_out/r05_call13.txt
