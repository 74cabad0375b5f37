#!/bin/bash
# round 5, GPU call 11: the every-lobe (FEAT = 15) and Whitted instantiations of the tile kernel compiled for 2 waves per SIMD (256 VGPRs: the
# per-hit copy of a textured material no longer spills) against 3 (-DTR_MIN_WAVES_SIDE=3 = rounds 3-4), same box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
for lib in libtrayhip.so libtrayhip_side3.so libtrayhip.so libtrayhip_side3.so; do
  echo "== $lib"; TRAYHIP_LIB=$ROOT/tray_rust_amd/$lib python tools/r4_side_paths.py /tmp/side 2>&1 | grep -i "whitted\|textured"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "whitted or textured or ggx or texture or lobes or materials" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|rendering took" | tail -3
} 2>&1 | tee gpurun_out/r05_call11.txt
