#!/bin/bash
# Round 6, GPU call 5: progressive tile slices (whole frame and the 8 shards), the SLP reduction program, the SLP build's samples against the default's
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
timeout 900 python tools/tile_slices_ab.py dragon cornell_box 2>&1 | grep -v "^Frame" | tee gpurun_out/r06_tile_slices_progressive_ab.txt
{
for v in on off; do /opt/rocm/bin/hipcc -O3 -ffp-contract=off $([ $v = off ] && echo -fno-slp-vectorize) --offload-arch=gfx950 tools/experiments/slp_rect_test.hip -o /tmp/slp_rect_$v 2>/dev/null; echo "== slp_rect_test, SLP $v"; /tmp/slp_rect_$v; done
for sc in cornell_box dragon; do
  DUMP_SAVE=/tmp/dump_$sc.npy TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_dump.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 default_flags 2>&1 | grep -v "^Frame" | head -1
  DUMP_COMPARE=/tmp/dump_$sc.npy TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_slpdump.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 slp_on 2>&1 | grep -v "^Frame" | grep -v "   px"
done
} 2>&1 | tee gpurun_out/r06_slp_reduction.txt
