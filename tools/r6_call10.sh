#!/bin/bash
# Round 6, GPU call 10: the transform table's records at 128 bytes (one cache line each, TR_XF_REC 32) against 112 (camoff_: the cooperative fill at 112),
# two records per lane and trip in the fill (two_), the camera's record requested before the fill (early_, earlytwo_); moving_box on the tile kernel and the C5 stand-in (whose stage kernels gather the records directly).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
LIBS="libtrayhip_camoff.so libtrayhip.so libtrayhip_two.so libtrayhip_early.so libtrayhip_earlytwo.so"
AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r06_moving_box_rec128 $LIBS $LIBS
{
for fr in 64 127; do C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_camoff.so libtrayhip.so libtrayhip_camoff.so libtrayhip.so; done
} 2>&1 | tee gpurun_out/r06_c5_rec128_ab.txt
{
echo "== parity of moving scenes and of table mode, libtrayhip.so"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "moving or anim or transform_table or c5 or tr15" 2>&1 | tail -3
echo "== libtrayhip_two.so"
TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_two.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "moving or transform_table" 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r06_moving_box_rec128_parity.txt
