#!/bin/bash
# Round 6, GPU call 9: the cooperative table fill handing the moving CAMERA's record over as well (one trip to the table per regeneration step instead of two;
# camoff_ = fill only), against the build before (nomov_) and the all-in-L2 ceiling (colkmask_, WRONG pictures); table-mode parity tests on the new default.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
LIBS=${CALL9_LIBS:-"libtrayhip_nomov.so libtrayhip_camoff.so libtrayhip.so libtrayhip_colkmask.so"}
AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r06_moving_box_cam $LIBS $LIBS
{
echo "== parity of moving scenes and of table mode, libtrayhip.so"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "moving or anim or transform_table or whitted or sampler" 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r06_moving_box_cam_parity.txt
