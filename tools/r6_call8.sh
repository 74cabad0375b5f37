#!/bin/bash
# Round 6, GPU call 8 (moving scenes on the tile kernel): the per-lane pass of the flat loop taking the MOVING spheres / rectangles / disks
# (FlatInst::lane_pass = 2), the first pending mover's transform requested inside the uniform loop (prefetch), the fill of the cache columns from the
# frame's table dealt out to the whole wave (coop), and the two ceilings: cache columns that stay in L2 (colmask), columns + table in L2 (colkmask; WRONG pictures both).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r06_moving_box libtrayhip_nomov.so libtrayhip.so libtrayhip_prefetch.so libtrayhip_coop.so libtrayhip_cooppre.so libtrayhip_colmask.so libtrayhip_colkmask.so \
    libtrayhip_nomov.so libtrayhip.so libtrayhip_prefetch.so libtrayhip_coop.so libtrayhip_cooppre.so
for lib in libtrayhip.so libtrayhip_cooppre.so; do
  echo "== parity of moving scenes, $lib"
  TRAYHIP_LIB=$ROOT/tray_rust_amd/$lib timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "moving or anim" 2>&1 | tail -3
done 2>&1 | tee gpurun_out/r06_moving_box_parity.txt
