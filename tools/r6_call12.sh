#!/bin/bash
# Round 6, GPU call 12 (C4): the workgroup gathers the rays that enter the dragon's mesh and walks the tree with full waves (TRAYHIP_WG_COMPACT=1;
# dev_geom.h: DevScene::wg_compact). prev_ = the build before the code existed, _ = with the code, switched off / on. Full-size mesh (871 200 triangles).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
export AB_DIR=/tmp/mini_full MINI_DRAGON_GRID=660
WL="cornell_box:64 smallpt:64 dragon:32 dragon:256"
{
echo "== switched off"
AB_WORKLOADS="$WL" bash tools/ab.sh r06_wg_off libtrayhip_prev.so libtrayhip.so libtrayhip_prev.so libtrayhip.so
echo "== TRAYHIP_WG_COMPACT=1 (only the dragon has a large mesh)"
TRAYHIP_WG_COMPACT=1 AB_WORKLOADS="dragon:32 dragon:256 cornell_box:64" bash tools/ab.sh r06_wg_on libtrayhip.so libtrayhip_prev.so libtrayhip.so
} 2>&1 | tee gpurun_out/r06_c4_wg_compact_ab.txt
TRAYHIP_WG_COMPACT=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "dragon or c4 or mesh" 2>&1 | grep "passed\|failed" | tail -2 | tee -a gpurun_out/r06_c4_wg_compact_ab.txt
