#!/bin/bash
# Builds the staged variant libraries (and clk: the instrumented build with wave clocks per stage) next to libtrayhip.so (hipcc cross-compiles gfx950 without a GPU; ~45 s each). They are
# git-ignored and travel to the GPU box with gpurun; tools/try_variants.sh A/Bs them there.
set -e
cd "$(dirname "$0")/../tray_rust_amd/csrc"
for v in qwide:-DTR_QWIDE exact:-DTR_EXACT_FLAT m2c:-DTR_MESH_TWO_CHILDREN lazy:-DTR_RECT_LAZY "state:-DTR_REMAT_WO -DTR_REMAT_BITAN -DTR_NO_LANE_O -DTR_SHARE_WIL -DTR_WAVE_COUNTERS -DTR_CAMERA_PTR" "wq3:-DTR_REMAT_WO -DTR_REMAT_BITAN -DTR_NO_LANE_O -DTR_SHARE_WIL -DTR_WAVE_COUNTERS -DTR_CAMERA_PTR -DWF_QUERY_WAVES=3" "state2:-DTR_REMAT_WO -DTR_REMAT_BITAN -DTR_NO_LANE_O -DTR_SHARE_WIL -DTR_WAVE_COUNTERS -DTR_CAMERA_PTR -DTR_MIN_WAVES=2" clk:-DTR_STAGE_CLOCKS=1; do
  make OUT=../libtrayhip_${v%%:*}.so KOBJ=hip/kernels_${v%%:*}.o EXTRA_HIPFLAGS="${v#*:}"
done
ls -la ../libtrayhip_*.so
