#!/bin/bash
# Second short GPU call: the mesh-heavy case for -DTR_MESH_TWO_CHILDREN (320 000 triangles, mesh 5x larger on screen), default vs variant
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; D=/tmp/mini_ab2; mkdir -p gpurun_out
{
timeout 12 python - <<'PY'
import sys
sys.path.insert(0, ".")
from tray_rust_amd import scenes
scenes.write_dragon_assets("/tmp/mini_ab2", film=(1920, 1080, 32), grid=400, extent=1.0)
PY
timeout 10 python tools/mini_ab.py run $D default dragon:32
TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_m2c.so timeout 10 python tools/mini_ab.py run $D m2c dragon:32
} 2>&1 | grep -v "^Frame" | tee gpurun_out/mini_ab_dragon.log
