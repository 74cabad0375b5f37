#!/bin/bash
# One short GPU call: default library and every staged variant on cut-down C2 / C4 / C5 workloads (tools/mini_ab.py, torch-free).
#   tools/build_variants.sh && gpurun --timeout 60 -- 'bash tools/mini_ab.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; D=/tmp/mini_ab; mkdir -p gpurun_out
L=$ROOT/tray_rust_amd
{
timeout 15 python tools/mini_ab.py prepare $D
timeout 10 python tools/mini_ab.py run $D default cornell_box:64 dragon:32 tr15_like:16
TRAYHIP_LIB=$L/libtrayhip_m2c.so   timeout 10 python tools/mini_ab.py run $D m2c dragon:32
TRAYHIP_LIB=$L/libtrayhip_state.so timeout 10 python tools/mini_ab.py run $D state cornell_box:64 dragon:32 tr15_like:16
TRAYHIP_LIB=$L/libtrayhip_wq3.so    timeout 10 python tools/mini_ab.py run $D wq3 tr15_like:16
TRAYHIP_LIB=$L/libtrayhip_state2.so timeout 10 python tools/mini_ab.py run $D state2 cornell_box:64 dragon:32
TRAYHIP_LIB=$L/libtrayhip_lazy.so  timeout 10 python tools/mini_ab.py run $D lazy cornell_box:64 dragon:32
TRAYHIP_LIB=$L/libtrayhip_exact.so timeout 10 python tools/mini_ab.py run $D exact cornell_box:64 dragon:32
TRAYHIP_LIB=$L/libtrayhip_qwide.so TRAYHIP_WF_WIDE=1 timeout 10 python tools/mini_ab.py run $D qwide tr15_like:16
TRAYHIP_LIB=$L/libtrayhip_combo3.so timeout 10 python tools/mini_ab.py run $D combo3 cornell_box:64 dragon:32 tr15_like:16
TRAYHIP_LIB=$L/libtrayhip_combo2.so timeout 10 python tools/mini_ab.py run $D combo2 cornell_box:64 dragon:32
TRAYHIP_MODE=wave timeout 10 python tools/mini_ab.py run $D wave dragon:32
# where the wave cycles go (instrumented build: stages of the tile kernel, parts of the BSDF queries)
[ -f $L/libtrayhip_clk.so ] && TRAYHIP_LIB=$L/libtrayhip_clk.so TRAYHIP_STATS=1 timeout 10 python tools/mini_ab.py run $D clk cornell_box:64 dragon:32
[ -f $L/libtrayhip_clkstate.so ] && TRAYHIP_LIB=$L/libtrayhip_clkstate.so TRAYHIP_STATS=1 timeout 10 python tools/mini_ab.py run $D clkstate cornell_box:64
[ -x tools/ubench_valu ] && timeout 20 tools/ubench_valu
} 2>&1 | grep -v "^Frame" | tee gpurun_out/mini_ab.log
