#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; D=/tmp/mini_ab; mkdir -p gpurun_out
L=$ROOT/tray_rust_amd
{
timeout 15 python tools/mini_ab.py prepare $D
timeout 10 python tools/mini_ab.py run $D default cornell_box:64
for v in wo bitan nolo wil wcnt camp; do
TRAYHIP_LIB=$L/libtrayhip_s_$v.so timeout 10 python tools/mini_ab.py run $D $v cornell_box:64
done
timeout 10 python tools/mini_ab.py run $D default2 cornell_box:64
} 2>&1 | grep -v "^Frame" | tee gpurun_out/ab_single.log
