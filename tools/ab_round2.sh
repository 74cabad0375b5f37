#!/bin/bash
# A/B of the round-2 default build against variant builds named on the command line (libtrayhip_<name>.so), cut-down workloads
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; D=/tmp/mini_ab; mkdir -p gpurun_out
L=$ROOT/tray_rust_amd
{
timeout 20 python tools/mini_ab.py prepare $D
timeout 15 python tools/mini_ab.py run $D default cornell_box:64 smallpt:64 dragon:32 tr15_like:16
for v in "$@"; do
  [ -f $L/libtrayhip_$v.so ] || continue
  if [ $v = clk ]; then TRAYHIP_LIB=$L/libtrayhip_$v.so TRAYHIP_STATS=1 timeout 15 python tools/mini_ab.py run $D $v cornell_box:64 smallpt:64 dragon:32
  else TRAYHIP_LIB=$L/libtrayhip_$v.so timeout 15 python tools/mini_ab.py run $D $v cornell_box:64 smallpt:64 dragon:32; fi
done
timeout 15 python tools/mini_ab.py run $D default2 cornell_box:64
} 2>&1 | grep -v "^Frame" | tee gpurun_out/ab_round2.log
