#!/bin/bash
# Same-box A/B of library builds on the cut-down workloads: tools/ab.sh <tag> <lib>... (each lib twice, interleaved; torch-free)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
D=${AB_DIR:-/tmp/mini_ab}   # AB_DIR=/tmp/mini_full MINI_DRAGON_GRID=660 MINI_TR15_DETAIL=1.0: the scenes at full size
[ -f $D/cornell_box.json ] || python tools/mini_ab.py prepare $D > /dev/null 2>&1
WL=${AB_WORKLOADS:-"cornell_box:64 smallpt:64 dragon:32"}
for rep in 1; do
  for lib in "$@"; do
    TRAYHIP_LIB=$ROOT/tray_rust_amd/$lib python tools/mini_ab.py run $D "$(basename $lib .so | sed 's/libtrayhip_\?//')_" $WL 2>&1 | grep Msamples
  done
done | tee gpurun_out/ab_$TAG.log
