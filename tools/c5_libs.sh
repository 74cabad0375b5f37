#!/bin/bash
# C5 stand-in at full detail for several library builds: gpurun -- 'bash tools/c5_libs.sh <spp> libtrayhip.so libtrayhip_x.so ...'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; SPP=${1:-32}; shift
specs=()
for l in "$@"; do specs+=("$(basename $l .so | sed 's/libtrayhip_\?//')_=TRAYHIP_LIB=$ROOT/tray_rust_amd/$l"); done
bash tools/c5_env.sh $SPP "${specs[@]}"
