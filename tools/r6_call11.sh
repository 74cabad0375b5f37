#!/bin/bash
# Round 6, GPU call 11: 128-byte table records on the tile kernel's moving scenes (camoff_ = the same build at 112 bytes), then measurement cycle d of the final build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r06_moving_box_rec128_only libtrayhip_camoff.so libtrayhip.so libtrayhip_camoff.so libtrayhip.so
bash tools/r6_measure.sh d
