#!/bin/bash
# Material sort of the wavefront schedule's shading stage (k_wf_begin's LDS counting sort -> k_wf_query_kind) on / off, and the
# wavefront schedule against the tile megakernel on the small scenes.   gpurun -- 'bash tools/ab_sort.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; D=/tmp/mini_ab; mkdir -p gpurun_out
{
timeout 20 python tools/mini_ab.py prepare $D
echo "== tile megakernel (default schedule of the small scenes)"
timeout 15 python tools/mini_ab.py run $D mega cornell_box:64 smallpt:64 dragon:32
echo "== wavefront schedule, shading sorted by material kind (default)"
TRAYHIP_MODE=wave timeout 30 python tools/mini_ab.py run $D wave+sort cornell_box:64 smallpt:64 dragon:32 tr15_like:16
echo "== wavefront schedule, one thread per pool slot in the shading stage (TRAYHIP_WF_SORT=0)"
TRAYHIP_MODE=wave TRAYHIP_WF_SORT=0 timeout 30 python tools/mini_ab.py run $D wave-sort cornell_box:64 smallpt:64 dragon:32 tr15_like:16
} 2>&1 | grep -v "^Frame" | tee gpurun_out/ab_sort.log
