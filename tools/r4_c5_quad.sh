#!/bin/bash
# Round 4, first GPU call of the quad-record traversal: same-box A/B against round 3's library on the C5 stand-in at full detail,
# record order (breadth-first levels) sweep, per-ray step counts, phase clocks, and the wavefront parity tests.
#   gpurun --timeout 900 -- 'bash tools/r4_c5_quad.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
{
echo "== A/B 32 spp"; bash tools/c5_libs.sh 32 libtrayhip_r3.so libtrayhip.so libtrayhip_r3.so libtrayhip.so
echo "== record order"; bash tools/c5_env.sh 32 "bfs0=TRAYHIP_QUAD_BFS=0" "bfs3=TRAYHIP_QUAD_BFS=3" "bfs7=TRAYHIP_QUAD_BFS=7" "bfs9=TRAYHIP_QUAD_BFS=9"
echo "== steps per ray"; TRAYHIP_STATS=1 bash tools/c5_libs.sh 32 libtrayhip_stats.so 2>&1 | grep -v "^\[trayhip\] \(traversal\|tile\|dynamic\)" | tail -8
echo "== phase clocks"; TRAYHIP_STATS=1 bash tools/c5_libs.sh 32 libtrayhip_clocks.so 2>&1 | grep "wave cycles\|Msamples" | tail -8
} 2>&1 | tee gpurun_out/r4_c5_quad.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wavefront or tr15 or dragon or frame_update or moving_scene_image" 2>&1 | tail -5 | tee gpurun_out/r4_c5_quad_tests.log
timeout 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "c5 or c4" 2>&1 | tail -12 | tee -a gpurun_out/r4_c5_quad_tests.log
