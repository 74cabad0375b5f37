#!/bin/bash
# Round 6, GPU call 6: k_wf_begin fetching its records together with the flags word (against the build of measurement cycle a), the persistent
# k_sampler_pass (rates of the side paths + their GPU parity tests)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
{
for fr in 64 127; do C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_prev.so libtrayhip.so libtrayhip_prev.so libtrayhip.so; done
} 2>&1 | tee gpurun_out/r06_c5_begin_prefetch_ab.txt
python tools/r4_side_paths.py 2>&1 | grep -v "^Frame" | tee gpurun_out/r06_side_paths_persistent.txt
TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_prev.so python tools/r4_side_paths.py 2>&1 | grep -v "^Frame" | grep "Uniform\|Adaptive\|flag" | sed 's/^/previous build: /' | tee -a gpurun_out/r06_side_paths_persistent.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "sampler or adaptive or uniform or rank_4 or animated_mesh or whitted or c2 or c3 or c4" 2>&1 | tail -5
