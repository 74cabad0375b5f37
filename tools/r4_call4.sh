cd $GRAFT_REPO_ROOT
bash tools/c5_libs.sh 32 libtrayhip_r3.so libtrayhip.so libtrayhip_r3.so libtrayhip.so 2>&1 | tee gpurun_out/r4_call4.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wavefront or tr15 or frame_update_equals" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee -a gpurun_out/r4_call4.log
