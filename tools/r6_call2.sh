#!/bin/bash
# Round 6, GPU call 2: (1) ray binning before the traversal stages (TRAYHIP_WF_BIN: 0 off, 1 stage A, 2 stage B, 3 both; the bin3 build has 8 cells
# per axis instead of 4) on C5's frames 64 and 127 at 128 spp; (2) the SLP vectoriser / strict-aliasing builds on the cut-down tile workloads
# (RMSE against the default build's render + rate); (3) C5's film against the oracle with binning on (the GPU suite's full-size C5 test).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -2
{
for fr in 64 127; do
  C5_FRAME=$fr bash tools/c5_env.sh 128 "bin0_f$fr=TRAYHIP_WF_BIN=0" "bin1_f$fr=TRAYHIP_WF_BIN=1" "bin3_f$fr=TRAYHIP_WF_BIN=3" "bin0_f$fr=TRAYHIP_WF_BIN=0" "bin3_f$fr=TRAYHIP_WF_BIN=3"
  [ -f tray_rust_amd/libtrayhip_bin3.so ] && C5_FRAME=$fr bash tools/c5_env.sh 128 "cells8_bin3_f$fr=TRAYHIP_WF_BIN=3 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_bin3.so" "cells8_bin1_f$fr=TRAYHIP_WF_BIN=1 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_bin3.so"
done
} 2>&1 | tee gpurun_out/r06_c5_binning_ab.txt
LIBS="libtrayhip.so"; for v in slp slpnsa nsa; do [ -f tray_rust_amd/libtrayhip_$v.so ] && LIBS="$LIBS libtrayhip_$v.so"; done
bash tools/ab.sh r06_slp $LIBS $LIBS 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "c5" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^Frame [0-9]*: rendering took" | tail -15 | tee gpurun_out/r06_c5_fullsize_binned.txt
