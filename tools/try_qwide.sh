#!/bin/bash
# First GPU check of the quantised 4-wide traversal (variant build -DTR_QWIDE, hip/wavefront_wide.h + host/wide_nodes.hpp).
# Build here (no GPU needed):   make -C tray_rust_amd/csrc OUT=../libtrayhip_qwide.so KOBJ=hip/kernels_qwide.o EXTRA_HIPFLAGS=-DTR_QWIDE
# Then on the GPU box:          gpurun --timeout 900 -- 'bash tools/try_qwide.sh'
# 1. the wavefront / mesh / tr15 parity tests against the variant library with the wide traversal on,
# 2. the C5 and C4 stand-ins at 32 spp, default library vs variant (Msamples/s).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/qwide; mkdir -p "$OUT"
cd "$ROOT"
V=$ROOT/tray_rust_amd/libtrayhip_qwide.so
[ -f "$V" ] || { echo "build the variant first (see the header of this script)"; exit 1; }
TRAYHIP_LIB=$V TRAYHIP_WF_WIDE=1 TRAYHIP_MODE=wave timeout 600 python -m pytest tests -q -m gpu -k "wavefront or dragon or mesh or tr15 or moving or intersect" 2>&1 | tail -5 | tee "$OUT/tests.log"
for wl in tr15_like dragon; do
  echo "== $wl default";      TRAYHIP_MODE=wave timeout 300 python tools/bench_small.py 32 2 $wl 2>&1 | tail -2 | tee "$OUT/${wl}_default.log"
  echo "== $wl qwide";        TRAYHIP_LIB=$V TRAYHIP_WF_WIDE=1 TRAYHIP_MODE=wave timeout 300 python tools/bench_small.py 32 2 $wl 2>&1 | tail -2 | tee "$OUT/${wl}_qwide.log"
done
