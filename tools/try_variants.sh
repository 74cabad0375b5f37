#!/bin/bash
# First GPU check of the staged variant builds (compiled here, no GPU needed; both are validated functionally on the host by
# tests/test_device_emulation.py):
#   qwide    -DTR_QWIDE            64-B quantised 4-wide BVH<Triangle> nodes in the wavefront traversal (needs TRAYHIP_WF_WIDE=1)
#   m2c      -DTR_MESH_TWO_CHILDREN mesh_traverse of the tile kernel tests both children per step (k_wf_trace_dyn's node step)
#   exact    -DTR_EXACT_FLAT       the flat instance loop tests the BVH<Instance> leaf box too (closes the deviation class; costs a slab test per instance)
#   build:   tools/build_variants.sh
#   run:     gpurun --timeout 1200 -- 'bash tools/try_variants.sh'
# 1. the GPU parity suite against each variant library, 2. the four workloads at 64 spp, default library vs variant (Msamples/s).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/variants; mkdir -p "$OUT"
cd "$ROOT"
for v in m2c qwide exact; do
  V=$ROOT/tray_rust_amd/libtrayhip_$v.so
  [ -f "$V" ] || { echo "variant $v not built (see the header of this script)"; continue; }
  EXTRA=""; [ $v = qwide ] && EXTRA="TRAYHIP_WF_WIDE=1"
  echo "== tests, variant $v"
  env TRAYHIP_LIB=$V $EXTRA timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee "$OUT/tests_$v.log"
done
for wl in cornell_box smallpt dragon tr15_like; do
  echo "== $wl default"; timeout 300 python tools/bench_small.py 64 2 $wl 2>&1 | tail -2 | tee "$OUT/${wl}_default.log"
  [ $wl != tr15_like ] && { echo "== $wl exact"; TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_exact.so timeout 300 python tools/bench_small.py 64 2 $wl 2>&1 | tail -2 | tee "$OUT/${wl}_exact.log"; }
done
for wl in tr15_like; do
  echo "== $wl qwide";   TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_qwide.so TRAYHIP_WF_WIDE=1 timeout 300 python tools/bench_small.py 64 2 $wl 2>&1 | tail -2 | tee "$OUT/${wl}_qwide.log"
done
echo "== dragon m2c (DRAGON_EXTENT 0.2 and 1.0)"
TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_m2c.so timeout 300 python tools/bench_small.py 64 2 dragon 2>&1 | tail -1
DRAGON_EXTENT=1.0 timeout 300 python tools/bench_small.py 64 2 dragon 2>&1 | tail -1
DRAGON_EXTENT=1.0 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_m2c.so timeout 300 python tools/bench_small.py 64 2 dragon 2>&1 | tail -1
echo "== dragon, wavefront schedule, default vs qwide"
TRAYHIP_MODE=wave timeout 300 python tools/bench_small.py 64 2 dragon 2>&1 | tail -1
TRAYHIP_MODE=wave TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_qwide.so TRAYHIP_WF_WIDE=1 timeout 300 python tools/bench_small.py 64 2 dragon 2>&1 | tail -1
