#!/bin/bash
# PMC passes around ONE launch of the tile kernel (torch-free runner: tools/mini_ab.py), one counter set per rocprofv3 run, plus
# the FETCH_SIZE / WRITE_SIZE calibration on a known scratch pattern (tools/scratch_calib).   tools/pmc_tile.sh <tag> [scene] [spp]
set -u
TAG=${1:-x}; WL=${2:-cornell_box}; PSPP=${3:-64}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
D=${MINI_AB_DIR:-/tmp/mini_ab}
cd "$ROOT"; timeout 30 python tools/mini_ab.py prepare $D > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo "$set" | cut -d' ' -f1)
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$name" -- python "$ROOT/tools/mini_ab.py" run $D pmc $WL:$PSPP > "$OUT/pmc_$name.log" 2>&1
done
[ -x "$ROOT/tools/scratch_calib" ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$ROOT/tools/scratch_calib.hip" -o "$ROOT/tools/scratch_calib" > "$OUT/calib_build.log" 2>&1
if [ -x "$ROOT/tools/scratch_calib" ]; then
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 60 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/calib_$set" -- "$ROOT/tools/scratch_calib" 64 > "$OUT/calib_$set.log" 2>&1
  done
fi
cd "$ROOT"; python tools/summarize_pmc.py "$OUT" "$TAG" "$WL" "$PSPP"
