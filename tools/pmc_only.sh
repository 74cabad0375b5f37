#!/bin/bash
# PMC passes only (no full bench step): tools/pmc_only.sh <tag> <workload> <spp>
set -u
TAG=${1:-x}; WL=${2:-cornell_box}; PSPP=${3:-64}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
if [ "${PMC_QUICK:-0}" = "1" ]; then
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
    name=$(echo "$set" | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$name" -- python "$ROOT/tools/bench_small.py" "$PSPP" 1 "$WL" > "$OUT/pmc_$name.log" 2>&1
  done
  python "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG" "$WL"
  exit 0
fi
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo "$set" | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$name" -- python "$ROOT/tools/bench_small.py" "$PSPP" 1 "$WL" > "$OUT/pmc_$name.log" 2>&1
done
python "$ROOT/tools/summarize_prof.py" "$OUT" "$TAG" "$WL"
