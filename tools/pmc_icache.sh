#!/bin/bash
# Instruction-cache counters of the tile kernel (one launch, torch-free runner).   gpurun -- 'bash tools/pmc_icache.sh [scene] [spp]'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; WL=${1:-cornell_box}; SPP=${2:-64}; D=/tmp/mini_ab; OUT=$ROOT/gpurun_out/icache; mkdir -p $OUT
cd "$ROOT"; timeout 30 python tools/mini_ab.py prepare $D > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ"; do
  name=$(echo "$set" | cut -d' ' -f1)
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -- python "$ROOT/tools/mini_ab.py" run $D pmc $WL:$SPP > $OUT/$name.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, sys, collections
c = collections.defaultdict(float)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_path_tiles" in r["Kernel_Name"]: c[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(c.items()): print(f"{k:32s} {v:.4g}")
if c.get("SQC_ICACHE_REQ"): print("icache miss rate", c["SQC_ICACHE_MISSES"] / c["SQC_ICACHE_REQ"], "incl duplicates", (c["SQC_ICACHE_MISSES"] + c["SQC_ICACHE_MISSES_DUPLICATE"]) / c["SQC_ICACHE_REQ"])
PY
