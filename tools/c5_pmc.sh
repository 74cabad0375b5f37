#!/bin/bash
# Memory-side counters of the wavefront schedule's kernels on the C5 stand-in at full detail (one rocprofv3 --pmc pass per set).
#   gpurun --timeout 600 -- 'bash tools/c5_pmc.sh <tag> [spp]'   (run tools/c5_variants.sh or c5_full.sh first in the same call: they write /tmp/c5_run.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-x}; OUT=$ROOT/gpurun_out/c5pmc_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  name=$(echo "$set" | cut -d' ' -f1)
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc_$name" -- python /tmp/c5_run.py > "$OUT/pmc_$name.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); ns = collections.defaultdict(float)
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void tr::", "")
        tab[k][row["Counter_Name"]] += float(row["Counter_Value"])
for path in glob.glob(os.path.join(out, "pmc_FETCH_SIZE", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void tr::", "")
        calls[k] += 1; ns[k] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
with open(os.path.join(out, "summary.txt"), "w") as f:
    for k in sorted(tab, key=lambda k: -ns[k]):
        c = tab[k]; n = max(calls[k], 1)
        line = f"{k[:34]:34s} calls {n:5d} avg {ns[k] / n / 1e3:8.1f} us | per call: " + "  ".join(f"{name} {v / n:.4g}" for name, v in sorted(c.items()))
        print(line); f.write(line + "\n")
PY
