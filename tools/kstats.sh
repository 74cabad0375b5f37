#!/bin/bash
# per-kernel time of one bench_small launch: tools/kstats.sh <tag> <workload> <spp>
set -u
TAG=${1:-x}; WL=${2:-cornell_box}; PSPP=${3:-64}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/kstats_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python "$ROOT/tools/bench_small.py" "$PSPP" 1 "$WL" > "$OUT/run.log" 2>&1
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 "$f" | sed 's/(tr::DevScene.*)"/"/; s/(.*)"/"/' | head -14
