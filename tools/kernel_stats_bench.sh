#!/bin/bash
# rocprofv3 --kernel-trace --stats around the default bench.py command (the launches bench.py times with HIP events): the summary that
# profiles/ carries beside the bench line.   gpurun -- 'bash tools/kernel_stats_bench.sh <tag> [bench args]'
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/kstats_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_profiled.log" 2>&1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${TAG}_kernel_stats.csv" && head -5 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-160
tail -1 "$OUT/bench_profiled.log" | cut -c1-300
