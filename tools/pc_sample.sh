#!/bin/bash
# PC sampling of ONE torch-free launch of the tile kernel (tools/mini_ab.py) with a -gline-tables-only build of the library:
#   tools/pc_sample.sh <tag> [scene] [spp] [lib] [method: host_trap|stochastic] [interval]
# -> gpurun_out/pcs_<tag>/{list.txt, run.log, pcs_lines.txt, pcs_summary.txt, raw_head.csv}. Every rocprofv3 call sits under its own timeout.
set -u
TAG=${1:-x}; WL=${2:-cornell_box}; PSPP=${3:-64}; LIB=${4:-tray_rust_amd/libtrayhip_g.so}; METHOD=${5:-host_trap}; INTERVAL=${6:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pcs_$TAG; mkdir -p "$OUT"
D=${MINI_AB_DIR:-/tmp/mini_ab}
cd "$ROOT"; [ -f $D/cornell_box.json ] || timeout 60 python tools/mini_ab.py prepare $D > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 60 rocprofv3 -L 2>&1 | grep -i -B2 -A12 "pc.sampl" | head -80 > "$OUT/list.txt"
if [ "$METHOD" = stochastic ]; then UNIT=cycles; INTERVAL=${INTERVAL:-1048576}; else UNIT=time; INTERVAL=${INTERVAL:-1000}; fi
RAW=/tmp/pcs_raw_$TAG; rm -rf $RAW
TRAYHIP_LIB=$ROOT/$LIB timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL \
    --output-format csv -d $RAW -- python "$ROOT/tools/mini_ab.py" run $D pcs $WL:$PSPP > "$OUT/run.log" 2>&1
echo "rocprofv3 exit $?" >> "$OUT/run.log"
find $RAW -name '*.csv' -exec ls -la {} \; >> "$OUT/run.log" 2>&1
f=$(find $RAW -name '*pc_sampling*.csv' ! -name '*stats*' | head -1)
[ -n "$f" ] && head -40 "$f" > "$OUT/raw_head.csv"
cd "$ROOT"; python tools/summarize_pcs.py $RAW "$OUT/pcs" >> "$OUT/run.log" 2>&1
tail -5 "$OUT/run.log"
