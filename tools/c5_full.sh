#!/bin/bash
# C5 stand-in at FULL detail (59 instances, 3.1 M triangles), frame 330, 1080p at a reduced sample count: per-kernel time of the
# wavefront schedule (rocprofv3 --kernel-trace --stats) with and without the material sort.
#   gpurun --timeout 600 -- 'bash tools/c5_full.sh [spp]'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; SPP=${1:-32}; OUT=$ROOT/gpurun_out/c5_full; mkdir -p "$OUT"
L=$ROOT/tray_rust_amd
python - <<PY
import sys
sys.path.insert(0, ".")
from tray_rust_amd import scenes
scenes.write_tr15_like_assets("/tmp/c5", film=(1920, 1080, $SPP))
PY
cat > /tmp/c5_run.py <<PY
import os, sys, time
sys.path.insert(0, "$ROOT")
import tray_rust_amd as T
scene, rt, spp, fi = T.Scene.load_file("/tmp/c5/tr15_like.json")
fi = T.FrameInfo(fi.frames, fi.time, 330, 330)
hip = T.Hip(0, seed=1)
for rep in range(2):
    rt.clear()
    sys.stdout = open(os.devnull, "w")
    hip.render(scene, rt, T.Config("/tmp/c5", "tr15_like", $SPP, 1, fi, (0, 0)))
    sys.stdout = sys.__stdout__
    t = hip.last_timing
    print(f"{os.environ.get('LABEL', 'default'):10s} tr15_like full detail frame 330 1080p $SPP spp: {t.samples / t.render_ms / 1e3:7.2f} Msamples/s  {t.render_ms:.1f} ms  launches {t.launches}  V {t.vertices / t.samples:.3f}", flush=True)
PY
{
LABEL=default timeout 120 python /tmp/c5_run.py
LABEL=nosort TRAYHIP_WF_SORT=0 timeout 120 python /tmp/c5_run.py
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python /tmp/c5_run.py > "$OUT/prof.log" 2>&1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | sed 's/(tr::DevScene.*)"/"/; s/(.*)"/"/' | head -16
} 2>&1 | tee $OUT/c5_full.log
