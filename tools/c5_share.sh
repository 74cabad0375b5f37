#!/bin/bash
# One GPU's share of a C5 frame when its TILES are dealt to N GPUs (4050 of the 32 400 at N = 8): the first <tiles> tiles of the Morton queue of the
# C5 stand-in at full detail, under environment switches.   gpurun -- 'bash tools/c5_share.sh <spp> <tiles> "LABEL=VAR=val ..." ...'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; SPP=${1:-128}; TILES=${2:-4050}; shift; shift
[ -f /tmp/c5s/tr15_like.json ] || python - <<PY
import sys
sys.path.insert(0, ".")
from tray_rust_amd import scenes
scenes.write_tr15_like_assets("/tmp/c5s", film=(1920, 1080, $SPP))
PY
cat > /tmp/c5s_run.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import tray_rust_amd as T
scene, rt, spp, fi = T.Scene.load_file("/tmp/c5s/tr15_like.json")
fi = T.FrameInfo(fi.frames, fi.time, 330, 330)
hip = T.Hip(0, seed=1)
for rep in range(2):
    rt.clear()
    sys.stdout = open(os.devnull, "w")
    hip.render(scene, rt, T.Config("/tmp/c5s", "tr15_like", $SPP, 1, fi, (0, $TILES)))
    sys.stdout = sys.__stdout__
    t = hip.last_timing
print(f"{os.environ.get('LABEL', 'default'):14s} {$TILES} tiles of the C5 stand-in, $SPP spp: {t.samples / t.render_ms / 1e3:7.2f} Msamples/s  {t.render_ms:.1f} ms  launches {t.launches}", flush=True)
PY
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}
  env LABEL=$label $rest timeout 300 python /tmp/c5s_run.py 2>&1 | grep "tiles of"
done
