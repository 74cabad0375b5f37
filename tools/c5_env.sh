#!/bin/bash
# C5 stand-in at full detail, default library, under different environment switches: gpurun -- 'bash tools/c5_env.sh <spp> "LABEL=VAR=val ..." ...'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; SPP=${1:-32}; shift
[ -f /tmp/c5/tr15_like.json ] || python - <<PY
import sys
sys.path.insert(0, ".")
from tray_rust_amd import scenes
scenes.write_tr15_like_assets("/tmp/c5", film=(1920, 1080, $SPP))
PY
cat > /tmp/c5_run.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import tray_rust_amd as T
scene, rt, spp, fi = T.Scene.load_file("/tmp/c5/tr15_like.json")
FR = int(os.environ.get("C5_FRAME", "330"))
fi = T.FrameInfo(fi.frames, fi.time, FR, FR)
hip = T.Hip(0, seed=1)
for rep in range(2):
    rt.clear()
    sys.stdout = open(os.devnull, "w")
    hip.render(scene, rt, T.Config("/tmp/c5", "tr15_like", $SPP, 1, fi, (0, 0)))
    sys.stdout = sys.__stdout__
    t = hip.last_timing
print(f"{os.environ.get('LABEL', 'default'):14s} tr15_like full detail frame {FR} 1080p $SPP spp: {t.samples / t.render_ms / 1e3:7.2f} Msamples/s  {t.render_ms:.1f} ms  launches {t.launches}  V {t.vertices / t.samples:.4f}", flush=True)
PY
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}
  env LABEL=$label $rest timeout 200 python /tmp/c5_run.py 2>&1 | grep "full detail"
done
