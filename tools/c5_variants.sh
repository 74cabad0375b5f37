#!/bin/bash
# C5 stand-in at full detail for the default library and variant builds (tools/variant.sh): gpurun -- 'bash tools/c5_variants.sh <spp> <name>...'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; SPP=${1:-32}; shift; L=$ROOT/tray_rust_amd
python - <<PY
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
fi = T.FrameInfo(fi.frames, fi.time, 330, 330)
hip = T.Hip(0, seed=1)
for rep in range(2):
    rt.clear()
    sys.stdout = open(os.devnull, "w")
    hip.render(scene, rt, T.Config("/tmp/c5", "tr15_like", $SPP, 1, fi, (0, 0)))
    sys.stdout = sys.__stdout__
    t = hip.last_timing
print(f"{os.environ.get('LABEL', 'default'):10s} tr15_like full detail frame 330 1080p $SPP spp: {t.samples / t.render_ms / 1e3:7.2f} Msamples/s  {t.render_ms:.1f} ms  launches {t.launches}", flush=True)
PY
LABEL=default TRAYHIP_STATS=1 timeout 120 python /tmp/c5_run.py 2>&1 | grep -v "^\[trayhip\] \(trace\|vertex\|quer\|regen\|  \)" | tail -4
for v in "$@"; do
  [ -f $L/libtrayhip_$v.so ] && LABEL=$v TRAYHIP_LIB=$L/libtrayhip_$v.so TRAYHIP_STATS=1 timeout 120 python /tmp/c5_run.py 2>&1 | grep "dynamic-fetch\|full detail" | sort | uniq | tail -2
done
