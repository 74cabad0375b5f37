#!/bin/bash
# rocprofv3 kernel stats of ONE cornell_box 1920x1080 render under sampler::Adaptive::new(dim, 4, 32): the rounds of k_sampler_pass / k_sampler_decide
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
cat > /tmp/adaptive_run.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import tray_rust_amd as T
from tray_rust_amd import scenes
scenes.write_assets("/tmp/adp", cornell=(1920, 1080, 64), small=(1920, 1080, 64))
scene, rt, spp, fi = T.Scene.load_file("/tmp/adp/cornell_box.json")
hip = T.Hip(0, seed=1, sampler=lambda dim, spp: T.sampler.Adaptive(dim, 4, 32))
for rep in range(2):
    rt.clear(); hip.render(scene, rt, T.Config("/tmp/adp", "c", 1, 1, fi, (0, 0)))
t = hip.last_timing
print(f"Adaptive(4, 32): {t.samples} samples, {t.samples / t.render_ms / 1e3:.1f} Msamples/s, {t.launches} launches")
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_adaptive_kstats -- python /tmp/adaptive_run.py 2>&1 | grep "Adaptive("
cd $ROOT; python tools/kstats_table.py gpurun_out/r04_adaptive_kstats 2>&1 | head -8 | tee gpurun_out/r04_adaptive_kernel_times.txt
