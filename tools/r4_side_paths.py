"""Rates of the paths beside the hot ones, at 1920x1080 on one GPU (torch-free, the library's own HIP events):
   python tools/r4_side_paths.py [dir]
Uniform / Adaptive samplers (k_sampler_pass), the AnimatedMesh scene (k_sampler_pass<3>), Whitted (the tile kernel's Whitted instantiation) and a
textured scene (FEAT = all + textures)."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tray_rust_amd as T
from tray_rust_amd import scenes

W, H = 1920, 1080
d = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp()


def rate(label, scene, rt, fi, spp, make_sampler=None, frame=None):
    hip = T.Hip(device=0, seed=1, sampler=make_sampler)
    if frame is not None:
        fi = T.FrameInfo(fi.frames, fi.time, frame, frame)
    best, t = 0.0, None
    for rep in range(2):
        rt.clear()
        sys.stdout = open(os.devnull, "w")
        try:
            hip.render(scene, rt, T.Config(d, "s", spp, 1, fi, (0, 0)))
        finally:
            sys.stdout = sys.__stdout__
        t = hip.last_timing
        best = max(best, t.samples / (t.render_ms * 1e-3) / 1e6)
    print(f"{label:44s} {t.samples / (W * H):7.2f} samples per pixel  {best:8.1f} Msamples/s  ({t.render_ms:.1f} ms, {t.launches} launches)", flush=True)


scenes.write_assets(d, cornell=(W, H, 64), small=(W, H, 64))
scene, rt, _, fi = T.Scene.load_file(os.path.join(d, "cornell_box.json"))
rate("cornell_box LowDiscrepancy 64 spp (tile kernel)", scene, rt, fi, 64)
rate("cornell_box Uniform", scene, rt, fi, 1, lambda dim, spp: T.sampler.Uniform(dim))
rate("cornell_box Adaptive(4, 32)", scene, rt, fi, 1, lambda dim, spp: T.sampler.Adaptive(dim, 4, 32))
rate("cornell_box Adaptive(16, 64)", scene, rt, fi, 1, lambda dim, spp: T.sampler.Adaptive(dim, 16, 64))
scene, rt, _, fi = T.Scene.load_file(scenes.write_waving_flag(d, grid=48, n_keys=4, width=W, height=H, samples=16))
rate("waving_flag (AnimatedMesh) 16 spp", scene, rt, fi, 16, frame=1)
doc = json.load(open(os.path.join(d, "smallpt.json")))
doc["integrator"]["type"] = "whitted"
json.dump(doc, open(os.path.join(d, "smallpt_whitted.json"), "w"))
scene, rt, _, fi = T.Scene.load_file(os.path.join(d, "smallpt_whitted.json"))
rate("smallpt Whitted 64 spp", scene, rt, fi, 64)
scene, rt, _, fi = T.Scene.load_file(scenes.write_textured_box(d, width=W, height=H, samples=64))
rate("textured_box 64 spp", scene, rt, fi, 64)
