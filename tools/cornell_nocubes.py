"""Experiment: cornell_box with and without its two cube meshes -> path vertices per second (is the 12-triangle leaf the cost?)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tray_rust_amd as T
from tray_rust_amd import scenes
d = "/tmp/sc_nc"
scenes.write_assets(d)
for label, drop in (("with cubes", False), ("without cubes", True), ("cubes -> spheres", "sphere")):
    desc = scenes.cornell_box(1920, 1080, 64)
    if drop is True:
        desc["objects"] = desc["objects"][:2]
    elif drop == "sphere":
        for o in desc["objects"][2:]:
            o["geometry"] = {"type": "sphere", "radius": 1.0}
    scene, rt, spp, fi = T.Scene.load_string(json.dumps(desc), d)
    hip = T.Hip(0, seed=1)
    buf = torch.zeros(1080 * 1920 * 4, dtype=torch.float32, device="cuda")
    for rep in range(2):
        hip.render_device(scene, 0, (0, 0), spp, buf.data_ptr())
        tim = hip.timing(scene)
    print(f"{label}: {tim.render_ms:.1f} ms  {tim.samples / tim.render_ms / 1e3:.1f} Msamples/s  V {tim.vertices / tim.samples:.3f}  "
          f"{tim.vertices / tim.render_ms / 1e6:.3f} Gvertices/s  {tim.rays / tim.render_ms / 1e6:.3f} Grays/s", flush=True)
