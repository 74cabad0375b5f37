"""Experiment (torch-free): cornell_box with / without its two cube meshes / with spheres in their place -> what the cooperative
12-triangle test costs per path vertex.   gpurun -- 'python tools/cornell_nocubes.py'"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tray_rust_amd as T
from tray_rust_amd import scenes
d = "/tmp/sc_nc"
scenes.write_assets(d)
hip = T.Hip(0, seed=1)
for label, drop in (("with cubes", False), ("without cubes", True), ("cubes -> spheres", "sphere")):
    desc = scenes.cornell_box(1920, 1080, 64)
    if drop is True:
        desc["objects"] = desc["objects"][:2]
    elif drop == "sphere":
        for o in desc["objects"][2:]:
            o["geometry"] = {"type": "sphere", "radius": 1.0}
    scene, rt, spp, fi = T.Scene.load_string(json.dumps(desc), d)
    for rep in range(2):
        rt.clear()
        sys.stdout = open(os.devnull, "w")
        try:
            hip.render(scene, rt, T.Config(d, "cornell_box", spp, 1, fi, (0, 0)))
        finally:
            sys.stdout = sys.__stdout__
        tim = hip.last_timing
    print(f"{label}: {tim.render_ms:.1f} ms  {tim.samples / tim.render_ms / 1e3:.1f} Msamples/s  V {tim.vertices / tim.samples:.3f}  "
          f"{tim.vertices / tim.render_ms / 1e6:.3f} Gvertices/s  {tim.rays / tim.render_ms / 1e6:.3f} Grays/s", flush=True)
