import os, sys, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O
name = sys.argv[1]
d = "/tmp/chk"; os.makedirs(d, exist_ok=True)
scenes.write_assets(d, cornell=(160, 120, 32), small=(160, 120, 32))
scene, rt, spp, fi = T.Scene.load_file(os.path.join(d, name + ".json"))
hip = T.Hip(0, seed=3)
rt.clear(); hip.render(scene, rt, T.Config(d, name, 32, 1, fi, (0, 0)))
gpu = rt.get_renderf32().reshape(120, 160, 4)
cpu, st = O.render_tiles(scene.flatten(0), 32, seed=3)
rgb = lambda i: i[..., :3] / np.maximum(i[..., 3:], 1e-20)
print(name, os.environ.get("TRAYHIP_FEAT_ALL"), os.environ.get("TRAYHIP_MODE"), "RMSE", float(np.sqrt(np.mean((rgb(gpu) - rgb(cpu)) ** 2))), "V", hip.last_timing.vertices, st.vertices, "mean gpu", rgb(gpu).mean(), "cpu", rgb(cpu).mean())
