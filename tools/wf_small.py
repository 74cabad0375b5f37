import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O
d = "/tmp/sc_wf"; W, H, SPP = 48, 32, 16
scenes.write_assets(d, cornell=(W, H, SPP))
scene, rt, spp, fi = T.Scene.load_file(f"{d}/cornell_box.json")
hip = T.Hip(0, seed=3)
t0 = time.time()
hip.render(scene, rt, T.Config(d, "c", spp, 1, fi, (0, 0)))
print("render wall", time.time() - t0, "launches", hip.last_timing.launches, "samples", hip.last_timing.samples, "vertices", hip.last_timing.vertices, flush=True)
gpu = rt.get_renderf32().reshape(H, W, 4)
cpu, st = O.render_tiles(scene.flatten(0), spp, seed=3)
a = gpu[..., :3] / np.maximum(gpu[..., 3:], 1e-20); b = cpu[..., :3] / np.maximum(cpu[..., 3:], 1e-20)
print("RMSE", np.sqrt(np.mean((a - b) ** 2)), "oracle vertices", st.vertices, "weight diff", np.abs(gpu[..., 3] - cpu[..., 3]).max())
