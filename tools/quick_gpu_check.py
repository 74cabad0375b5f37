"""Torch-free GPU check that fits in well under a minute: the smoke render (cornell_box 64x64x16) and a small mesh + MERL scene
through the C ABI, each against the oracle (pixel RMSE < 1e-4), plus exact mesh-hit normals from tray_debug_intersect."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
t0 = time.time()
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O


def rgb(img):
    return img[..., :3] / np.maximum(img[..., 3:], 1e-20)


d = tempfile.mkdtemp()
scenes.write_assets(d, cornell=(64, 64, 16), small=(64, 64, 16))
scenes.write_dragon_assets(d, film=(64, 48, 8), grid=16, extent=1.0)
hip = T.Hip(device=0, seed=7)
for name in ("cornell_box", "dragon"):
    scene, rt, spp, fi = T.Scene.load_file(os.path.join(d, name + ".json"))
    hip.render(scene, rt, T.Config(d, name, spp, 1, fi, (0, 0)))
    gpu = rt.get_renderf32().reshape(rt.height, rt.width, 4)
    cpu, _ = O.render_tiles(scene.flatten(0), spp, seed=7)
    r = float(np.sqrt(np.mean((rgb(gpu) - rgb(cpu)) ** 2)))
    print(f"{name}: RMSE vs oracle {r:.3e} ({time.time() - t0:.1f}s)", flush=True)
    assert r < 1e-4
    if name == "dragon":
        flat = scene.flatten(0)
        rays = O.camera_rays(flat, np.random.default_rng(1).uniform(0, [64, 48], (20000, 2)))
        a = O.intersect(flat, rays)
        b = np.zeros(len(rays), dtype=O.HIT_DTYPE)
        T.check(T.lib().tray_debug_intersect(scene.device_scene(0, 0), len(rays), rays.ctypes.data, b.ctypes.data))
        mesh = a["inst"] == 6
        exact = bool((a["n"][mesh] == b["n"][mesh]).all() and (a["t"] == b["t"]).all() and (a["inst"] == b["inst"]).all())
        print(f"dragon: {int(mesh.sum())} mesh hits, normals bit-identical to the oracle: {exact}", flush=True)
        assert exact
print("QUICK_GPU_CHECK_OK", flush=True)
