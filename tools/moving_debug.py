import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O
d = "/tmp/mvdbg"
scene, rt, spp, fi = T.Scene.load_file(scenes.write_moving_box(d, width=160, height=120, samples=32))
rng = np.random.default_rng(1)
n = 200000
px = rng.integers(0, 160, n).astype(np.uint32); py = rng.integers(0, 120, n).astype(np.uint32); si = rng.integers(0, 32, n).astype(np.uint32)
for frame in (0, 5):
    flat = scene.flatten(frame)
    a = O.sample_radiance(flat, px, py, si, 32, seed=4)
    dev = scene.device_scene(frame, 0)
    b = np.zeros((n, 8), np.float32)
    T.check(T.lib().tray_debug_sample_radiance(dev, n, px.ctypes.data, py.ctypes.data, si.ctypes.data, 32, 4, b.ctypes.data))
    dd = np.abs(a[:, :3] - b[:, :3]).max(axis=1)
    print("frame", frame, "pos equal", (a[:, 3:5] == b[:, 3:5]).all(), "vertex count equal", (a[:, 5] == b[:, 5]).mean(),
          "frac>1e-3", (dd > 1e-3).mean(), "frac>1e-5", (dd > 1e-5).mean(), "median", np.median(dd), "mean signed", (a[:, :3] - b[:, :3]).mean(axis=0))
    bad = np.argsort(-dd)[:8]
    for k in bad:
        print("   ", px[k], py[k], si[k], a[k, :3], b[k, :3], "V", a[k, 5], b[k, 5])
