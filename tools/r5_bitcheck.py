"""Torch-free: per camera sample, the GPU (tray_debug_sample_radiance) against the oracle on the cut-down workloads of tools/mini_ab.py:
share of samples whose radiance is the oracle's bit for bit, share that took another path, per-sample RMSE.
    python tools/r5_bitcheck.py [dir] [n]          (TRAYHIP_LIB selects the build)
Round 5: with glibc's libm restated on the device (dev_libm.h) every sample should be bit-identical; rounds 1-4 (ocml) had 77 - 84 %."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tray_rust_amd as T
import _oracle as O

d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/mini_ab"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
W, H = 1920, 1080
if not os.path.exists(os.path.join(d, "cornell_box.json")):
    os.system(f"python {ROOT}/tools/mini_ab.py prepare {d} > /dev/null 2>&1")
for name, frame, spp in (("cornell_box", 0, 64), ("smallpt", 0, 64), ("dragon", 0, 32), ("moving_box", 3, 32), ("tr15_like", 330, 16), ("tr15_like", 127, 16)):
    t0 = time.time()
    scene, rt, _, fi = T.Scene.load_file(os.path.join(d, "tr15" if name == "tr15_like" else "", name + ".json"))
    flat = scene.flatten(frame)
    rng = np.random.default_rng(5)
    px = rng.integers(0, W, n).astype(np.uint32); py = rng.integers(0, H, n).astype(np.uint32); si = rng.integers(0, spp, n).astype(np.uint32)
    a = O.sample_radiance(flat, px, py, si, spp, seed=3)
    b = np.zeros((n, 8), np.float32)
    T.check(T.lib().tray_debug_sample_radiance(scene.device_scene(frame, 0), n, px.ctypes.data, py.ctypes.data, si.ctypes.data, spp, 3, b.ctypes.data))
    flipped = (a[:, 5] != b[:, 5]) | (a[:, 6] != b[:, 6])
    same = (a[:, :3] == b[:, :3]).all(axis=1)
    se = ((np.clip(a[:, :3], 0, 1) - np.clip(b[:, :3], 0, 1)) ** 2).sum(axis=1)
    print(f"{name:12s} frame {frame:3d}: {n} samples, {100 * same.mean():.3f} % bit-identical radiance ({int((~same).sum())} differ), {int(flipped.sum())} on another path, "
          f"per-sample RMSE {np.sqrt(se.mean() / 3):.3e}, positions equal {bool((a[:, 3:5] == b[:, 3:5]).all())}  ({time.time() - t0:.1f}s)", flush=True)
