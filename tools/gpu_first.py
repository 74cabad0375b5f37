"""First GPU contact: per-ray, per-sample and per-image parity against the oracle + a timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes, _lib as L
import _oracle as O

d = "/tmp/sc_gpu"
W, H, SPP = 128, 96, 64
scenes.write_assets(d, cornell=(W, H, SPP), small=(W, H, SPP))
lib = T.lib()
for name in ("cornell_box", "smallpt"):
    scene, rt, spp, fi = T.Scene.load_file(f"{d}/{name}.json")
    flat = scene.flatten(0)
    hip = T.Hip(0, seed=3)
    dev = scene.device_scene(0, 0)
    # --- rays
    rng = np.random.default_rng(0)
    xy = rng.uniform(0, [W, H], size=(20000, 2)).astype(np.float32)
    rays = O.camera_rays(flat, xy)
    ho = O.intersect(flat, rays)
    hg = np.zeros(len(rays), dtype=O.HIT_DTYPE)
    T.check(lib.tray_debug_intersect(dev, len(rays), rays.ctypes.data, hg.ctypes.data))
    same_inst = (ho["inst"] == hg["inst"]).mean()
    hit = ho["inst"] != 0xffffffff
    dt = np.abs(ho["t"][hit] - hg["t"][hit]) / np.maximum(ho["t"][hit], 1e-9)
    print(name, "primary rays: inst agreement", same_inst, "max rel dt", dt.max() if hit.any() else 0,
          "bitexact t", (ho["t"][hit] == hg["t"][hit]).mean(), "max |dn|", np.abs(ho["n"][hit] - hg["n"][hit]).max(),
          "max|dp|", np.abs(ho["p"][hit]-hg["p"][hit]).max())
    # --- samples
    n = 40000
    px = rng.integers(0, W, n).astype(np.uint32); py = rng.integers(0, H, n).astype(np.uint32); si = rng.integers(0, SPP, n).astype(np.uint32)
    ro = O.sample_radiance(flat, px, py, si, SPP, seed=3)
    rg = np.zeros((n, 8), dtype=np.float32)
    T.check(lib.tray_debug_sample_radiance(dev, n, px.ctypes.data, py.ctypes.data, si.ctypes.data, SPP, 3, rg.ctypes.data))
    dr = np.abs(ro[:, :3] - rg[:, :3]).max(axis=1)
    print(name, "samples: pos equal", (ro[:, 3:5] == rg[:, 3:5]).all(), "vertex count equal", (ro[:, 5] == rg[:, 5]).mean(),
          "rays equal", (ro[:, 6] == rg[:, 6]).mean(), "bitexact rgb", (dr == 0).mean(), "max diff", dr.max(),
          "n diff>1e-4", (dr > 1e-4).sum(), "n diff>1e-2", (dr > 1e-2).sum())
    bad = np.argsort(-dr)[:5]
    for b in bad:
        print("   worst", px[b], py[b], si[b], ro[b], rg[b])
    # --- image
    t0 = time.time()
    cfg = T.Config(d, name, spp, 1, fi, (0, 0))
    hip.render(scene, rt, cfg)
    t1 = time.time()
    tim = hip.last_timing
    print(name, "gpu render wall", t1 - t0, "kernel ms", tim.render_ms, "samples", tim.samples, "V", tim.vertices / max(tim.samples, 1),
          "rays/sample", tim.rays / max(tim.samples, 1), "Msamples/s", tim.samples / tim.render_ms / 1e3)
    gpu = rt.get_renderf32().reshape(H, W, 4)
    cpu, st = O.render_tiles(flat, spp, seed=3)
    a = gpu[..., :3] / np.maximum(gpu[..., 3:], 1e-20); b = cpu[..., :3] / np.maximum(cpu[..., 3:], 1e-20)
    print(name, "image RMSE", np.sqrt(np.mean((a - b) ** 2)), "max", np.abs(a - b).max(), "weight max diff", np.abs(gpu[..., 3] - cpu[..., 3]).max(),
          "oracle V", st.vertices / st.samples, "oracle s", st.seconds)
    scene.close()
# bigger timing run
scenes.write_assets(d, cornell=(1920, 1080, 64))
scene, rt, spp, fi = T.Scene.load_file(f"{d}/cornell_box.json")
hip = T.Hip(0, seed=1)
import torch
buf = torch.zeros(1080 * 1920 * 4, dtype=torch.float32, device="cuda")
for rep in range(2):
    hip.render_device(scene, 0, (0, 0), 64, buf.data_ptr())
    tim = hip.timing(scene)
    print("1080p 64spp: kernel ms", tim.render_ms, "Msamples/s", tim.samples / tim.render_ms / 1e3, "V", tim.vertices / tim.samples)
