"""Which camera samples make two builds of the library render different images? (torch-free; one process per library because
TRAYHIP_LIB is read at import)
    python tools/diff_builds.py prepare <dir> [scene] [spp]
    TRAYHIP_LIB=<A> python tools/diff_builds.py render <dir> a
    TRAYHIP_LIB=<B> python tools/diff_builds.py render <dir> b
    TRAYHIP_LIB=<A> python tools/diff_builds.py samples <dir> a      # per-sample radiance of every sample of the differing pixels
    TRAYHIP_LIB=<B> python tools/diff_builds.py samples <dir> b
    python tools/diff_builds.py report <dir>                         # differing samples, both values, the oracle's value
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes

W, H = 1920, 1080
mode, d = sys.argv[1], sys.argv[2]
meta_path = os.path.join(d, "diff_meta.json")
if mode == "prepare":
    name = sys.argv[3] if len(sys.argv) > 3 else "cornell_box"
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    os.makedirs(d, exist_ok=True)
    scenes.write_assets(d, cornell=(W, H, spp), small=(W, H, spp))
    json.dump({"scene": name, "spp": spp}, open(meta_path, "w"))
    sys.exit(0)
meta = json.load(open(meta_path))
name, spp = meta["scene"], meta["spp"]
scene, rt, _, fi = T.Scene.load_file(os.path.join(d, name + ".json"))
if mode == "render":
    tag = sys.argv[3]
    hip = T.Hip(device=0, seed=1)
    rt.clear()
    hip.render(scene, rt, T.Config(d, name, spp, 1, fi, (0, 0)))
    np.save(os.path.join(d, f"img_{tag}.npy"), rt.get_renderf32().reshape(H, W, 4))
    print(tag, "rendered", hip.last_timing.samples, "samples", hip.last_timing.vertices, "vertices", hip.last_timing.rays, "rays")
elif mode == "samples":
    tag = sys.argv[3]
    a, b = np.load(os.path.join(d, "img_a.npy")), np.load(os.path.join(d, "img_b.npy"))
    diff = np.abs(a - b).max(axis=2)
    ys, xs = np.nonzero(diff > 1e-5)   # atomics reorder sums at the 1e-7 level; a moved sample shows at 1e-3 and above
    print(f"{len(xs)} pixels differ by more than 1e-5 (max {diff.max():.3e})")
    if len(xs) > 4000:
        keep = np.argsort(-diff[ys, xs])[:4000]
        ys, xs = ys[keep], xs[keep]
    px = np.repeat(xs.astype(np.uint32), spp); py = np.repeat(ys.astype(np.uint32), spp)
    si = np.tile(np.arange(spp, dtype=np.uint32), len(xs))
    out = np.zeros((len(px), 8), np.float32)
    if len(px):
        dev = scene.device_scene(0, 0)
        T.check(T.lib().tray_debug_sample_radiance(dev, len(px), px.ctypes.data, py.ctypes.data, si.ctypes.data, spp, 1, out.ctypes.data))
    np.savez(os.path.join(d, f"samples_{tag}.npz"), px=px, py=py, si=si, out=out)
else:
    import _oracle as O
    sa, sb = np.load(os.path.join(d, "samples_a.npz")), np.load(os.path.join(d, "samples_b.npz"))
    assert (sa["px"] == sb["px"]).all()
    oa, ob = sa["out"], sb["out"]
    differ = np.nonzero((oa[:, :3] != ob[:, :3]).any(axis=1) | (oa[:, 5:7] != ob[:, 5:7]).any(axis=1))[0]
    print(f"{len(differ)} of {len(oa)} samples of the differing pixels differ between the builds (debug kernel)")
    flat = scene.flatten(0)
    if len(differ):
        sel = differ[:40]
        oc = O.sample_radiance(flat, sa["px"][sel], sa["py"][sel], sa["si"][sel], spp, seed=1)
        for k, i in enumerate(sel):
            print(f"px ({sa['px'][i]},{sa['py'][i]}) s {sa['si'][i]}: A rgb {oa[i, :3]} v/r {oa[i, 5]:.0f}/{oa[i, 6]:.0f} | B rgb {ob[i, :3]} v/r {ob[i, 5]:.0f}/{ob[i, 6]:.0f} | oracle rgb {oc[k, :3]} v/r {oc[k, 5]:.0f}/{oc[k, 6]:.0f}")
