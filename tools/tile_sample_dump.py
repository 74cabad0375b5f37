"""The TILE KERNEL's own per-sample radiance against the oracle (torch-free), for builds made with -DTR_SAMPLE_DUMP (tools/variant.sh): k_path_tiles
writes the unclamped radiance and the vertex count of every camera sample it finishes into a [pixel][sample] buffer, which the library copies
into the file TRAYHIP_SAMPLE_DUMP names. The GPU suite's per-sample bit checks go through k_debug_sample_radiance -- one thread per sample, another
kernel --; this is the kernel the bench runs.
    TRAYHIP_LIB=<a -DTR_SAMPLE_DUMP build> python tools/tile_sample_dump.py [scene] [WxHxspp] [label]
Prints the share of samples whose radiance / vertex count are the oracle's bit for bit, and for the first differing samples both values."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
w, h, spp = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "160x120x32").split("x"))
label = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.environ.get("TRAYHIP_LIB", "libtrayhip.so"))
d = tempfile.mkdtemp()
scenes.write_assets(d, cornell=(w, h, spp), small=(w, h, spp))
if name == "dragon": scenes.write_dragon_assets(d, film=(w, h, spp), grid=220, extent=0.2)
scene, rt, _, fi = T.Scene.load_file(os.path.join(d, name + ".json"))
dump = os.path.join(d, "dump.bin")
os.environ["TRAYHIP_SAMPLE_DUMP"] = dump
hip = T.Hip(0, seed=3)
rt.clear()
sys.stdout = open(os.devnull, "w")
hip.render(scene, rt, T.Config(d, name, spp, 1, fi, (0, 0)))
sys.stdout = sys.__stdout__
if not os.path.exists(dump):
    raise SystemExit(f"{label}: no dump written -- the library was not built with -DTR_SAMPLE_DUMP")
both = np.fromfile(dump, np.float32).reshape(h, w, spp, 2, 4)
got, thr = both[..., 0, :], both[..., 1, :]
flat = scene.flatten(0)
yy, xx, ss = np.meshgrid(np.arange(h, dtype=np.uint32), np.arange(w, dtype=np.uint32), np.arange(spp, dtype=np.uint32), indexing="ij")
ref = O.sample_radiance(flat, xx.ravel(), yy.ravel(), ss.ravel(), spp, seed=3).reshape(h, w, spp, -1)
same_v = got[..., 3] == ref[..., 5]
# the oracle returns the CLAMPED sample (multithreaded.rs:98-99); compare after the same clamp
same_rgb = (np.clip(got[..., :3], 0, 1).view(np.uint32) == np.clip(ref[..., :3], 0, 1).view(np.uint32)).all(axis=-1)
n = same_v.size
print(f"{label} {name} {w}x{h}x{spp}: {n} camera samples of the TILE KERNEL: {100 * same_rgb.mean():.4f} % bit-identical radiance ({int((~same_rgb).sum())} differ), "
      f"{int((~same_v).sum())} with another vertex count", flush=True)
bad = np.argwhere(~same_rgb | ~same_v)
for y, x, s_ in bad[:12]:
    print(f"   px ({x},{y}) s {s_}: tile kernel {got[y, x, s_, :3]} v {got[y, x, s_, 3]:.0f} | oracle {ref[y, x, s_, :3]} v {ref[y, x, s_, 5]:.0f}")
if os.environ.get("DUMP_SAVE"):   # a second build's run compares its throughputs with the saved ones (the oracle does not return them)
    np.save(os.environ["DUMP_SAVE"], both)
if os.environ.get("DUMP_COMPARE") and os.path.exists(os.environ["DUMP_COMPARE"]):
    other = np.load(os.environ["DUMP_COMPARE"])
    d_il = (other[..., 0, :3].view(np.uint32) != both[..., 0, :3].view(np.uint32)).any(axis=-1)
    d_th = (other[..., 1, :3].view(np.uint32) != both[..., 1, :3].view(np.uint32)).any(axis=-1)
    print(f"   against the saved build: radiance differs in {int(d_il.sum())} samples, final throughput in {int(d_th.sum())}; both {int((d_il & d_th).sum())}, "
          f"radiance only {int((d_il & ~d_th).sum())}, throughput only {int((~d_il & d_th).sum())}")
if len(bad):
    v = ref[..., 5][~same_rgb | ~same_v]
    print("   vertex counts (oracle) of the differing samples:", np.bincount(v.astype(int)).tolist())
