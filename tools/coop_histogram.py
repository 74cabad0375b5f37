"""How many rays of a wave reach the triangles of a small mesh (the cooperative test of dev_geom.h: mesh_leaf_coop) -- CPU only: the tile kernel in
the SIMT emulation of tests/emu built with -DTR_COOP_HIST, over a few tiles of a 1920x1080 film (the bench's ray coherence).
usage: python tools/coop_histogram.py [spp] [tiles]"""
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
import _emu as E

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = tempfile.mkdtemp(prefix="coophist")
scenes.write_assets(d, cornell=(1920, 1080, spp))
scene, *_ = T.Scene.load_file(os.path.join(d, "cornell_box.json"))
flat = scene.flatten(0)
tiles = np.array(T.BlockQueue((1920, 1080), (8, 8)).blocks, np.uint32).reshape(-1, 2)
tiles = np.ascontiguousarray(tiles[(np.arange(n_tiles) * 2 + 1) * len(tiles) // (2 * n_tiles)])
lib = E.emu(defines=("TR_COOP_HIST",))
lib.emu_coop_hist.restype = C.POINTER(C.c_ulonglong)
img, stats = E.render_tiles(flat, tiles, spp, 1, defines=("TR_COOP_HIST",))
h = np.ctypeslib.as_array(lib.emu_coop_hist(), (65 * 65,)).reshape(65, 65).astype(np.int64)
by_n = h.sum(axis=1)
calls = by_n.sum()
print(f"cornell_box 1920x1080, {spp} spp, {n_tiles} tiles: {stats[0]} samples, {calls} cooperative tests that staged a ray")
print("rays staged n : share of the tests, cumulative, passes of 16 rays today")
cum = 0
for n in range(1, 65):
    if by_n[n]:
        cum += by_n[n]
        print(f"  {n:3d} : {100 * by_n[n] / calls:5.1f} %  {100 * cum / calls:5.1f} %  {-(-n // 16)}")
n = np.arange(65)
print(f"mean n {float((by_n * n).sum() / calls):.2f}; triangle passes today (3 tests each, 12 triangles) per test: {float((by_n * -(-n // 16)).sum() / calls):.3f}")
for name, rule in (("8 lanes per ray up to 8 rays (2 tests a pass), else 4 (3 tests)", lambda k: 2 * -(-k // 8) if k <= 8 else 3 * -(-k // 16)),
                   ("16 lanes up to 4 (1 test), 8 up to 8 (2 tests), else 4 (3 tests)", lambda k: 1 if k <= 4 else 2 if k <= 8 else 3 * -(-k // 16)),
                   ("16 lanes per ray always (1 test per pass of 4 rays)", lambda k: -(-k // 4)),
                   ("best of the three widths per call", lambda k: min(-(-k // 4), 2 * -(-k // 8), 3 * -(-k // 16)))):
    today = float((by_n[1:] * np.array([3 * -(-k // 16) for k in range(1, 65)])).sum())
    new = float((by_n[1:] * np.array([rule(k) for k in range(1, 65)])).sum())
    print(f"  triangle tests per lane, {name}: {new / today:.3f} of today's")
