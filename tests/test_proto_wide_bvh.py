"""Design prototype for the next device traversal (oracle/proto_wide_bvh.hpp, CPU only): a 4-wide collapse of the reference's
binary BVH<Triangle> with a pop-time re-test returns the SAME hit records as the binary traversal, with fewer dependent node
fetches. Nothing in the product uses it yet (Next / C5)."""
import ctypes as C

import numpy as np
import pytest

import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O


def both(flat, rays, qbits=0):
    o = O.oracle()
    o.oracle_proto_wide_bvh.restype = C.c_int
    o.oracle_proto_wide_bvh.argtypes = [C.POINTER(T._lib.TrayFlatScene), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    a = np.zeros(len(rays), dtype=O.HIT_DTYPE); b = np.zeros(len(rays), dtype=O.HIT_DTYPE)
    cnt = np.zeros(4, np.uint64)
    assert o.oracle_proto_wide_bvh(flat, len(rays), rays.ctypes.data, a.ctypes.data, b.ctypes.data, cnt.ctypes.data, qbits) == 0
    return a, b, cnt


@pytest.mark.parametrize("grid", [24, 120])
def test_wide_collapse_returns_the_same_hits(grid, tmp_path, built):
    p, _ = scenes.write_dragon_assets(str(tmp_path), film=(160, 120, 4), grid=grid, extent=1.0)
    scene, *_ = T.Scene.load_file(p)
    flat = scene.flatten(0)
    rng = np.random.default_rng(grid)
    rays = O.camera_rays(flat, rng.uniform(0, [160, 120], (40000, 2)))
    n = 40000   # rays from inside the box towards the mesh: deep traversals, grazing hits, shadow-like segments
    o = rng.uniform([-14, 1, -18], [14, 23, 19], (n, 3)); tgt = rng.normal([8.5, 3.7, 1.5], 4.0, (n, 3)); d = tgt - o
    seg = rng.uniform(0, 1, n) < 0.5
    d[~seg] /= np.linalg.norm(d[~seg], axis=1, keepdims=True)
    inner = np.concatenate([o, d, np.full((n, 1), 0.001), np.where(seg, 0.999, np.inf)[:, None], np.zeros((n, 1))], axis=1).astype(np.float32)
    rays = np.concatenate([rays, inner])
    a, b, cnt = both(flat, rays)
    ref = O.intersect(flat, rays)
    assert a.tobytes() == ref.tobytes()            # the counting binary traversal is the oracle's
    assert a.tobytes() == b.tobytes()              # 4-wide collapse: identical records, bit for bit
    assert (a["inst"] == 6).mean() > 0.05
    assert cnt[2] == cnt[3]                        # the same leaves are visited
    ratio = float(cnt[1]) / float(cnt[0])
    print(f"grid {grid}: dependent fetches per ray binary {cnt[0] / len(rays):.2f} wide {cnt[1] / len(rays):.2f} (x{ratio:.2f})")
    assert ratio < 0.75
    # quantised slot boxes (conservative): no candidate is lost; how many extra fetches, how many records change
    for qbits in (16, 8):
        a2, q, cq = both(flat, rays, qbits)
        differ = int((a2["inst"] != q["inst"]).sum() + ((a2["inst"] == q["inst"]) & ((a2["prim"] != q["prim"]) | (a2["t"] != q["t"]))).sum())
        extra = float(cq[1]) / float(cnt[1]) - 1.0
        print(f"   {qbits}-bit boxes: +{100 * extra:.1f} % fetches, {differ} of {len(rays)} hit records differ")
        assert differ <= 2 and extra < (0.02 if qbits == 16 else 0.5)
        hit = a2["inst"] != 0xffffffff
        assert (q["t"][hit] <= a2["t"][hit]).all() or differ > 0   # a conservative box can only add candidates


def packed_nodes(flat, quantised):
    """The product's packed wide nodes of every mesh (tray_debug_wide_nodes), concatenated the way the library uploads them."""
    L = T._lib
    lib = L.lib()
    fs = flat.contents
    words, first, roots = [], [], []
    total = 0
    for m in range(fs.n_meshes):
        n = C.c_uint64(0); root = C.c_uint32(0)
        L.check(lib.tray_debug_wide_nodes(C.cast(flat, C.c_void_p), m, quantised, None, 0, C.byref(n), C.byref(root)))
        buf = np.zeros(max(int(n.value), 1), np.uint32)
        L.check(lib.tray_debug_wide_nodes(C.cast(flat, C.c_void_p), m, quantised, buf.ctypes.data, int(n.value), C.byref(n), C.byref(root)))
        first.append(total); roots.append(root.value); words.append(buf[:int(n.value)]); total += int(n.value)
    return np.concatenate(words) if total else np.zeros(1, np.uint32), np.array(first, np.uint64), np.array(roots, np.uint32)


def packed_both(flat, rays, quantised):
    o = O.oracle()
    o.oracle_proto_packed_wide.restype = C.c_int
    o.oracle_proto_packed_wide.argtypes = [C.POINTER(T._lib.TrayFlatScene), C.c_uint32] + [C.c_void_p] * 7 + [C.c_int]
    words, first, roots = packed_nodes(flat, quantised)
    a = np.zeros(len(rays), dtype=O.HIT_DTYPE); b = np.zeros(len(rays), dtype=O.HIT_DTYPE)
    cnt = np.zeros(4, np.uint64)
    assert o.oracle_proto_packed_wide(flat, len(rays), rays.ctypes.data, a.ctypes.data, b.ctypes.data, cnt.ctypes.data,
                                      words.ctypes.data, first.ctypes.data, roots.ctypes.data, quantised) == 0
    return a, b, cnt, words


def mixed_rays(flat, rng, n, width, height, lo, hi, centre, spread):
    cam = O.camera_rays(flat, rng.uniform(0, [width, height], (n, 2)))
    o = rng.uniform(lo, hi, (n, 3)); d = rng.normal(centre, spread, (n, 3)) - o
    seg = rng.uniform(0, 1, n) < 0.5
    d[~seg] /= np.linalg.norm(d[~seg], axis=1, keepdims=True)
    inner = np.concatenate([o, d, np.full((n, 1), 0.001), np.where(seg, 0.999, np.inf)[:, None], np.zeros((n, 1))], axis=1).astype(np.float32)
    return np.concatenate([cam, inner])


@pytest.mark.parametrize("quantised", [0, 1])
def test_product_packed_nodes_walked_like_the_device(quantised, tmp_path, built):
    """What the device will read (the library's own node buffer, both formats), walked with the device kernel's node step on
    the CPU: exact boxes give the binary traversal's records bit for bit; 8-bit boxes visit a superset and change (almost) none."""
    p, _ = scenes.write_dragon_assets(str(tmp_path), film=(160, 120, 4), grid=96, extent=1.0)
    scene, *_ = T.Scene.load_file(p)
    flat = scene.flatten(0)
    rays = mixed_rays(flat, np.random.default_rng(5), 30000, 160, 120, [-14, 1, -18], [14, 23, 19], [8.5, 3.7, 1.5], 4.0)
    a, b, cnt, words = packed_both(flat, rays, quantised)
    assert a.tobytes() == O.intersect(flat, rays).tobytes()
    assert (a["inst"] == 6).mean() > 0.05
    differ = int((a["inst"] != b["inst"]).sum() + ((a["inst"] == b["inst"]) & ((a["prim"] != b["prim"]) | (a["t"] != b["t"]))).sum())
    per_ray = cnt[:2] / len(rays)
    print(f"packed ({'8-bit' if quantised else 'exact'}, {words.nbytes / 1e6:.1f} MB): fetches per ray binary {per_ray[0]:.2f} packed {per_ray[1]:.2f}; {differ} records differ")
    if quantised:
        assert differ <= 2 and cnt[3] >= cnt[2]     # a superset of the leaves
    else:
        assert a.tobytes() == b.tobytes() and cnt[2] == cnt[3]
    assert per_ray[1] < 0.75 * per_ray[0]


def test_quantised_boxes_enclose_the_exact_ones(tmp_path, built):
    """Every dequantised slot box of the 64-B format contains the slot box of the 128-B format (same collapse, same order)."""
    p, _ = scenes.write_dragon_assets(str(tmp_path), film=(160, 120, 4), grid=48, extent=0.2)   # small extent: |lo| >> box size
    scene, *_ = T.Scene.load_file(p)
    flat = scene.flatten(0)
    exact, fe, re_ = packed_nodes(flat, 0)
    quant, fq, rq = packed_nodes(flat, 1)
    assert len(exact) // 32 == len(quant) // 16 and (re_ == rq).all()
    e = exact.reshape(-1, 32); q = quant.reshape(-1, 16)
    assert (e[:, 24:28] == q[:, 12:16]).all()                                   # same references
    meta = (q[:, 3] & 3) | ((q[:, 4] & 3) << 2) | ((q[:, 5] & 3) << 4)
    assert (meta == e[:, 28]).all()                                             # split axes ride in the scales' low bits
    lo = q[:, 0:3].copy().view(np.float32); sc = q[:, 3:6].copy().view(np.float32)
    used = e[:, 24:28] != 0xffffffff
    for k in range(3):
        for s in range(4):
            qmin = ((q[:, 6 + k] >> (8 * s)) & 0xff).astype(np.float32); qmax = ((q[:, 9 + k] >> (8 * s)) & 0xff).astype(np.float32)
            dmin = lo[:, k] + (qmin * sc[:, k]).astype(np.float32); dmax = lo[:, k] + (qmax * sc[:, k]).astype(np.float32)
            bmin = e[:, 4 * k + s].copy().view(np.float32); bmax = e[:, 12 + 4 * k + s].copy().view(np.float32)
            u = used[:, s]
            assert (dmin[u] <= bmin[u]).all() and (dmax[u] >= bmax[u]).all()
            assert (bmin[u] - dmin[u] <= 1.01 * sc[u, k] + 1e-6 * np.abs(bmin[u])).all()   # ... and not by more than a step
