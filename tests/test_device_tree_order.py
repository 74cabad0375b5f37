"""The device order of the BVHs (tray_rust_amd/csrc/host/gates.hpp: pair_tree) and the wavefront traversal's instance records
(wf_inst_records), read back through the host emulation library and checked against the reference-order arrays of the flat scene:
the paired tree must be the SAME tree -- walking it first child / second child reproduces the reference's preorder array (bvh.rs:278-295)
node for node --, siblings must be neighbours on a 64-byte boundary, and every node's descriptor must say what its struct says.
The wavefront traversal's 128-byte quad records (quad_tree) must be that tree as well: expanding records from the entry reaches every
node of the reference array exactly once -- a leaf child as a slot of its own, an interior child through its two children's slots --
with the reference's boxes, primitives and split axes, interior slots pointing at their own records."""
import ctypes as C

import numpy as np
import pytest

import tray_rust_amd as T
from tray_rust_amd import _lib as L
from tray_rust_amd import scenes
import _emu

NODE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<u4"), ("desc", "<u4")])
REF = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<u4"), ("count", "<u2"), ("axis", "u1"), ("pad", "u1")])
MESH = np.dtype([("node_offset", "<u4"), ("node_count", "<u4"), ("tri_offset", "<u4"), ("tri_count", "<u4")])
WFI = np.dtype([("inv", "<f4", 12), ("flags", "<u4"), ("inst", "<u4"), ("a", "<u4"), ("b", "<u4")])
QUAD = np.dtype([("lo", "<f4", (3, 4)), ("hi", "<f4", (3, 4)), ("desc", "<u4", 4), ("meta", "<u4", 4)])
WI_POINT, WI_ANIMATED, WI_AFFINE = 8, 16, 32
assert QUAD.itemsize == 128


def quad_trees(flat, bfs_levels=-1):
    h = _emu.emu()
    h.emu_quad_trees.restype = C.c_int
    h.emu_quad_trees.argtypes = [C.POINTER(L.TrayFlatScene), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    counts = np.zeros(5, np.uint32)
    assert h.emu_quad_trees(flat, bfs_levels, counts.ctypes.data, None, None, None, None) == 0
    top = np.zeros(counts[0], QUAD); mesh = np.zeros(counts[1], QUAD); first = np.zeros(counts[2], np.uint32)
    narrow = C.c_int(0)
    assert h.emu_quad_trees(flat, bfs_levels, counts.ctypes.data, top.ctypes.data, mesh.ctypes.data, first.ctypes.data, C.addressof(narrow)) == 0
    return top, mesh, first, int(counts[3]), int(counts[4]), bool(narrow.value)


def check_quad_tree(ref, recs, collapsed=True, leaf_records=False):
    """ref: the reference's preorder array; recs: the quad records of the same tree, entry record first. Returns the number of records.
    leaf_records (BVH<Instance>, round 5): a leaf's slot refers to one more record that holds the leaf's primitives one per slot, in leaf order, each
    behind a box of its own, visited in slot order whatever the direction (order word 0)."""
    n = len(ref)
    seen_nodes, seen_recs = set(), set()
    empty_lo, empty_hi = np.float32(np.inf), np.float32(np.inf)      # every plane at +inf: nothing enters

    def slot_is(rec, s, x):
        """slot s of record rec is node x of the reference; returns x's record if x is interior"""
        assert x not in seen_nodes; seen_nodes.add(x)
        assert (rec["lo"][:, s] == ref[x]["bmin"]).all() and (rec["hi"][:, s] == ref[x]["bmax"]).all()
        d = int(rec["desc"][s])
        assert d >> 28 == 0
        if ref[x]["count"] > 0 and leaf_records:
            assert (d >> 23) == 0 and d < len(recs) and d not in seen_recs
            seen_recs.add(d)
            leaf = recs[d]
            cnt, off = int(ref[x]["count"]), int(ref[x]["offset"])
            assert 1 <= cnt <= 4 and int(leaf["meta"][0]) == 0 and int(leaf["meta"][1]) == 0
            for k in range(4):
                if k < cnt:
                    assert int(leaf["desc"][k]) == (off + k) | (1 << 23)
                    assert (leaf["lo"][:, k] <= leaf["hi"][:, k]).all()
                else:
                    slot_empty(leaf, k)
            return None
        if ref[x]["count"] > 0:
            assert d == int(ref[x]["offset"]) | (int(ref[x]["count"]) << 23)
            return None
        assert (d >> 23) == 0 and d < len(recs)
        return d

    def slot_empty(rec, s):
        assert (rec["lo"][:, s] == empty_lo).all() and (rec["hi"][:, s] == empty_hi).all()

    def inside(c, p):
        return bool((ref[c]["bmin"] >= ref[p]["bmin"]).all() and (ref[c]["bmax"] <= ref[p]["bmax"]).all())

    stack = []
    r0 = recs[0]
    root_collapsible = ref[0]["count"] == 0 and inside(1, 0) and inside(int(ref[0]["offset"]), 0)
    seen_recs.add(0)
    if root_collapsible:
        seen_nodes.add(0)
        stack.append((0, 0))
    else:           # a leaf root, or a root whose box has to be tested: the entry record holds it as its only slot
        d = slot_is(r0, 0, 0)
        for s in (1, 2, 3): slot_empty(r0, s)
        if d is not None: stack.append((d, 0)); seen_recs.add(d)
    while stack:
        j, i = stack.pop()                      # record j holds the slots of interior node i
        rec = recs[j]
        axes = int(rec["meta"][0])
        assert axes & 3 == ref[i]["axis"]
        for octant in range(8):       # the visiting order the record tabulates per direction octant = the near-child rule on the three axes
            want = ((octant >> (axes & 3)) & 1) | (((octant >> ((axes >> 2) & 3)) & 1) << 1) | (((octant >> ((axes >> 4) & 3)) & 1) << 2)
            assert (int(rec["meta"][1]) >> (3 * octant)) & 7 == want
        for g, c in enumerate((i + 1, int(ref[i]["offset"]))):
            if ref[c]["count"] == 0 and inside(c + 1, c) and inside(int(ref[c]["offset"]), c):      # replaced by its children
                assert collapsed
                seen_nodes.add(c)
                assert (axes >> (2 + 2 * g)) & 3 == ref[c]["axis"]
                for k, x in enumerate((c + 1, int(ref[c]["offset"]))):
                    d = slot_is(rec, 2 * g + k, x)
                    if d is not None:
                        assert d not in seen_recs; seen_recs.add(d); stack.append((d, x))
            else:
                d = slot_is(rec, 2 * g, c)
                slot_empty(rec, 2 * g + 1)
                if d is not None:
                    assert d not in seen_recs; seen_recs.add(d); stack.append((d, c))
    assert len(seen_nodes) == n, "every node of the reference tree is behind exactly one slot"
    return len(seen_recs)


def device_trees(flat):
    h = _emu.emu()
    h.emu_device_trees.restype = C.c_int
    h.emu_device_trees.argtypes = [C.POINTER(L.TrayFlatScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    counts = np.zeros(4, np.uint32)
    assert h.emu_device_trees(flat, counts.ctypes.data, None, None, None, None, None) == 0
    top = np.zeros(counts[0], NODE); mesh = np.zeros(counts[1], NODE); meshes = np.zeros(counts[2], MESH); recs = np.zeros(counts[3], WFI)
    narrow = C.c_int(0)
    assert h.emu_device_trees(flat, counts.ctypes.data, top.ctypes.data, mesh.ctypes.data, meshes.ctypes.data, recs.ctypes.data, C.addressof(narrow)) == 0
    return top, mesh, meshes, recs, bool(narrow.value)


def as_array(ptr, n, dtype):
    return np.frombuffer(C.string_at(ptr, int(n) * dtype.itemsize), dtype=dtype)


def check_tree(ref, dev):
    """ref: the reference's preorder array; dev: the same tree in device order"""
    assert len(dev) == len(ref) + 1
    assert np.isposinf(dev[1]["bmin"]).all() and np.isneginf(dev[1]["bmax"]).all()      # the root's twin: a box no ray enters
    order = []                      # device indices in the reference's preorder
    stack = [0]
    while stack:
        j = stack.pop()
        order.append(j)
        d = int(dev[j]["desc"])
        if (d >> 23) & 31 == 0:     # interior: the pair at `offset`
            c = int(dev[j]["offset"])
            assert c % 2 == 0 and c >= 2, "siblings share one 64-byte aligned record"
            stack.append(c + 1); stack.append(c)
    assert len(order) == len(ref) and len(set(order)) == len(order) and 1 not in order
    got = dev[order]
    assert (got["bmin"] == ref["bmin"]).all() and (got["bmax"] == ref["bmax"]).all()
    leaf = ref["count"] > 0
    assert ((got["desc"] >> 23) & 31 == ref["count"]).all()
    assert (((got["desc"] >> 28) & 3)[~leaf] == ref["axis"][~leaf]).all()
    assert (got["desc"] >> 30 == 0).all()                               # a descriptor is a STK_NODE stack entry as it stands
    assert (got["offset"][leaf] == ref["offset"][leaf]).all()           # leaves keep their primitives
    assert ((got["desc"] & 0x7fffff) == (got["offset"] & 0x7fffff)).all()
    # an interior node's second child in the reference array = the node the device keeps at offset + 1
    pos = {j: k for k, j in enumerate(order)}
    for k in np.nonzero(~leaf)[0][:2000]:
        c = int(got["offset"][k])
        assert pos[c] == k + 1 and pos[c + 1] == int(ref["offset"][k])


def scene_files(tmp_path):
    scenes.write_assets(str(tmp_path), cornell=(64, 64, 4), small=(64, 64, 4))
    out = [str(tmp_path / "cornell_box.json"), str(tmp_path / "smallpt.json")]
    p = scenes.write_tr15_like_assets(str(tmp_path / "t"), film=(64, 64, 4), detail=0.02)
    out.append(p if isinstance(p, str) else p[0])
    return out


def test_device_order_is_the_same_tree(tmp_path):
    for path in scene_files(tmp_path):
        scene, rt, spp, fi = T.Scene.load_file(path)
        flat = scene.flatten(0)
        f = flat.contents
        top, mesh, meshes, recs, narrow = device_trees(flat)
        assert narrow
        check_tree(as_array(f.top_nodes, f.n_top_nodes, REF), top)
        ref_meshes = as_array(f.meshes, f.n_meshes, MESH)
        ref_nodes = as_array(f.mesh_nodes, f.n_mesh_nodes, REF)
        assert len(meshes) == f.n_meshes
        for m in range(f.n_meshes):
            r, d = ref_meshes[m], meshes[m]
            assert d["tri_offset"] == r["tri_offset"] and d["tri_count"] == r["tri_count"] and d["node_offset"] % 2 == 0
            check_tree(ref_nodes[r["node_offset"]:r["node_offset"] + r["node_count"]], mesh[d["node_offset"]:d["node_offset"] + d["node_count"]])


@pytest.mark.parametrize("bfs_levels", [-1, 0, 2])
def test_quad_records_are_the_same_tree(tmp_path, bfs_levels):
    for path in scene_files(tmp_path):
        scene, rt, spp, fi = T.Scene.load_file(path)
        flat = scene.flatten(0)
        f = flat.contents
        top, mesh, first, top_pend, mesh_pend, narrow = quad_trees(flat, bfs_levels)
        assert narrow
        assert check_quad_tree(as_array(f.top_nodes, f.n_top_nodes, REF), top, leaf_records=True) == len(top)
        ref_meshes = as_array(f.meshes, f.n_meshes, MESH)
        ref_nodes = as_array(f.mesh_nodes, f.n_mesh_nodes, REF)
        assert len(first) == f.n_meshes
        ends = list(first[1:]) + [len(mesh)]
        for m in range(f.n_meshes):
            r = ref_meshes[m]
            recs = mesh[first[m]:ends[m]]
            assert check_quad_tree(ref_nodes[r["node_offset"]:r["node_offset"] + r["node_count"]], recs) == len(recs)
        assert top_pend >= 1 and (f.n_meshes == 0 or mesh_pend >= 1)


def test_a_box_that_does_not_contain_its_children_keeps_its_own_test(tmp_path):
    """A caller-built BVH may hold a node whose box is NOT the union of its children's: the implied-box argument does not hold for
    it, so it stays a slot of its own (explicit test) -- and the quad traversal still answers like the reference's binary one."""
    d = str(tmp_path)
    p, _ = scenes.write_dragon_assets(d, film=(64, 64, 4), grid=24, extent=1.0)
    scene, *_ = T.Scene.load_file(p)
    flat = scene.flatten(0)
    f = flat.contents
    nodes = as_array(f.mesh_nodes, f.n_mesh_nodes, REF).copy()
    interior = np.nonzero(nodes["count"] == 0)[0]
    rng = np.random.default_rng(5)
    for i in rng.choice(interior[1:], 40, replace=False):      # shrink the box: some children now stick out
        c = 0.5 * (nodes["bmin"][i] + nodes["bmax"][i])
        nodes["bmin"][i] = c + 0.6 * (nodes["bmin"][i] - c); nodes["bmax"][i] = c + 0.6 * (nodes["bmax"][i] - c)
    keep = f.mesh_nodes
    f.mesh_nodes = C.cast(nodes.ctypes.data, type(f.mesh_nodes))
    try:
        top, mesh, first, *_ = quad_trees(flat)
        me = as_array(f.meshes, f.n_meshes, MESH)[0]
        check_quad_tree(nodes[me["node_offset"]:me["node_offset"] + me["node_count"]], mesh[first[0]:], collapsed=True)
        import _oracle as O
        cam = O.camera_rays(flat, rng.uniform(0, [64, 64], (6000, 2)))
        ref = _emu.debug_intersect(flat, cam, O.HIT_DTYPE)     # trace_bvh: the reference's traversal over the (broken) binary tree
        hit, t, inst, prim = _emu.wf_trace(flat, cam, 0)
        ref_hit = ref["inst"] != 0xffffffff
        assert (hit == ref_hit).all() and (t[hit] == ref["t"][hit]).all() and (prim[hit] == ref["prim"][hit]).all()
        assert ref_hit.mean() > 0.3
    finally:
        f.mesh_nodes = keep


def test_instance_records(tmp_path):
    path = scene_files(tmp_path)[2]
    scene, rt, spp, fi = T.Scene.load_file(path)
    flat = scene.flatten(330)
    f = flat.contents
    top, mesh, meshes, recs, narrow = device_trees(flat)
    quad_first = quad_trees(flat)[2]
    assert len(recs) == f.n_top_order
    seen_mesh = seen_moving = 0
    for k in range(f.n_top_order):
        i = f.top_order[k]
        inst = f.instances[i]
        r = recs[k]
        assert r["inst"] == i and (r["flags"] & 7) == inst.geom_type
        assert (r["inv"] == np.array(inst.inv[:12], np.float32)).all()
        assert bool(r["flags"] & WI_POINT) == (inst.kind == L.TRAY_INST_POINT_EMITTER if hasattr(L, "TRAY_INST_POINT_EMITTER") else inst.kind == 2)
        assert bool(r["flags"] & WI_AFFINE) == (tuple(inst.inv[12:16]) == (0.0, 0.0, 0.0, 1.0))
        assert bool(r["flags"] & WI_ANIMATED) == bool(inst.animated)
        if inst.animated:
            assert r["flags"] >> 8 == inst.moving_slot; seen_moving += 1
        if inst.geom_type == 3:
            assert r["a"] == quad_first[inst.mesh_id] and r["b"] == meshes[inst.mesh_id]["tri_offset"]; seen_mesh += 1
        else:
            assert (np.array([r["a"], r["b"]], np.uint32).view(np.float32) == np.array(inst.geom_params[:2], np.float32)).all()
    assert seen_mesh > 0 and seen_moving > 0


def test_a_broken_tree_is_refused(tmp_path):
    scene, rt, spp, fi = T.Scene.load_file(scene_files(tmp_path)[0])
    flat = scene.flatten(0)
    f = flat.contents
    nodes = as_array(f.top_nodes, f.n_top_nodes, REF).copy()
    interior = np.nonzero(nodes["count"] == 0)[0]
    nodes["offset"][interior[0]] = f.n_top_nodes + 5        # second child outside the array
    keep = f.top_nodes
    f.top_nodes = C.cast(nodes.ctypes.data, type(f.top_nodes))
    try:
        h = _emu.emu()
        h.emu_device_trees.restype = C.c_int
        h.emu_device_trees.argtypes = [C.POINTER(L.TrayFlatScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        counts = np.zeros(4, np.uint32)
        assert h.emu_device_trees(flat, counts.ctypes.data, None, None, None, None, None) == 1
    finally:
        f.top_nodes = keep
