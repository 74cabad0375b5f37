"""The device order of the BVHs (tray_rust_amd/csrc/host/gates.hpp: pair_tree) and the wavefront traversal's instance records
(wf_inst_records), read back through the host emulation library and checked against the reference-order arrays of the flat scene:
the paired tree must be the SAME tree -- walking it first child / second child reproduces the reference's preorder array (bvh.rs:278-295)
node for node --, siblings must be neighbours on a 64-byte boundary, and every node's descriptor must say what its struct says."""
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
WI_POINT, WI_ANIMATED, WI_AFFINE = 8, 16, 32


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


def test_instance_records(tmp_path):
    path = scene_files(tmp_path)[2]
    scene, rt, spp, fi = T.Scene.load_file(path)
    flat = scene.flatten(330)
    f = flat.contents
    top, mesh, meshes, recs, narrow = device_trees(flat)
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
            assert r["a"] == meshes[inst.mesh_id]["node_offset"] and r["b"] == meshes[inst.mesh_id]["tri_offset"]; seen_mesh += 1
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
