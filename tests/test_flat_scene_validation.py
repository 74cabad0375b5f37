"""tray_scene_create() checks every index of a caller-built TrayFlatScene before anything is uploaded (csrc/host/validate.hpp):
the device kernels follow BVH children, ordered lists, light and table indices without bounds checks. Runs without a GPU --
the check comes before the first HIP call -- by patching copies of a loader-produced scene."""
import ctypes as C

import numpy as np
import pytest

import tray_rust_amd as T
from tray_rust_amd import _lib as L, scenes


def clone(flat):
    f = L.TrayFlatScene()
    C.memmove(C.byref(f), flat, C.sizeof(L.TrayFlatScene))
    return f


def array_copy(ptr, n, ctype):
    buf = (ctype * n)()
    C.memmove(buf, ptr, n * C.sizeof(ctype))
    return buf


def create(f):
    d = C.c_void_p()
    rc = L.lib().tray_scene_create(C.byref(f), C.byref(d))
    msg = (L.lib().tray_last_error() or b"").decode()
    if rc == 0:
        L.lib().tray_scene_destroy(d)
    return rc, msg


@pytest.fixture(scope="module")
def dragon(tmp_path_factory, built):
    d = str(tmp_path_factory.mktemp("val"))
    p, _ = scenes.write_dragon_assets(d, film=(64, 64, 4), grid=8, extent=1.0)
    scene, *_ = T.Scene.load_file(p)
    return scene, scene.flatten(0)


def test_loader_scene_passes_the_check(dragon):
    rc, msg = create(clone(dragon[1]))
    assert rc == 0 or "inconsistent" not in msg   # without a GPU the call fails later, at hipSetDevice


CASES = {
    "top leaf past the ordered list": lambda f, keep: patch_node(f, keep, "top_nodes", f.n_top_nodes, leaf=True),
    "top interior child out of range": lambda f, keep: patch_node(f, keep, "top_nodes", f.n_top_nodes, leaf=False),
    "mesh leaf past the triangles": lambda f, keep: patch_node(f, keep, "mesh_nodes", f.n_mesh_nodes, leaf=True),
    "mesh interior child out of range": lambda f, keep: patch_node(f, keep, "mesh_nodes", f.n_mesh_nodes, leaf=False),
    "ordered instance id": lambda f, keep: patch_u32(f, keep, "top_order", f.n_top_order, 0, 1000),
    "light id": lambda f, keep: patch_u32(f, keep, "lights", f.n_lights, 0, 1000),
    "light is a receiver": lambda f, keep: patch_u32(f, keep, "lights", f.n_lights, 0, receiver_of(f)),
    "material id": lambda f, keep: patch_instance(f, keep, receiver_of(f), "material_id", 77),
    "mesh id": lambda f, keep: patch_instance(f, keep, mesh_instance_of(f), "mesh_id", 9),
    "geometry type": lambda f, keep: patch_instance(f, keep, receiver_of(f), "geom_type", 11),
    "spline stack": lambda f, keep: patch_instance(f, keep, 0, "xf_first", 10 ** 6),
    "mesh triangle range": lambda f, keep: patch_mesh(f, keep, "tri_count", 10 ** 8),
    "mesh node range": lambda f, keep: patch_mesh(f, keep, "node_offset", 10 ** 8),
    "merl table offset": lambda f, keep: patch_merl(f, keep),
    "null array": lambda f, keep: setattr(f, "tri_attrs", C.cast(None, L._P(L.TrayTriAttrs))),
}


def receiver_of(f):
    return next(i for i in range(f.n_instances) if f.instances[i].kind == 0)


def mesh_instance_of(f):
    return next(i for i in range(f.n_instances) if f.instances[i].geom_type == 3)


def patch_node(f, keep, field, n, leaf):
    nodes = array_copy(getattr(f, field), n, L.TrayBvhNode)
    i = next(k for k in range(n) if (nodes[k].count > 0) == leaf)
    nodes[i].offset = 0x7fffffff if leaf else (i + 1 if i + 1 < n else 0)   # second child == first child: not a tree
    keep.append(nodes)
    setattr(f, field, C.cast(nodes, L._P(L.TrayBvhNode)))


def patch_u32(f, keep, field, n, i, value):
    a = array_copy(getattr(f, field), n, C.c_uint32)
    a[i] = value
    keep.append(a)
    setattr(f, field, C.cast(a, L._P(C.c_uint32)))


def patch_instance(f, keep, i, member, value):
    a = array_copy(f.instances, f.n_instances, L.TrayInstance)
    setattr(a[i], member, value)
    keep.append(a)
    f.instances = C.cast(a, L._P(L.TrayInstance))


def patch_mesh(f, keep, member, value):
    a = array_copy(f.meshes, f.n_meshes, L.TrayMesh)
    setattr(a[0], member, value)
    keep.append(a)
    f.meshes = C.cast(a, L._P(L.TrayMesh))


def patch_merl(f, keep):
    assert f.n_merl >= 1
    a = array_copy(f.merl_tables, f.n_merl, L.TrayMerlTable)
    a[0].offset = f.n_merl_floats - 5
    keep.append(a)
    f.merl_tables = C.cast(a, L._P(L.TrayMerlTable))


@pytest.mark.parametrize("case", sorted(CASES))
def test_inconsistent_scene_is_rejected_before_any_device_call(case, dragon):
    f = clone(dragon[1])
    keep = []
    CASES[case](f, keep)
    rc, msg = create(f)
    assert rc == L.TRAY_E_INVALID and "inconsistent scene" in msg, (case, rc, msg)


# ---- moving scenes: the device indexes the per-path transform cache by moving_slot and searches the knots of every moving level
@pytest.fixture(scope="module")
def moving(tmp_path_factory, built):
    import json
    import os
    d = str(tmp_path_factory.mktemp("valmov"))
    scenes.write_assets(d)
    p = os.path.join(d, "moving_box.json")
    json.dump(scenes.moving_box(64, 64, 4), open(p, "w"))
    scene, *_ = T.Scene.load_file(p)
    return scene, scene.flatten(1)


def animated_instances(f):
    return [i for i in range(f.n_instances) if f.instances[i].animated]


def moving_level(f):
    return next(l for l in range(f.n_xf_levels) if f.xf_levels[l].kf_count >= 2)


def patch_knot(f, keep, level, k, value):
    a = array_copy(f.knots, f.n_knots, C.c_float)
    a[f.xf_levels[level].knot_first + k] = value
    keep.append(a)
    f.knots = C.cast(a, L._P(C.c_float))


MOVING_CASES = {
    "moving_slot past the animated instances": lambda f, keep: patch_instance(f, keep, animated_instances(f)[0], "moving_slot", 0xffffffff),
    "moving_slot shared by two instances": lambda f, keep: patch_instance(f, keep, animated_instances(f)[0], "moving_slot",
                                                                          f.instances[animated_instances(f)[1]].moving_slot),
    "decreasing knots": lambda f, keep: patch_knot(f, keep, moving_level(f), 1, 1e9),
    "NaN knot": lambda f, keep: patch_knot(f, keep, moving_level(f), 2, float("nan")),
}


def test_moving_scene_of_the_loader_passes_the_check(moving):
    f = moving[1].contents
    assert len(animated_instances(f)) >= 2
    rc, msg = create(clone(moving[1]))
    assert rc == 0 or "inconsistent" not in msg


@pytest.mark.parametrize("case", sorted(MOVING_CASES))
def test_inconsistent_moving_scene_is_rejected(case, moving):
    f = clone(moving[1])
    keep = []
    MOVING_CASES[case](f, keep)
    rc, msg = create(f)
    assert rc == L.TRAY_E_INVALID and "inconsistent scene" in msg, (case, rc, msg)
