"""AnimatedMesh (geometry/animated_mesh.rs): vertices, normals and texcoords interpolated between keyframes at ray.time
behind ONE tree built for the first keyframe interval. The reference's loader has no branch that constructs one (scene.rs:588-628), so the
scene entry follows the module's own documentation (animated_mesh.rs:14-27) and the checks are analytic: at a keyframe's time the mesh IS
that keyframe's static mesh; inside the first interval the tree is exact (brute force agrees); outside it the never-rebuilt tree loses
hits (quirk Q13: Boundable::update_deformation has no caller). The DEVICE code (ANIM = 3 instantiations, host emulation) against the
oracle bit for bit. GPU: tests/test_gpu_parity.py::test_gpu_animated_mesh."""
import json
import os

import numpy as np
import pytest

import tray_rust_amd as T
from tray_rust_amd import _lib as L
from tray_rust_amd import scenes
import _emu as E
import _oracle as O

N_KEYS, GRID = 4, 12


@pytest.fixture(scope="module")
def flag(tmp_path_factory, built):
    d = str(tmp_path_factory.mktemp("flag"))
    path = scenes.write_waving_flag(d, grid=GRID, n_keys=N_KEYS, width=64, height=48, samples=8, frames=8, scene_time=2.0)
    scene, rt, spp, fi = T.Scene.load_file(path)
    return d, scene, fi


def flag_rays(flat, n, time, seed=0):
    """rays from the camera side towards the sheet's box, and a few from behind it; all with ray.time = time"""
    rng = np.random.default_rng(seed)
    o = rng.uniform([-14, 2, -30], [14, 22, -10], (n, 3))
    tgt = rng.uniform([-8, 4, 0], [6, 18, 8], (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    back = rng.uniform(0, 1, n) < 0.2
    o[back] = tgt[back] + 12 * d[back]; d[back] = -d[back]
    rays = np.concatenate([o, d, np.zeros((n, 1)), np.full((n, 1), np.inf), np.full((n, 1), time)], axis=1).astype(np.float32)
    return rays


def test_loader_flattens_an_animated_mesh(flag):
    d, scene, fi = flag
    flat = scene.flatten(0)
    fs = flat.contents
    inst = [fs.instances[i] for i in range(fs.n_instances) if fs.instances[i].geom_type == 5]      # TRAY_GEOM_ANIMATED_MESH
    assert len(inst) == 1 and fs.animated == 1
    m = inst[0].mesh_id
    assert fs.n_mesh_keys == fs.n_meshes and fs.mesh_keys[m].n_keys == N_KEYS
    times = [fs.key_times[fs.mesh_keys[m].time_first + k] for k in range(N_KEYS)]
    assert np.allclose(times, [0, 2 / 3, 4 / 3, 2.0])
    me = fs.meshes[m]
    assert me.tri_count == 2 * GRID * GRID and me.tri_offset + N_KEYS * me.tri_count <= fs.n_tris
    # the triangles of every keyframe sit in the SAME (leaf) order: tri_id is the OBJ triangle, and the texcoords (the sheet's own
    # parameters) do not move
    for k in range(1, N_KEYS):
        for j in (0, 17, me.tri_count - 1):
            a, b = fs.tri_verts[me.tri_offset + j], fs.tri_verts[me.tri_offset + k * me.tri_count + j]
            assert a.tri_id == b.tri_id and list(a.pa)[:2] == list(b.pa)[:2] and a.pa[2] != b.pa[2] or a.pa[0] == -6.0
            ta, tb = fs.tri_attrs[me.tri_offset + j], fs.tri_attrs[me.tri_offset + k * me.tri_count + j]
            assert list(ta.ta) == list(tb.ta)
    # the tree's root box is the union of keyframes 0 and 1 (AnimatedTriangle::bounds at times[0], times[1]), not of all four
    root = fs.mesh_nodes[me.node_offset]
    z01 = [fs.tri_verts[me.tri_offset + k * me.tri_count + j].__getattribute__(f)[2] for k in (0, 1) for j in range(me.tri_count) for f in ("pa", "pb", "pc")]
    assert root.bmin[2] == min(z01) and root.bmax[2] == max(z01)


def _static_twin(d, key):
    """the same scene with the sheet as a plain "mesh" of keyframe `key`"""
    doc = scenes.waving_flag(width=64, height=48, samples=8, frames=8, scene_time=2.0, n_keys=N_KEYS)
    for o in doc["objects"]:
        if o.get("name") == "flag":
            o["geometry"] = {"type": "mesh", "file": "models/flag_%d.obj" % key, "model": "Flag"}
    p = os.path.join(d, "static_%d.json" % key)
    json.dump(doc, open(p, "w"))
    scene, *_ = T.Scene.load_file(p)
    return scene


@pytest.mark.parametrize("key", [0, 1])
def test_at_a_keyframe_time_the_mesh_is_that_keyframe(flag, key):
    """position(i, times[k]) returns keyframe k's vertex (animated_mesh.rs:73-76: binary_search -> Ok(k) -> (k, None)): every ray meets
    what it meets in the scene with the static mesh of that keyframe -- t, p, n, uv to the bit (prim differs: another tree, another leaf order)"""
    d, scene, fi = flag
    flat = scene.flatten(0)
    time = [0.0, 2.0 / 3.0][key]
    fs = flat.contents
    m = [fs.instances[i] for i in range(fs.n_instances) if fs.instances[i].geom_type == 5][0].mesh_id
    time = np.float32(fs.key_times[fs.mesh_keys[m].time_first + key])
    rays = flag_rays(flat, 4000, time)
    a = O.intersect(flat, rays)
    twin = _static_twin(d, key)
    b = O.intersect(twin.flatten(0), rays)
    flag_inst = [i for i in range(fs.n_instances) if fs.instances[i].geom_type == 5][0]
    assert (a["inst"] == flag_inst).mean() > 0.3
    for f in ("t", "inst", "p", "n", "ng", "u", "v", "dp_du", "dp_dv"):
        assert np.array_equal(a[f], b[f], equal_nan=True), f


def test_inside_the_first_interval_the_tree_is_exact_and_outside_it_loses_hits(flag):
    d, scene, fi = flag
    flat = scene.flatten(0)
    fs = flat.contents
    flag_inst = [i for i in range(fs.n_instances) if fs.instances[i].geom_type == 5][0]
    for time in (0.1, 0.33, 0.6):      # between keyframes 0 and 1: a lerped vertex lies between its two ends, inside the triangle's box
        rays = flag_rays(flat, 3000, time, seed=3)
        a, b = O.intersect(flat, rays), O.intersect(flat, rays, flags=O.BRUTE_FORCE)
        assert (a["inst"] == flag_inst).mean() > 0.3
        for f in a.dtype.names:
            assert np.array_equal(a[f], b[f], equal_nan=True), (time, f)
    lost = 0
    for time in (1.0, 1.5, 1.9):       # keyframes 1..3: the tree still bounds keyframes 0 and 1 only (quirk Q13)
        rays = flag_rays(flat, 3000, time, seed=4)
        a, b = O.intersect(flat, rays), O.intersect(flat, rays, flags=O.BRUTE_FORCE)
        on_a, on_b = a["inst"] == flag_inst, b["inst"] == flag_inst
        assert not (on_a & ~on_b & (a["t"] < b["t"])).any()      # the tree never finds a hit brute force does not
        lost += int((on_b & ~on_a).sum())
    assert lost > 50


def test_interpolation_is_the_reference_lerp(flag):
    """a ray straight down a vertex's column: the hit height is a * (1 - x) + b * x of the two keyframes' z (linalg::lerp, f32), x from the times"""
    d, scene, fi = flag
    flat = scene.flatten(0)
    fs = flat.contents
    inst = [fs.instances[i] for i in range(fs.n_instances) if fs.instances[i].geom_type == 5][0]
    me = fs.meshes[inst.mesh_id]
    inv = np.array(list(inst.inv), np.float32).reshape(4, 4); mat = np.array(list(inst.mat), np.float32).reshape(4, 4)
    f32 = np.float32
    t0, t1 = f32(fs.key_times[0]), f32(fs.key_times[1])
    time = f32(0.25)
    x = f32(f32(time - t0) / f32(t1 - t0))
    checked = 0
    for j in range(0, me.tri_count, 7):
        v0, v1 = fs.tri_verts[me.tri_offset + j], fs.tri_verts[me.tri_offset + me.tri_count + j]
        # object-space point strictly inside the lerped triangle: its centroid; shoot along -z of object space
        tri = []
        for name in ("pa", "pb", "pc"):
            a, b = np.array(list(getattr(v0, name)), f32), np.array(list(getattr(v1, name)), f32)
            tri.append((a * f32(f32(1.0) - x) + b * x).astype(f32))
        c = (tri[0] + tri[1] + tri[2]) / f32(3)
        o_obj = np.array([c[0], c[1], c[2] + 5.0, 1.0], np.float32); d_obj = np.array([0, 0, -1, 0], np.float32)
        o_w, d_w = mat @ o_obj, mat @ d_obj
        ray = np.concatenate([o_w[:3], d_w[:3], [0.0, np.inf, time]]).astype(np.float32)[None]
        h = O.intersect(flat, ray)[0]
        if h["inst"] == 0xffffffff:
            continue
        p_obj = inv @ np.append(h["p"], 1.0).astype(np.float32)
        # the hit lies in the plane of the lerped triangle
        n = np.cross(tri[1] - tri[0], tri[2] - tri[0]); n /= np.linalg.norm(n)
        assert abs(float(np.dot(p_obj[:3] - tri[0], n))) < 1e-4
        checked += 1
    assert checked > 20


def test_device_code_of_the_animated_mesh_is_bit_identical(flag):
    d, scene, fi = flag
    for frame in (0, 1, 5):
        flat = scene.flatten(frame)
        cam = flat.contents.camera
        for time in (cam.shutter_open, 0.5 * (cam.shutter_open + cam.shutter_close), 0.0, 2.0 / 3.0, 5.0, -1.0):
            rays = flag_rays(flat, 1500, time, seed=frame)
            a, b = O.intersect(flat, rays), E.debug_intersect(flat, rays, O.HIT_DTYPE)
            for f in a.dtype.names:
                assert np.array_equal(a[f], b[f], equal_nan=True), (frame, time, f)
        # per-sample radiance through the lane machine (k_debug_sample_radiance<3>)
        rng = np.random.default_rng(frame)
        px = rng.integers(0, 64, 300).astype(np.uint32); py = rng.integers(0, 48, 300).astype(np.uint32); si = rng.integers(0, 8, 300).astype(np.uint32)
        a, b = O.sample_radiance(flat, px, py, si, 8, seed=5), E.sample_radiance(flat, px, py, si, 8, 5)
        assert np.array_equal(a, b)


def test_renders_of_the_animated_mesh_scene(flag):
    """LowDiscrepancy renders of such a scene run k_sampler_pass<3> one thread per sample -- the tile kernel's samples under the same keys --
    and the other samplers apply unchanged"""
    d, scene, fi = flag
    flat = scene.flatten(1)
    q = np.array(list(T.BlockQueue((64, 48))), np.uint32)
    ref, st = O.render_tiles(flat, 8, seed=2, threads=1)
    img, (samples, vertices, rays) = E.render_sampler(flat, q, O.SAMPLER_LOW_DISCREPANCY, 8, 8, seed=2)
    assert samples == st.samples == 64 * 48 * 8 and vertices == st.vertices and rays == st.rays
    np.testing.assert_allclose(img, ref, rtol=5e-5, atol=5e-5)
    ref, st, counts = O.render_tiles_sampler(flat, O.SAMPLER_ADAPTIVE, 2, 16, seed=2, threads=1)
    img, (samples, _, _) = E.render_sampler(flat, q, O.SAMPLER_ADAPTIVE, 2, 16, seed=2)
    assert samples == st.samples and counts.max() > 2
    np.testing.assert_allclose(img, ref, rtol=5e-5, atol=5e-5)
    # the sheet is in the picture and moves between frames
    a, _ = O.render_tiles(scene.flatten(0), 8, seed=2)
    b, _ = O.render_tiles(scene.flatten(3), 8, seed=2)
    ra, rb = a[..., :3] / np.maximum(a[..., 3:], 1e-20), b[..., :3] / np.maximum(b[..., 3:], 1e-20)
    assert np.sqrt(np.mean((ra - rb) ** 2)) > 0.01


def test_loader_refuses_broken_animated_meshes(flag, tmp_path):
    d, scene, fi = flag
    def load(mutate):
        doc = scenes.waving_flag(width=64, height=48, samples=8, n_keys=N_KEYS)
        geo = [o for o in doc["objects"] if o.get("name") == "flag"][0]["geometry"]
        mutate(geo)
        p = os.path.join(d, "broken.json")
        json.dump(doc, open(p, "w"))
        return T.Scene.load_file(p)
    with pytest.raises(T.TrayError, match="at least two keyframes"):
        load(lambda g: g.update(keyframes=g["keyframes"][:1]))
    with pytest.raises(T.TrayError, match="ascending"):
        load(lambda g: g["keyframes"].reverse())
    with pytest.raises(T.TrayError, match="Keyframes are required"):
        load(lambda g: g.pop("keyframes"))
    with pytest.raises(T.TrayError, match="was not found"):
        load(lambda g: g.update(model="Flagg"))
    with open(os.path.join(d, "models", "small.obj"), "w") as f:
        f.write(scenes.flag_obj(GRID - 1, 0.0))
    with pytest.raises(T.TrayError, match="vertices"):
        load(lambda g: g["keyframes"][2].update(file="models/small.obj"))
    # an area light cannot be an animated mesh (not Sampleable, scene.rs:584-654)
    doc = scenes.waving_flag(width=64, height=48, samples=8, n_keys=N_KEYS)
    light = [o for o in doc["objects"] if o.get("type") == "emitter"][0]
    light["geometry"] = [o for o in doc["objects"] if o.get("name") == "flag"][0]["geometry"]
    p = os.path.join(d, "light.json"); json.dump(doc, open(p, "w"))
    with pytest.raises(T.TrayError, match="not sampleable"):
        T.Scene.load_file(p)
