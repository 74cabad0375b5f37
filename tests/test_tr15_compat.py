"""The loader against the reference's own scenes/tr15.json (BASELINE.json configs[4]): 600 frames, a camera and five objects
/ groups on cubic B-splines, ten lights with animated emission, 13 OBJ files with 25 models, 5 MERL BRDFs. The models and
BRDFs are not distributed with the reference, so seeded stand-ins are generated at the paths the scene names. Runs only
where /root/reference exists (this container); nothing here travels to the GPU box."""
import json
import os
import shutil

import numpy as np
import pytest

import ctypes as C

import tray_rust_amd as T
from tray_rust_amd import _lib as L
from tray_rust_amd import scenes
import _oracle as O

TR15 = "/root/reference/scenes/tr15.json"
pytestmark = pytest.mark.skipif(not os.path.exists(TR15), reason="reference scenes are only present in the build container")


@pytest.fixture(scope="module")
def tr15(tmp_path_factory, built):
    d = str(tmp_path_factory.mktemp("tr15"))
    shutil.copy(TR15, os.path.join(d, "tr15.json"))
    desc = json.load(open(TR15))
    files = {}

    def walk(objs):
        for o in objs:
            g = o.get("geometry")
            if g and g["type"] == "mesh":
                files.setdefault(g["file"], set()).add(g["model"])
            if o["type"] == "group":
                walk(o["objects"])
    walk(desc["objects"])
    seed = 0
    for path, models in sorted(files.items()):
        objs = []
        for m in sorted(models):
            seed += 1
            objs.append((m,) + scenes.knot_mesh(10, 6, seed=seed, extent=1.0))
        scenes.write_obj(os.path.join(d, path), objs)
    first = None
    for m in desc["materials"]:
        if m["type"] == "merl":
            dst = os.path.join(d, m["file"])
            if first is None:
                scenes.write_merl_binary(dst)
                first = dst
            else:
                os.link(first, dst)
    return desc, T.Scene.load_file(os.path.join(d, "tr15.json"))


def test_tr15_loads_with_every_feature(tr15):
    desc, (scene, rt, spp, fi) = tr15
    assert (rt.width, rt.height, spp) == (1920, 1080, 2048)
    assert (fi.frames, fi.time, fi.start, fi.end) == (600, 25.0, 0, 599)
    assert scene.info.n_lights == 10 and scene.info.n_instances == 59 and scene.info.n_meshes == 25
    fs = scene.flatten(0).contents
    assert (fs.min_depth, fs.max_depth) == (5, 10) and fs.n_merl == 5
    assert fs.n_instances > 16   # BVH<Instance> path on the device


@pytest.mark.parametrize("frame,n_moving", [(0, 8), (150, 8), (330, 2), (599, 1)])
def test_tr15_frames_flatten_and_trace(tr15, frame, n_moving):
    desc, (scene, rt, spp, fi) = tr15
    flat = scene.flatten(frame)
    fs = flat.contents
    step = np.float32(25.0) / np.float32(600)
    assert fs.camera.shutter_open == np.float32(frame) * step
    assert np.isclose(fs.camera.shutter_close - fs.camera.shutter_open, 0.5 * step, rtol=1e-4)
    assert fs.camera.animated == (1 if frame >= 108 else 0) and fs.animated == 1   # the camera spline starts at t = 4.5
    # 14 instances have splines (3 walls with own + group splines, 4 x (light, cone) under moving groups, dragon, rust_logo,
    # cow); only those whose knot domain the open shutter enters need per-ray evaluation
    splined = [i for i in range(fs.n_instances)
               if any(fs.xf_levels[fs.instances[i].xf_first + l].kf_count > 1 for l in range(fs.instances[i].xf_count))]
    assert len(splined) == 14
    moving = [i for i in range(fs.n_instances) if fs.instances[i].animated]
    assert len(moving) == n_moving and set(moving) <= set(splined)
    assert sorted(fs.instances[i].moving_slot for i in moving) == list(range(n_moving))
    keyed = [i for i in range(fs.n_instances) if fs.instances[i].emis_count >= 2]
    assert len(keyed) == 10
    # the library's spline evaluation (instance matrices at shutter_open) against the oracle's, bit for bit
    for i in splined:
        inst = fs.instances[i]
        out = np.zeros(32, np.float32)
        assert O.oracle().oracle_stack_transform(flat, inst.xf_first, inst.xf_count, float(fs.camera.shutter_open), out.ctypes.data) == 0
        assert (np.frombuffer(inst.mat, np.float32) == out[:16]).all() and (np.frombuffer(inst.inv, np.float32) == out[16:]).all()
    img, st = O.render_tiles(flat, 4, seed=1, tile_start=16000, tile_count=6)
    assert np.isfinite(img).all() and st.samples == 6 * 64 * 4


@pytest.mark.parametrize("name", ["cornell_box", "smallpt", "logo_with_friends", "suzanne_scene", "logo_shadow"])
def test_every_bundled_reference_scene_loads_and_renders(name, tmp_path, built):
    """All scene files the reference ships (besides tr15.json above): the reference's own JSON, its own models where it
    ships them (cube.obj), generated stand-ins for the models and BRDF tables it does not distribute."""
    src_dir = os.path.dirname(TR15)
    d = str(tmp_path)
    shutil.copy(os.path.join(src_dir, name + ".json"), os.path.join(d, name + ".json"))
    desc = json.load(open(os.path.join(src_dir, name + ".json")))
    files = {}

    def walk(objs):
        for o in objs:
            g = o.get("geometry")
            if g and g["type"] == "mesh":
                files.setdefault(g["file"], set()).add(g["model"])
            if o["type"] == "group":
                walk(o["objects"])
    walk(desc["objects"])
    seed = 50
    for path, models in sorted(files.items()):
        dst = os.path.join(d, path)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if os.path.exists(os.path.join(src_dir, path)):
            shutil.copy(os.path.join(src_dir, path), dst)
            continue
        objs = []
        for m in sorted(models):
            seed += 1
            objs.append((m,) + scenes.knot_mesh(10, 6, seed=seed, extent=1.0))
        scenes.write_obj(dst, objs)
    first = None
    for m in desc["materials"]:
        if m["type"] == "merl":
            dst = os.path.join(d, m["file"])
            if first is None:
                scenes.write_merl_binary(dst)
                first = dst
            elif not os.path.exists(dst):
                os.link(first, dst)
    scene, rt, spp, fi = T.Scene.load_file(os.path.join(d, name + ".json"))
    assert (rt.width, rt.height, spp) == (desc["film"]["width"], desc["film"]["height"], desc["film"]["samples"])
    flat = scene.flatten(0)
    fs = flat.contents
    assert fs.n_lights >= 1 and fs.animated == 0
    n_tiles = (rt.width // 8) * (rt.height // 8)
    img, st = O.render_tiles(flat, 4, seed=1, tile_start=n_tiles // 2, tile_count=4)
    assert np.isfinite(img).all() and st.samples == 4 * 64 * 4


def test_generated_cube_and_scenes_equal_the_reference_files(tmp_path, built):
    """scenes.cube_obj() / cornell_box() / smallpt() restate scenes/models/cube.obj, cornell_box.json and smallpt.json (those
    files do not exist on the GPU box): the flattened scenes must be identical, bit for bit."""
    src_dir = os.path.dirname(TR15)
    ref_dir, gen_dir = str(tmp_path / "ref"), str(tmp_path / "gen")
    os.makedirs(os.path.join(ref_dir, "models")); os.makedirs(gen_dir)
    shutil.copy(os.path.join(src_dir, "models", "cube.obj"), os.path.join(ref_dir, "models", "cube.obj"))
    for name, build in (("cornell_box", scenes.cornell_box), ("smallpt", scenes.smallpt)):
        desc = json.load(open(os.path.join(src_dir, name + ".json")))
        shutil.copy(os.path.join(src_dir, name + ".json"), os.path.join(ref_dir, name + ".json"))
        scenes.write_assets(gen_dir, cornell=(desc["film"]["width"], desc["film"]["height"], desc["film"]["samples"]),
                            small=(desc["film"]["width"], desc["film"]["height"], desc["film"]["samples"]))
        a = T.Scene.load_file(os.path.join(ref_dir, name + ".json"))[0]
        b = T.Scene.load_file(os.path.join(gen_dir, name + ".json"))[0]
        fa, fb = a.flatten(0).contents, b.flatten(0).contents
        assert (fa.n_instances, fa.n_tris, fa.n_mesh_nodes, fa.n_materials, fa.n_lights) == (fb.n_instances, fb.n_tris, fb.n_mesh_nodes, fb.n_materials, fb.n_lights)
        for field, n, size in (("instances", fa.n_instances, C.sizeof(L.TrayInstance)), ("tri_verts", fa.n_tris, 48), ("tri_attrs", fa.n_tris, 64),
                               ("mesh_nodes", fa.n_mesh_nodes, 32), ("top_nodes", fa.n_top_nodes, 32), ("materials", fa.n_materials, C.sizeof(L.TrayMaterial))):
            ba = C.string_at(C.cast(getattr(fa, field), C.c_void_p), n * size)
            bb = C.string_at(C.cast(getattr(fb, field), C.c_void_p), n * size)
            assert ba == bb, (name, field)
        assert bytes(fa.camera) == bytes(fb.camera) and bytes(fa.film) == bytes(fb.film)


def test_tr15_spline_stacks_equal_the_independent_reading(tr15):
    """Keyframes, knots, degrees and the nesting of group splines of the reference's own tr15.json, read a second time from
    scene.rs:825-850 / animated_transform.rs:22-86 alone (tests/_indep_loader.py): per instance the stack is [own level, parent
    group's, grandparent's ...]; every control point's decomposed keyframe (translation, quaternion, scaling -- the product's
    Jacobi polar decomposition in place of la's SVD) recomposes to the control transform the JSON describes."""
    import _indep_loader as I
    desc, (scene, rt, spp, fi) = tr15
    fs = scene.flatten(0).contents
    stacks = I.instance_stacks(desc["objects"])
    assert len(stacks) == fs.n_instances == 59
    n_splines = 0
    for i, (name, levels) in enumerate(stacks):
        inst = fs.instances[i]
        assert inst.xf_count == len(levels), name
        for l, want in enumerate(levels):
            lv = fs.xf_levels[inst.xf_first + l]
            assert lv.kf_count == len(want["mats"]), (name, l)
            if want["knots"] is not None:
                n_splines += 1
                assert lv.degree == want["degree"], (name, l)
                knots = [fs.knots[lv.knot_first + k] for k in range(lv.knot_count)]
                assert knots == [float(np.float32(x)) for x in want["knots"]], (name, l)
                assert lv.knot_count == lv.kf_count + lv.degree + 1
            for k, m in enumerate(want["mats"]):
                kf = fs.keyframes[lv.kf_first + k]
                got = I.keyframe_matrix(list(kf.translation), list(kf.rotation), list(kf.scaling))
                want_m = I.keyframe_of(m)   # (what Keyframe::new keeps of the control transform: T, the polar rotation, the DIAGONAL of the stretch)
                assert np.abs(got - want_m).max() < 2e-5 * max(1.0, np.abs(m).max()), (name, l, k, got, want_m)
    assert n_splines >= 14   # the 14 splined instances of test_tr15_frames_flatten_and_trace (group splines count once per member)
    # the camera's own spline
    cam = desc["camera"]["keyframes"]
    assert fs.camera.xf_count == 1
    lv = fs.xf_levels[fs.camera.xf_first]
    assert lv.kf_count == len(cam["control_points"]) and lv.degree == cam.get("degree", 3)
    assert [fs.knots[lv.knot_first + k] for k in range(lv.knot_count)] == [float(np.float32(x)) for x in cam["knots"]]
    for k, c in enumerate(cam["control_points"]):
        kf = fs.keyframes[lv.kf_first + k]
        m = I.load_transform(c["transform"])
        assert np.abs(I.keyframe_matrix(list(kf.translation), list(kf.rotation), list(kf.scaling)) - I.keyframe_of(m)).max() < 2e-5 * max(1.0, np.abs(m).max()), k
