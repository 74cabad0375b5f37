"""The product's loader (csrc/host/scene.cpp -> TrayFlatScene) against an INDEPENDENT reading of the same scene files
(tests/_indep_loader.py, written from the reference's src/scene.rs alone): oracle and product consume the same TrayFlatScene, so
without this a loader bug -- a material bound to the wrong object, a transform composed in the wrong order, a plane that is not
2x2 -- would be invisible to every parity test. Bundled scenes (scenes/*.json, restated from the reference's files) always;
the reference's own files when /root/reference exists (build container)."""
import ctypes as C
import os

import numpy as np
import pytest

import tray_rust_amd as T
import _indep_loader as I

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = [os.path.join(ROOT, "scenes", n) for n in ("cornell_box.json", "smallpt.json")]
FILES += [os.path.join("/root/reference/scenes", n) for n in ("cornell_box.json", "smallpt.json") if os.path.exists("/root/reference/scenes")]


@pytest.mark.parametrize("path", FILES)
def test_flattened_scene_equals_the_independent_reading(path, built):
    want = I.load(path)
    scene, rt, spp, fi = T.Scene.load_file(path)
    fs = scene.flatten(0).contents
    assert (fs.film.width, fs.film.height, spp) == (want["width"], want["height"], want["samples"])
    assert (fs.min_depth, fs.max_depth) == (want["min_depth"], want["max_depth"])
    table = np.array(list(fs.film.table), np.float64).reshape(16, 16)
    assert np.abs(table - want["filter_table"]).max() < 2e-6
    cam = np.array(list(fs.camera.cam_world), np.float64).reshape(4, 4)
    assert np.abs(cam - want["cam_world"]).max() < 1e-4 * max(1.0, np.abs(want["cam_world"]).max())
    insts = want["instances"]
    assert fs.n_instances == len(insts)
    assert [fs.lights[i] for i in range(fs.n_lights)] == want["lights"]
    for i, w in enumerate(insts):
        g = fs.instances[i]
        assert (g.kind, g.geom_type) == (w["kind"], w["geom_type"]), w["name"]
        mat = np.array(list(g.mat), np.float64).reshape(4, 4)
        inv = np.array(list(g.inv), np.float64).reshape(4, 4)
        scale = max(1.0, np.abs(w["mat"]).max())
        assert np.abs(mat - w["mat"]).max() < 1e-4 * scale, (w["name"], mat, w["mat"])
        assert np.abs(inv - np.linalg.inv(w["mat"])).max() < 1e-4 * max(1.0, np.abs(np.linalg.inv(w["mat"])).max()), w["name"]
        if w["kind"] != 0:
            e = w["emission"]   # Colorf(r, g, b) * strength (scene.rs load_color / emitters)
            got = [g.emission[0], g.emission[1], g.emission[2]]
            assert np.allclose(got, [e[0] * e[3], e[1] * e[3], e[2] * e[3]], rtol=1e-6), w["name"]
        if w["geom_type"] == 4:
            continue
        for k, p in enumerate(w["params"]):
            assert abs(g.geom_params[k] - p) < 1e-6, (w["name"], k)
        if w["geom_type"] == 3:
            assert fs.meshes[g.mesh_id].tri_count == w["triangles"], w["name"]
        m, wm = fs.materials[g.material_id], w["material"]
        assert m.kind == wm["kind"], w["name"]
        for key, got in (("c0", m.c0), ("c1", m.c1)):
            if key in wm:
                assert np.allclose(list(got)[:3], wm[key], rtol=1e-6), (w["name"], key)
        for key, got in (("f0", m.f0), ("f1", m.f1)):
            if key in wm:
                assert abs(got - wm[key]) < 1e-6 * max(1.0, abs(wm[key])), (w["name"], key)


def _mesh_rows(fs, mesh_id):
    """the product's triangles of one mesh, put back into OBJ order with TrayTriVerts::tri_id"""
    m = fs.meshes[mesh_id]
    n = m.tri_count
    tv = np.ctypeslib.as_array(C.cast(fs.tri_verts, C.POINTER(C.c_float)), shape=(fs.n_tris, 12))[m.tri_offset:m.tri_offset + n]
    ta = np.ctypeslib.as_array(C.cast(fs.tri_attrs, C.POINTER(C.c_float)), shape=(fs.n_tris, 16))[m.tri_offset:m.tri_offset + n]
    ids = tv[:, 3].copy().view(np.uint32)
    order = np.argsort(ids)
    assert (ids[order] == np.arange(n)).all()   # every OBJ triangle exactly once
    pos = tv[order][:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(n, 3, 3)
    nrm = ta[order][:, 0:9].reshape(n, 3, 3)
    tex = ta[order][:, 9:15].reshape(n, 3, 2)
    return pos, nrm, tex, m


@pytest.mark.parametrize("which", ["cube", "dragon_stand_in"])
def test_obj_meshes_equal_the_independent_reading(which, tmp_path, built):
    """OBJ parsing (tobj through mesh.rs:50-78): every triangle's positions, shading normals and texture coordinates, in file order,
    and the mesh bounds -- the reference's own cube.obj and the 871 200-triangle grid's little brother (same writer, 48 x 48 grid)"""
    from tray_rust_amd import scenes
    d = str(tmp_path)
    if which == "cube":
        scenes.write_assets(d)
        scene, *_ = T.Scene.load_file(os.path.join(d, "cornell_box.json"))
        obj, model = os.path.join(d, "models", "cube.obj"), "Cube"
        if os.path.exists("/root/reference/scenes/models/cube.obj"):   # the bundled copy describes the reference's mesh
            for a, b in zip(I.read_obj_model(obj, model), I.read_obj_model("/root/reference/scenes/models/cube.obj", model)):
                assert a.shape == b.shape and (a == b).all()
    else:
        path, ntri = scenes.write_dragon_assets(d, film=(64, 48, 4), grid=48, extent=1.0)
        scene, *_ = T.Scene.load_file(path)
        obj, model = os.path.join(d, "models", "dragon.obj"), "dragon"
    fs = scene.flatten(0).contents
    want_p, want_n, want_t = I.read_obj_model(obj, model)
    mesh_ids = {fs.instances[i].mesh_id for i in range(fs.n_instances) if fs.instances[i].geom_type == 3}
    assert len(mesh_ids) == 1
    pos, nrm, tex, m = _mesh_rows(fs, mesh_ids.pop())
    assert m.tri_count == len(want_p) > 0
    assert (pos == want_p.astype(np.float32)).all() and (nrm == want_n.astype(np.float32)).all() and (tex == want_t.astype(np.float32)).all()
    root = fs.mesh_nodes[m.node_offset]
    assert np.allclose(list(root.bmin), want_p.reshape(-1, 3).min(axis=0), rtol=1e-6, atol=1e-6)
    assert np.allclose(list(root.bmax), want_p.reshape(-1, 3).max(axis=0), rtol=1e-6, atol=1e-6)


def test_merl_table_equals_the_independent_reading(tmp_path, built):
    """Merl::load_file (material/merl.rs:51-84): header, f64 planes, channel scales (1, 1, 1.66) / 1500, clamp, rgb interleave"""
    from tray_rust_amd import scenes
    path, _ = scenes.write_dragon_assets(str(tmp_path), film=(64, 48, 4), grid=16, extent=1.0)
    scene, *_ = T.Scene.load_file(path)
    fs = scene.flatten(0).contents
    assert fs.n_merl == 1
    tb = fs.merl_tables[0]
    assert (tb.n_theta_h, tb.n_theta_d, tb.n_phi_d) == (90, 90, 180)
    merl_files = [os.path.join(r, f) for r, _, fl in os.walk(str(tmp_path)) for f in fl if f.endswith(".binary")]
    assert len(merl_files) == 1
    want = I.read_merl(merl_files[0])
    got = np.ctypeslib.as_array(fs.merl_data, shape=(fs.n_merl_floats,))[tb.offset:tb.offset + want.size]
    assert got.size == want.size == 3 * 90 * 90 * 180
    assert (got == want).all()
    assert want.max() > 0.0 and (want.reshape(-1, 3)[:, 2] > 0).any()
