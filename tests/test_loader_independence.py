"""The product's loader (csrc/host/scene.cpp -> TrayFlatScene) against an INDEPENDENT reading of the same scene files
(tests/_indep_loader.py, written from the reference's src/scene.rs alone): oracle and product consume the same TrayFlatScene, so
without this a loader bug -- a material bound to the wrong object, a transform composed in the wrong order, a plane that is not
2x2 -- would be invisible to every parity test. Bundled scenes (scenes/*.json, restated from the reference's files) always;
the reference's own files when /root/reference exists (build container)."""
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
