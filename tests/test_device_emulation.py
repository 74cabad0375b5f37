"""The DEVICE source, compiled for the host (tests/emu: g++ behind a shim, one-lane waves), against the oracle -- no GPU needed.
Covers what is per-lane in the device code: Scene::intersect as k_debug_intersect runs it (flat instance loop or two-level
traversal, primitive tests, hit finishing) and the wavefront traversal kernels k_wf_trace_dyn / k_wf_trace_wide in all three
stages, with the stacks split between LDS and the HBM overflow column, through the pool fields and queues the stage kernels
use. k_wf_trace_wide is checked in both node formats: the exact 128-B nodes of the shipped library and the 64-B quantised
nodes staged behind -DTR_QWIDE (Next / C5) -- the variant's device code has not run on a GPU yet, this is its check.
Same libm on both sides here, so the records are compared bit for bit."""
import numpy as np
import pytest

import tray_rust_amd as T
from tray_rust_amd import scenes
import _emu as E
import _oracle as O
from test_proto_wide_bvh import mixed_rays

BOX = ([-14, 1, -18], [14, 23, 19])


@pytest.fixture(scope="module")
def dragon(tmp_path_factory, built):
    d = str(tmp_path_factory.mktemp("emu_dragon"))
    p, _ = scenes.write_dragon_assets(d, film=(160, 120, 4), grid=64, extent=1.0)
    scene, *_ = T.Scene.load_file(p)
    return scene, scene.flatten(0)


@pytest.fixture(scope="module")
def tr15(tmp_path_factory, built):
    """59 instances behind BVH<Instance>, many meshes; frame 0 with every ray at shutter-open time (static kernels)"""
    d = str(tmp_path_factory.mktemp("emu_tr15"))
    p = scenes.write_tr15_like_assets(d, film=(160, 96, 4), detail=0.02)
    scene, *_ = T.Scene.load_file(p if isinstance(p, str) else p[0])
    return scene, scene.flatten(0)


def rays_for(flat, seed, n, stage, centre, spread):
    rays = mixed_rays(flat, np.random.default_rng(seed), n, flat.contents.film.width, flat.contents.film.height, BOX[0], BOX[1], centre, spread)
    rays[:, 8] = flat.contents.camera.shutter_open
    if stage == 0:      # camera rays (min_t 0) and continuation rays (0.001), both unbounded
        rays[:, 7] = np.inf
    elif stage == 1:    # occlusion segments (light/mod.rs:21-23)
        rays[:, 6] = 0.001; rays[:, 7] = 0.999
    else:               # BSDF-sampled light rays
        rays[:, 6] = 0.001; rays[:, 7] = np.inf
    return rays


@pytest.mark.parametrize("name", ["cornell_box", "smallpt"])
def test_debug_intersect_device_code_is_bit_exact(name, tmp_path, built):
    scenes.write_assets(str(tmp_path), cornell=(160, 120, 4), small=(160, 120, 4))
    scene, *_ = T.Scene.load_file(str(tmp_path / (name + ".json")))
    flat = scene.flatten(0)
    rays = rays_for(flat, 3, 20000, 0, [0, 10, 0], 8.0)
    a, b = O.intersect(flat, rays), E.debug_intersect(flat, rays, O.HIT_DTYPE)
    assert (a["inst"] != 0xffffffff).mean() > 0.8
    for f in a.dtype.names:
        assert (a[f] == b[f]).all() or np.array_equal(a[f], b[f], equal_nan=True), f


def test_debug_intersect_device_code_on_a_mesh_scene(dragon):
    flat = dragon[1]
    rays = rays_for(flat, 4, 20000, 0, [8.5, 3.7, 1.5], 4.0)
    a, b = O.intersect(flat, rays), E.debug_intersect(flat, rays, O.HIT_DTYPE)
    assert (a["inst"] == 6).mean() > 0.05
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f], equal_nan=True), f


def check_stage(flat, rays, stage, got):
    ref = O.intersect(flat, rays)
    hit, t, inst, prim = got
    ref_hit = ref["inst"] != 0xffffffff
    assert (hit == ref_hit).all(), f"{(hit != ref_hit).sum()} of {len(rays)} rays differ in hit / occlusion"
    if stage != 1:      # the any-hit stage only reports occlusion
        assert (t[hit] == ref["t"][hit]).all() and (inst[hit] == ref["inst"][hit]).all() and (prim[hit] == ref["prim"][hit]).all()
    return float(ref_hit.mean())


@pytest.mark.parametrize("stage", [0, 1, 2])
@pytest.mark.parametrize("kernel,qwide", [(0, False), (1, False), (1, True)], ids=["dyn", "wide-exact", "wide-quantised"])
def test_wavefront_traversal_kernels_on_a_mesh_scene(dragon, kernel, qwide, stage):
    flat = dragon[1]
    rays = rays_for(flat, 10 + stage, 12000, stage, [8.5, 3.7, 1.5], 4.0)
    for lds_depth in (0, 3):    # everything in LDS / almost everything in the HBM overflow column
        got = E.wf_trace(flat, rays, kernel, stage, lds_depth=lds_depth, qwide=qwide)
        if qwide:   # a superset of the visits: a candidate the exact box culls by rounding may be accepted (none expected)
            ref = O.intersect(flat, rays)
            differ = int(((got[0] != (ref["inst"] != 0xffffffff))).sum())
            assert differ <= 1
            if differ:
                continue
        frac = check_stage(flat, rays, stage, got)
        assert 0.05 < frac <= 1.0


@pytest.mark.parametrize("stage", [0, 1])
@pytest.mark.parametrize("kernel,qwide", [(0, False), (1, False), (1, True)], ids=["dyn", "wide-exact", "wide-quantised"])
def test_wavefront_traversal_kernels_behind_bvh_of_instances(tr15, kernel, qwide, stage):
    flat = tr15[1]
    assert flat.contents.n_instances > 16
    rays = rays_for(flat, 20 + stage, 8000, stage, [0, 5, 0], 10.0)
    got = E.wf_trace(flat, rays, kernel, stage, lds_depth=4, qwide=qwide)
    check_stage(flat, rays, stage, got)
