"""The DEVICE source, compiled for the host (tests/emu: g++ behind a shim, one-lane waves), against the oracle -- no GPU needed.
Covers what is per-lane in the device code: Scene::intersect as k_debug_intersect runs it (flat instance loop or two-level
traversal, primitive tests, hit finishing) and the wavefront traversal kernel k_wf_trace_dyn in all three stages, with the
stacks split between LDS and the HBM overflow column, through the pool fields and queues the stage kernels use.
Same libm on both sides here, so the records are compared bit for bit."""
import os

import numpy as np
import pytest

import tray_rust_amd as T
from tray_rust_amd import scenes
import _emu as E
import _oracle as O

BOX = ([-14, 1, -18], [14, 23, 19])


def mixed_rays(flat, rng, n, width, height, lo, hi, centre, spread):
    """n camera rays + n rays from inside the scene's box, half of them segments (max_t 0.999), half unbounded and normalised"""
    cam = O.camera_rays(flat, rng.uniform(0, [width, height], (n, 2)))
    o = rng.uniform(lo, hi, (n, 3)); d = rng.normal(centre, spread, (n, 3)) - o
    seg = rng.uniform(0, 1, n) < 0.5
    d[~seg] /= np.linalg.norm(d[~seg], axis=1, keepdims=True)
    inner = np.concatenate([o, d, np.full((n, 1), 0.001), np.where(seg, 0.999, np.inf)[:, None], np.zeros((n, 1))], axis=1).astype(np.float32)
    return np.concatenate([cam, inner])



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
def test_wavefront_traversal_kernel_on_a_mesh_scene(dragon, stage):
    flat = dragon[1]
    rays = rays_for(flat, 10 + stage, 12000, stage, [8.5, 3.7, 1.5], 4.0)
    for lds_depth in (0, 3):    # everything in LDS / almost everything in the HBM overflow column
        got = E.wf_trace(flat, rays, stage, lds_depth=lds_depth)
        frac = check_stage(flat, rays, stage, got)
        assert 0.05 < frac <= 1.0


@pytest.mark.parametrize("stage", [0, 1])
def test_wavefront_traversal_kernel_behind_bvh_of_instances(tr15, stage):
    flat = tr15[1]
    assert flat.contents.n_instances > 16
    rays = rays_for(flat, 20 + stage, 8000, stage, [0, 5, 0], 10.0)
    got = E.wf_trace(flat, rays, stage, lds_depth=4)
    check_stage(flat, rays, stage, got)
    # (round 5) the instances of a BVH<Instance> leaf sit behind conservative boxes of their own: the answers are the reference's with and without them
    os.environ["TRAYHIP_NO_INSTANCE_BOXES"] = "1"
    try:
        plain = E.wf_trace(flat, rays, stage, lds_depth=4)
    finally:
        del os.environ["TRAYHIP_NO_INSTANCE_BOXES"]
    for a, b in zip(got, plain):
        assert np.array_equal(a[got[0]], b[got[0]]) if a is not got[0] else np.array_equal(a, b)


def wf_deferred():
    h = E.emu()
    h.emu_wf_deferred.restype = __import__("ctypes").c_uint
    return int(h.emu_wf_deferred())


@pytest.mark.parametrize("which", ["dragon", "tr15"])
@pytest.mark.parametrize("stage", [0, 1])
def test_wavefront_traversal_hands_axis_parallel_rays_to_the_reference_traversal(dragon, tr15, which, stage):
    """The quad records drop the box of a node that was replaced by its children; that is only sound for rays whose reciprocal direction
    is finite and nonzero (host/gates.hpp). Rays with a zero (+0 and -0), denormal or infinite direction component -- for which the
    reference's slab test itself is erratic -- must come back exactly as the reference's binary traversal answers them: k_wf_trace_dyn
    hands them to k_wf_trace_fallback (trace_bvh)."""
    flat = (dragon if which == "dragon" else tr15)[1]
    centre, spread = ([8.5, 3.7, 1.5], 4.0) if which == "dragon" else ([0, 5, 0], 10.0)
    rays = rays_for(flat, 40 + stage, 3000, stage, centre, spread)
    rng = np.random.default_rng(41)
    special = np.array([0.0, -0.0, 1e-42, -1e-42, 0.0, -0.0], np.float32)
    for k in range(len(rays)):      # one or two components of every ray's direction become +0 / -0 / denormal
        for c in rng.choice(3, rng.integers(1, 3), replace=False):
            rays[k, 3 + c] = special[rng.integers(0, len(special))]
    # ... and origins on the box planes of the scene, where 0 * inf shows up in the slab test
    rays[::7, 0] = np.float32(BOX[0][0]); rays[1::7, 1] = np.float32(BOX[0][1])
    wf_deferred()
    got = E.wf_trace(flat, rays, stage, lds_depth=4)
    n_deferred = wf_deferred()
    assert n_deferred >= len(rays), n_deferred        # (a ray is handed over at most once; camera + inner rays of rays_for)
    check_stage(flat, rays, stage, got)
    # the regular rays of the other tests are never handed over on the way into a mesh either
    regular = rays_for(flat, 50 + stage, 2000, stage, centre, spread)
    E.wf_trace(flat, regular, stage)
    assert wf_deferred() == 0


def test_wavefront_traversal_hands_over_a_ray_that_turns_axis_parallel_inside_an_instance(tmp_path, built):
    """A ray that is regular in world space but gets a zero direction component in an instance's object space (here: a sheared
    mesh instance and rays with d.x = -d.y) is handed over when it enters that instance's mesh -- whole, from its origin."""
    p, _ = scenes.write_dragon_assets(str(tmp_path), film=(160, 120, 4), grid=48, extent=1.0)
    scene, *_ = T.Scene.load_file(p)
    flat = scene.flatten(0)
    f = flat.contents
    mesh_inst = [i for i in range(f.n_instances) if f.instances[i].geom_type == 3][0]
    inst = f.instances[mesh_inst]
    inv0 = np.array(inst.inv[:16], np.float64).reshape(4, 4)
    sc = 0.078125                                                          # ~ the instance's own 1 / 13, exact in binary
    new_inv = np.array([[sc, sc, 0, inv0[0, 3]], [0, sc, 0, inv0[1, 3]], [0, 0, sc, inv0[2, 3]], [0, 0, 0, 1]])   # object x = s (x + y): zero for d.x = -d.y
    new_mat = np.linalg.inv(new_inv)
    for k in range(16):
        inst.inv[k] = float(np.float32(new_inv.flat[k])); inst.mat[k] = float(np.float32(new_mat.flat[k]))
    new_inv = new_inv.astype(np.float32)
    rng = np.random.default_rng(61)
    n = 4000
    o = rng.uniform(BOX[0], BOX[1], (n, 3)).astype(np.float32)
    target = rng.normal([8.5, 3.7, 1.5], 3.0, (n, 3)).astype(np.float32)
    d = target - o
    v = (0.5 * (d[:, 0] - d[:, 1])).astype(np.float32)
    d[:, 0] = v; d[:, 1] = -v
    rays = np.concatenate([o, d, np.full((n, 1), 0.001), np.full((n, 1), np.inf), np.zeros((n, 1))], axis=1).astype(np.float32)
    rays[:, 8] = f.camera.shutter_open
    row0 = new_inv[0, :3]
    assert ((row0[0] * rays[:, 3] + row0[1] * rays[:, 4]) + row0[2] * rays[:, 5] == 0).mean() > 0.5   # (xf_vector's order of operations)
    wf_deferred()
    got = E.wf_trace(flat, rays, 0, lds_depth=4)
    assert 0 < wf_deferred() < n          # only the rays that reach the instance
    check_stage(flat, rays, 0, got)


# ---- the whole per-sample path: sampler, camera, traversal, integrator, BSDFs (k_debug_sample_radiance, k_debug_bsdf)

@pytest.mark.parametrize("name,spp", [("cornell_box", 64), ("smallpt", 64), ("dragon", 16)])
def test_per_sample_radiance_of_the_device_code_is_bit_identical(name, spp, tmp_path, built):
    """With the same libm under both, the device source and the oracle produce the same bits for every camera sample: colour,
    sample position, number of path vertices and rays. (On the GPU the remaining differences are ocml vs glibc, tests -m gpu.)"""
    d = str(tmp_path)
    scenes.write_assets(d, cornell=(128, 96, spp), small=(128, 96, spp))
    if name == "dragon":
        scenes.write_dragon_assets(d, film=(128, 96, spp), grid=32, extent=1.0)   # 3-lobe MERL material, mesh, disk light
    scene, *_ = T.Scene.load_file(str(tmp_path / (name + ".json")))
    flat = scene.flatten(0)
    rng = np.random.default_rng(11)
    n = 12000
    px = rng.integers(0, 128, n).astype(np.uint32); py = rng.integers(0, 96, n).astype(np.uint32); si = rng.integers(0, spp, n).astype(np.uint32)
    a = O.sample_radiance(flat, px, py, si, spp, seed=21)
    b = E.sample_radiance(flat, px, py, si, spp, 21)
    assert a[:, 5].mean() > 1.5 and (a[:, :3] > 0).any()
    assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("name,frames", [("moving_box", (0, 3, 5)), ("tr15_like", (0, 127, 330))])
def test_per_sample_radiance_of_the_device_code_on_a_moving_scene(name, frames, tmp_path, built):
    """Moving scenes, bit for bit: the device evaluates the spline stacks with the libm the reference calls -- glibc's acosf / sinf / cosf
    restated in dev_libm.h (tools/libm_port_check.cpp compares them with the system's exhaustively) -- and applies Transform * Point's
    w quirk (Q5) with the inverse's own [3][3] element, which Matrix4::inverse leaves an ulp off one for some keyframes. Round 3 rounded
    f64 results and assumed w == 1: 0.8 .. 10 % of the samples of these scenes then differed in their last bits, a few took another path."""
    d = str(tmp_path)
    if name == "moving_box":
        scenes.write_moving_box(d, width=128, height=96, samples=32)
        scene, *_ = T.Scene.load_file(str(tmp_path / "moving_box.json"))
        w, h, spp, n = 128, 96, 32, 8000
    else:
        p = scenes.write_tr15_like_assets(d, film=(64, 48, 8), detail=0.02)
        scene, *_ = T.Scene.load_file(p if isinstance(p, str) else p[0])
        w, h, spp, n = 64, 48, 8, 4000
    lit = False
    for frame in frames:
        flat = scene.flatten(frame)
        assert flat.contents.animated
        rng = np.random.default_rng(12 + frame)
        px = rng.integers(0, w, n).astype(np.uint32); py = rng.integers(0, h, n).astype(np.uint32); si = rng.integers(0, spp, n).astype(np.uint32)
        a = O.sample_radiance(flat, px, py, si, spp, seed=5)
        b = E.sample_radiance(flat, px, py, si, spp, 5)
        assert a[:, 5].mean() > 1.0
        lit = lit or bool((a[:, :3] > 0).any())      # (the stand-in's lights are keyed: dark at frame 0)
        assert a.tobytes() == b.tobytes(), f"{name} frame {frame}: {(~(a == b).all(axis=1)).sum()} of {n} samples differ"
    assert lit


def test_spline_stacks_of_the_device_code_are_bit_identical(tmp_path, built):
    """AnimatedTransform::transform(time) of every moving stack (camera, instances, nested groups) at 4000 shutter times: rows of mat and
    inv and the inverse's [3][3] element, device source against the oracle, every bit"""
    import ctypes as C
    from tray_rust_amd import _lib as L
    d = str(tmp_path)
    scenes.write_moving_box(d, width=64, height=48, samples=4)
    scene, *_ = T.Scene.load_file(str(tmp_path / "moving_box.json"))
    h = E.emu()
    h.emu_stack_transform.restype = C.c_int
    h.emu_stack_transform.argtypes = [C.POINTER(L.TrayFlatScene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    o = O.oracle()
    o.oracle_stack_transform.restype = C.c_int
    o.oracle_stack_transform.argtypes = [C.POINTER(L.TrayFlatScene), C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
    off_one = 0
    for frame in (0, 3, 7):
        flat = scene.flatten(frame)
        f = flat.contents
        n = 4000
        times = np.random.default_rng(frame).uniform(f.camera.shutter_open, f.camera.shutter_close, n).astype(np.float32)
        stacks = [(f.camera.xf_first, f.camera.xf_count)] + [(f.instances[i].xf_first, f.instances[i].xf_count) for i in range(f.n_instances) if f.instances[i].animated]
        assert len(stacks) >= 4
        for first, count in stacks:
            dev = np.zeros((n, 28), np.float32)
            assert h.emu_stack_transform(flat, first, count, n, times.ctypes.data, dev.ctypes.data) == 0
            ora = np.zeros((n, 32), np.float32)
            for k in range(n):
                assert o.oracle_stack_transform(flat, first, count, float(times[k]), ora[k].ctypes.data) == 0
            assert dev[:, :12].tobytes() == np.ascontiguousarray(ora[:, :12]).tobytes()          # rows 0..2 of mat
            assert dev[:, 12:24].tobytes() == np.ascontiguousarray(ora[:, 16:28]).tobytes()      # rows 0..2 of inv
            assert (dev[:, 24] == ora[:, 31]).all() and (dev[:, 25] == ora[:, 15]).all()         # the [3][3] elements
            assert (ora[:, 12:15] == 0).all() and (ora[:, 28:31] == 0).all()                     # row 3 is (0, 0, 0, w)
            off_one += int((ora[:, 31] != 1.0).sum())
    assert off_one > 0, "the scene is meant to contain keyframes whose inverse has w != 1 (quirk Q5 then divides)"


MATERIALS = {
    "matte_lambert": {"type": "matte", "diffuse": [0.7, 0.5, 0.3], "roughness": 0.0},
    "matte_oren": {"type": "matte", "diffuse": [0.7, 0.5, 0.3], "roughness": 25.0},
    "plastic": {"type": "plastic", "diffuse": [0.8, 0.2, 0.2], "gloss": [0.6, 0.6, 0.6], "roughness": 0.3},
    "metal": {"type": "metal", "refractive_index": [0.155265, 0.116723, 0.138381], "absorption_coefficient": [4.82835, 3.12225, 2.14696], "roughness": 0.2},
    "glass": {"type": "glass", "reflect": [1, 1, 1], "transmit": [0.9, 0.95, 1.0], "eta": 1.52},
    "rough_glass": {"type": "rough_glass", "reflect": [1, 1, 1], "transmit": [1, 1, 1], "eta": 1.5, "roughness": 0.3},
    "specular_metal": {"type": "specular_metal", "refractive_index": [0.2, 0.9, 1.1], "absorption_coefficient": [3.9, 2.4, 2.2]},
}


@pytest.mark.parametrize("kind", sorted(MATERIALS))
def test_bsdf_eval_pdf_sample_of_the_device_code(kind, tmp_path, built):
    import json
    d = scenes.cornell_box(64, 64, 4)
    m = dict(MATERIALS[kind]); m["name"] = "probe"
    d["materials"].append(m)
    scenes.write_assets(str(tmp_path))
    json.dump(d, open(tmp_path / "s.json", "w"))
    scene, *_ = T.Scene.load_file(str(tmp_path / "s.json"))
    flat = scene.flatten(0)
    mid = flat.contents.n_materials - 1
    rng = np.random.default_rng(3)
    n = 20000
    dirs = rng.normal(size=(n, 6)).astype(np.float32)
    dirs[:, :3] /= np.linalg.norm(dirs[:, :3], axis=1, keepdims=True)
    dirs[:, 3:] /= np.linalg.norm(dirs[:, 3:], axis=1, keepdims=True)
    u3 = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    for flags in (0, 1):
        a = O.bsdf(flat, mid, flags, dirs, u3)
        b = E.bsdf(flat, mid, flags, dirs, u3)
        assert np.array_equal(a, b, equal_nan=True), (kind, flags, np.abs(a - b).max())


# ---- whole kernels as SIMT emulations: 256 fibers per workgroup, wave intrinsics and barriers are rendezvous (hip_emu.h)

def rgb(img):
    return img[..., :3] / np.maximum(img[..., 3:], 1e-20)


def tile_queue(width, height):
    return np.array(T.BlockQueue((width, height), (8, 8)).blocks, np.uint32).reshape(-1, 2)


@pytest.mark.parametrize("name,coop,film_rows,blocks", [("cornell_box", -1, -1, 1), ("cornell_box", 0, -1, 2), ("cornell_box", -1, 0, 1),
                                                        ("smallpt", -1, -1, 2), ("dragon", -1, -1, 1)])
def test_tile_megakernel_emulated_as_simt(name, coop, film_rows, blocks, tmp_path, built):
    """k_path_tiles itself -- persistent workgroups pulling tiles, wave-synchronous stages, path regeneration, the cooperative
    small-mesh test (cornell_box's cubes), the row-binned LDS film -- run as fibers on the host: the same samples, path vertices
    and rays as the oracle, and the same image up to the order of the film's f32 sums."""
    w, h, spp = 32, 24, 8
    d = str(tmp_path)
    scenes.write_assets(d, cornell=(w, h, spp), small=(w, h, spp))
    if name == "dragon":
        scenes.write_dragon_assets(d, film=(w, h, spp), grid=16, extent=1.0)
    scene, *_ = T.Scene.load_file(str(tmp_path / (name + ".json")))
    flat = scene.flatten(0)
    img, (samples, vertices, rays, feat) = E.render_tiles(flat, tile_queue(w, h), spp, 7, blocks=blocks, coop=coop, film_rows=film_rows)
    ref, st = O.render_tiles(flat, spp, seed=7)
    assert (samples, vertices, rays) == (st.samples, st.vertices, st.rays)
    assert feat == {"cornell_box": 0, "smallpt": 4, "dragon": 1}[name]            # the feature set tray_scene_create would pick
    assert np.abs(img[..., 3] - ref[..., 3]).max() < 1e-4 * ref[..., 3].max()
    assert np.abs(rgb(img) - rgb(ref)).max() < 2e-5
    assert float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6


def test_tile_megakernel_partial_queue_and_idle_workgroups(tmp_path, built):
    """skip(start).take(count) of the tile queue (block_queue.rs:39-41) and more workgroups than tiles"""
    w, h, spp = 32, 24, 4
    scenes.write_assets(str(tmp_path), cornell=(w, h, spp), small=(w, h, spp))
    scene, *_ = T.Scene.load_file(str(tmp_path / "cornell_box.json"))
    flat = scene.flatten(0)
    q = tile_queue(w, h)
    a, sa = E.render_tiles(flat, q[:5], spp, 3, blocks=1)
    b, sb = E.render_tiles(flat, q[5:], spp, 3, blocks=9)
    full, sf = E.render_tiles(flat, q, spp, 3, blocks=2)
    assert sa[0] + sb[0] == sf[0] == w * h * spp and sa[1] + sb[1] == sf[1]
    assert np.abs((a + b) - full).max() < 1e-4 * full.max()


def test_wavefront_schedule_with_more_chunks_than_queue_segments(tmp_path, built):
    """70 chunks over the 64 segments of every queue (segments 0..5 hold two chunks' entries; the one-thread-per-entry kernels run
    128 blocks, 58 of them empty; the traversal waves hop over all 64 segments): the oracle's counts and image."""
    w, h, spp = 96, 64, 4
    scenes.write_assets(str(tmp_path), cornell=(w, h, spp), small=(w, h, spp))
    scene, *_ = T.Scene.load_file(str(tmp_path / "smallpt.json"))
    flat = scene.flatten(0)
    img, (samples, vertices, rays, rounds) = E.render_wavefront(flat, tile_queue(w, h), spp, 8, trace=0, n_chunks=70, trace_blocks=3, lds_depth=4)
    ref, st = O.render_tiles(flat, spp, seed=8)
    assert (samples, vertices, rays) == (st.samples, st.vertices, st.rays)
    assert float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6


def test_wavefront_schedule_with_binned_rays(tmp_path, built, monkeypatch):
    """TRAYHIP_WF_BIN=3 (off by default: profiles/r06_c5_ray_binning_ab.txt): the rays of stages A and B pass through k_wf_bin_hist / k_wf_bin_scatter
    before they are traced -- another order, the same rays: the oracle's counts and image, on a static and on a moving scene."""
    monkeypatch.setenv("TRAYHIP_WF_BIN", "3")
    w, h, spp = 96, 64, 4
    scenes.write_assets(str(tmp_path), cornell=(w, h, spp), small=(w, h, spp))
    scenes.write_moving_box(str(tmp_path), width=w, height=h, samples=spp)
    for name, frame in (("smallpt", 0), ("moving_box", 3)):
        scene, *_ = T.Scene.load_file(str(tmp_path / (name + ".json")))
        flat = scene.flatten(frame)
        img, (samples, vertices, rays, rounds) = E.render_wavefront(flat, tile_queue(w, h), spp, 8, trace=0, n_chunks=70, trace_blocks=3, lds_depth=4)
        ref, st = O.render_tiles(flat, spp, seed=8)
        assert (samples, vertices, rays) == (st.samples, st.vertices, st.rays), name
        assert float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6, name


def test_ray_binning_kernels_sort_every_segment(built):
    """k_wf_bin_hist + k_wf_bin_scatter (wavefront.h, round 6): every segment of a ray queue comes out as a permutation of itself -- every record
    once, bit for bit -- in non-decreasing key order, the key being (Morton cell of the origin in the box, direction octant) as an independent
    numpy reading computes it; segments of 0, 1, one workgroup's worth and several workgroups' worth of entries (20 chunks: capacity 256), origins
    outside the box, on its faces and NaN, a flat axis."""
    rng = np.random.default_rng(12)
    n_chunks = 1400          # capacity ceil(1400 / 64) * 256 = 5632 entries per segment: two workgroups of 4096 per segment
    counts = rng.integers(0, 5633, 64).astype(np.uint32)
    counts[:6] = (0, 1, 255, 4096, 4097, 5632)
    bmin, bmax = np.array([-3.0, 0.0, 2.0], np.float32), np.array([5.0, 4.0, 2.0], np.float32)   # z is flat: one cell
    rays = []
    for s_ in range(64):
        n = int(counts[s_])
        r = np.zeros((n, 8), np.uint32)
        r[:, 0] = rng.permutation(n) + 7 * s_
        o = rng.uniform([-4, -1, 1], [6, 5, 3], (n, 3)).astype(np.float32)
        if n > 10:
            o[0] = bmin; o[1] = bmax; o[2] = np.nan; o[3] = (np.inf, -np.inf, 0.0)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        if n > 10: d[4] = (0.0, -0.0, 1.0)
        r[:, 1:4] = o.view(np.uint32); r[:, 4:7] = d.view(np.uint32); r[:, 7] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        rays.append(r)
    out, keys = E.wf_bin(n_chunks, counts, rays, bmin, bmax, stage=0)

    bits = int(np.log2(E.emu().emu_wf_bin_cells()))
    def key_of(r):
        o = r[:, 1:4].copy().view(np.float32); d = r[:, 4:7].copy().view(np.float32)
        cell = np.zeros(len(r), np.uint32)
        c = []
        for a in range(3):
            ext = bmax[a] - bmin[a]
            scale = np.float32(2 ** bits) / ext if ext > 0 else np.float32(0)
            lo = bmin[a] if ext > 0 else np.float32(0)
            with np.errstate(invalid="ignore"):
                v = (o[:, a] - lo) * scale
                v = np.where(np.isnan(v), np.float32(0), np.clip(v, 0, 2 ** bits - 1))   # fmaxf(NaN, 0) = 0
            c.append(v.astype(np.uint32))
        for b in range(bits):
            for a in range(3):
                cell |= ((c[a] >> b) & 1) << (3 * b + a)
        octant = (d[:, 0] < 0).astype(np.uint32) | ((d[:, 1] < 0).astype(np.uint32) << 1) | ((d[:, 2] < 0).astype(np.uint32) << 2)
        return (cell << 3) | octant
    for s_ in range(64):
        a, b = rays[s_], out[s_]
        assert len(a) == len(b) == counts[s_]
        order_a = np.lexsort(a.T[::-1]); order_b = np.lexsort(b.T[::-1])
        assert (a[order_a] == b[order_b]).all()                 # a permutation, every word of every record
        assert (keys[s_] == key_of(b)).all()                    # the device's key is the independent reading's
        assert (np.diff(keys[s_].astype(np.int64)) >= 0).all()  # sorted
    out_b, keys_b = E.wf_bin(n_chunks, counts, rays, bmin, bmax, stage=1)   # the occlusion stage's instantiation: the same sort
    assert all((np.sort(keys[s_]) == keys_b[s_]).all() for s_ in range(64))


def test_tile_kernel_on_random_scenes(tmp_path, built):
    """k_path_tiles itself (dynamic sample pairs, wave-aligned query passes, the instantiation with or without mis_ray_filter that
    tray_scene_create would pick) on scenes with every material kind: the oracle's samples, vertices, rays and image."""
    import json
    import _random_scenes as R
    d = str(tmp_path)
    for seed in (411, 412, 413, 414, 415, 416):
        p = R.write_random_scene(d, seed)
        desc = json.load(open(p))
        desc["film"].update(width=32, height=24, samples=8)
        json.dump(desc, open(p, "w"))
        scene, *_ = T.Scene.load_file(p)
        flat = scene.flatten(0)
        img, st = E.render_tiles(flat, tile_queue(32, 24), 8, seed, blocks=2)
        ref, ost = O.render_tiles(flat, 8, seed=seed)
        assert st[:3] == (ost.samples, ost.vertices, ost.rays), seed
        fin = np.isfinite(rgb(img)).all(axis=2) & np.isfinite(rgb(ref)).all(axis=2)
        assert fin.mean() > 0.99 and float(np.sqrt(np.mean((rgb(img)[fin] - rgb(ref)[fin]) ** 2))) < 2e-6, seed


def test_wavefront_schedule_on_random_scenes(tmp_path, built):
    """The wavefront schedule on scenes with every material kind (the rays it counts without tracing -- BSDF-sampled light rays that miss
    the light's primitive, occlusion rays whose BSDF value is black -- depend on the lobes), both light kinds and nested groups:
    the oracle's samples, vertices, rays and image."""
    import json
    import _random_scenes as R
    d = str(tmp_path)
    for seed in (401, 402, 403, 404, 405):
        p = R.write_random_scene(d, seed)
        desc = json.load(open(p))
        desc["film"].update(width=32, height=24, samples=4)
        json.dump(desc, open(p, "w"))
        scene, *_ = T.Scene.load_file(p)
        flat = scene.flatten(0)
        img, (samples, vertices, rays, rounds) = E.render_wavefront(flat, tile_queue(32, 24), 4, seed, trace=0, n_chunks=5, trace_blocks=2, lds_depth=4)
        ref, st = O.render_tiles(flat, 4, seed=seed)
        assert (samples, vertices, rays) == (st.samples, st.vertices, st.rays), seed
        fin = np.isfinite(rgb(img)).all(axis=2) & np.isfinite(rgb(ref)).all(axis=2)
        assert fin.mean() > 0.99 and float(np.sqrt(np.mean((rgb(img)[fin] - rgb(ref)[fin]) ** 2))) < 2e-6, seed


def test_tile_slices_are_the_same_samples(tmp_path, built, monkeypatch):
    """launch_tiles cuts tiles into slices of their samples when a launch has few tiles per workgroup (a GPU's share of the frame
    on an 8-GPU node): the slices of a tile are independent work items whose film contributions add up. Same samples, vertices and
    rays as the oracle's whole tiles, same image."""
    w, h, spp = 32, 24, 16
    scenes.write_assets(str(tmp_path), cornell=(w, h, spp), small=(w, h, spp))
    scene, *_ = T.Scene.load_file(str(tmp_path / "smallpt.json"))
    flat = scene.flatten(0)
    ref, st = O.render_tiles(flat, spp, seed=4)
    for slices in ("1", "3", "4", "5"):   # items per tile: whole tiles; 8 + 4 + 4 samples; 8 + 4 + 2 + 2; 8 + 4 + 2 + 1 + 1 (progressive, level-major)
        monkeypatch.setenv("TRAYHIP_TILE_SLICES", slices)
        img, s_ = E.render_tiles(flat, tile_queue(w, h), spp, 4, blocks=3)
        assert s_[:3] == (st.samples, st.vertices, st.rays)
        assert float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6


@pytest.fixture(scope="module")
def tr15_dir(tmp_path_factory):
    import pathlib
    d = tmp_path_factory.mktemp("emu_tr15_small")
    scenes.write_tr15_like_assets(str(d), film=(32, 24, 8), detail=0.02)
    return pathlib.Path(str(d))


WF_CASES = [("cornell_box", 0, 0), ("moving_box", 3, 0), ("tr15_like", 330, 0)]


@pytest.mark.parametrize("name,frame,trace", WF_CASES, ids=[n for n, _, t in WF_CASES])
def test_wavefront_schedule_emulated_as_simt(name, frame, trace, tmp_path, tr15_dir, built, monkeypatch):
    """The whole wavefront schedule -- k_wf_advance (film row bins, tile switch), k_wf_regen (camera samples, the per-path transform
    cache of moving scenes), the three traversal stages, k_wf_begin with its material sort, the kind-pure k_wf_query_kind, ray queues -- round
    after round until every tile is done, as fibers on the host: the oracle's samples, vertices, rays and image."""
    w, h, spp = 32, 24, 8
    d = str(tmp_path)
    if name == "tr15_like":
        tmp_path = tr15_dir                                                  # 59 instances, splines, keyed lights, MERL
    elif name == "moving_box":
        scenes.write_moving_box(d, width=w, height=h, samples=spp)
    else:
        scenes.write_assets(d, cornell=(w, h, spp), small=(w, h, spp))
    scene, *_ = T.Scene.load_file(str(tmp_path / (name + ".json")))
    flat = scene.flatten(frame)
    img, (samples, vertices, rays, rounds) = E.render_wavefront(flat, tile_queue(w, h), spp, 5, trace=trace, n_chunks=5, trace_blocks=2,
                                                                lds_depth=4)
    ref, st = O.render_tiles(flat, spp, seed=5)
    assert samples == st.samples == w * h * spp
    assert (vertices, rays) == (st.vertices, st.rays)      # moving scenes too: the spline stacks are evaluated bit for bit (dev_libm.h)
    assert rgb(ref).max() > 0.05
    assert float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6
    assert np.abs(img[..., 3] - ref[..., 3]).max() < 1e-4 * ref[..., 3].max()
    if flat.contents.animated:   # ... and with the frame's transform table (128-byte records by shutter-time index, indexed by the stage kernels): the same film in every bit
        monkeypatch.setenv("TRAYHIP_EMU_XF_TABLE", "1")
        img_t, st_t = E.render_wavefront(flat, tile_queue(w, h), spp, 5, trace=trace, n_chunks=5, trace_blocks=2, lds_depth=4)
        assert st_t == (samples, vertices, rays, rounds) and img_t.tobytes() == img.tobytes()


@pytest.mark.parametrize("slices", [2, 4])
def test_wavefront_schedule_with_tiles_cut_into_sample_slices(slices, tmp_path, built, monkeypatch):
    """launch_wavefront cuts the tiles' samples into work items when the pool has more chunks than the launch has tiles (round 4: the
    schedule's rate grows with the slots in flight). 12 tiles as 24 / 48 items over 17 chunks: two chunks of the same tile resolve their row
    bins separately, the (pixel, sample) pairs of a slice start at the slice's first sample -- same samples, vertices, rays, image."""
    w, h, spp = 32, 24, 8
    scenes.write_moving_box(str(tmp_path), width=w, height=h, samples=spp)
    scene, *_ = T.Scene.load_file(str(tmp_path / "moving_box.json"))
    flat = scene.flatten(3)
    monkeypatch.setenv("TRAYHIP_WF_SLICES", str(slices))
    img, (samples, vertices, rays, rounds) = E.render_wavefront(flat, tile_queue(w, h), spp, 5, n_chunks=17, trace_blocks=2, lds_depth=4)
    monkeypatch.delenv("TRAYHIP_WF_SLICES")
    ref, st = O.render_tiles(flat, spp, seed=5)
    assert samples == st.samples == w * h * spp and (vertices, rays) == (st.vertices, st.rays)
    assert float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6
    assert np.abs(img[..., 3] - ref[..., 3]).max() < 1e-4 * ref[..., 3].max()
    one, _ = E.render_wavefront(flat, tile_queue(w, h), spp, 5, n_chunks=17, trace_blocks=2, lds_depth=4)      # whole tiles: 12 of the 17 chunks busy
    assert np.abs(img - one).max() < 1e-5 * one.max()


def test_random_scene_sweep_of_the_per_sample_path(tmp_path, built):
    """60 random static scenes (tests/_random_scenes.py: all material kinds, spheres / disks / rectangles / meshes with texture
    coordinates, point and area lights, nested groups, every transform op, both filters, depths 0..10), 1000 camera samples
    each: the device source returns the oracle's bits. Every third scene has more than 16 instances and therefore goes through
    the two-level traversal; the others use the flat instance loop, which is exact too (dev_geom.h: trace_flat gates every instance
    by the box of its BVH<Instance> leaf and re-traces rays whose candidates tie the reference's way) -- round 1's build, without
    the gates, differed in 2 of 450 000 samples (occlusion rays grazing the wall they start on)."""
    import json
    import _random_scenes as R
    d = str(tmp_path)
    total = 0
    for seed in range(100, 160):
        p = R.write_random_scene(d, seed)
        many = seed % 3 == 0
        if many:
            desc = json.load(open(p))
            rng = np.random.default_rng(seed)
            for k in range(14):
                desc["objects"].append({"name": f"x{k}", "type": "receiver", "material": desc["materials"][k % len(desc["materials"])]["name"],
                                        "geometry": {"type": "sphere", "radius": float(rng.uniform(0.3, 1.2))},
                                        "transform": [{"type": "translate", "translation": [float(x) for x in rng.uniform([-12, 1, -12], [12, 20, 14])]}]})
            json.dump(desc, open(p, "w"))
        scene, *_ = T.Scene.load_file(p)
        flat = scene.flatten(0)
        assert (flat.contents.n_instances > 16) == many
        rng = np.random.default_rng(seed)
        n = 1000
        px = rng.integers(0, 64, n).astype(np.uint32); py = rng.integers(0, 48, n).astype(np.uint32); si = rng.integers(0, 8, n).astype(np.uint32)
        a = O.sample_radiance(flat, px, py, si, 8, seed=seed + 1)
        b = E.sample_radiance(flat, px, py, si, 8, seed + 1)
        same = (a == b).all(axis=1) | (np.isnan(a).any(axis=1) & np.isnan(b).any(axis=1))
        total += n
        assert same.all(), f"seed {seed}: {int((~same).sum())} of {n} samples differ"
    assert total == 60000


def test_tile_megakernel_of_moving_scenes_emulated_as_simt(tmp_path, built):
    """k_path_tiles<ANIM = 1>: the spline stacks of the moving instances are evaluated once per camera sample into the per-thread
    transform cache (xf_cache_fill) and read back by traversal, hit finishing and light sampling"""
    w, h, spp = 32, 24, 8
    scenes.write_moving_box(str(tmp_path), width=w, height=h, samples=spp)
    scene, *_ = T.Scene.load_file(str(tmp_path / "moving_box.json"))
    flat = scene.flatten(3)
    assert flat.contents.animated and flat.contents.n_instances <= 16
    img, (samples, vertices, rays, _) = E.render_tiles(flat, tile_queue(w, h), spp, 2, blocks=2)
    ref, st = O.render_tiles(flat, spp, seed=2)
    assert samples == st.samples and abs(vertices - st.vertices) <= 2e-3 * st.vertices and abs(rays - st.rays) <= 2e-3 * st.rays
    assert float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-3 and np.abs(img[..., 3] - ref[..., 3]).max() < 1e-4 * ref[..., 3].max()


def test_tile_megakernel_fills_its_cache_columns_from_the_transform_table(tmp_path, built, monkeypatch):
    """Round 5 / 6: launches of many samples read a moving scene's transforms from the frame's table over the 2^24 shutter-time indices
    (k_xf_table_build) -- camera_ray gathers the camera's record, xf_cache_fill_wave deals the (starting lane, moving instance) pairs of a
    step out to the whole wave, every lane copying one record into the column of the lane it works for. The emulation maps the table sparsely
    (records for the indices the frame's camera samples draw) and must render what per-sample evaluation renders: the same film in every bit
    (one schedule of the fibers, the same f32 sums), the same counts -- at one and at several workgroups, with whole and with sliced tiles."""
    w, h, spp = 32, 24, 8
    scenes.write_moving_box(str(tmp_path), width=w, height=h, samples=spp)
    scene, *_ = T.Scene.load_file(str(tmp_path / "moving_box.json"))
    for frame, blocks, slices in ((3, 2, None), (0, 1, "2")):
        flat = scene.flatten(frame)
        assert flat.contents.animated and flat.contents.camera.animated
        if slices: monkeypatch.setenv("TRAYHIP_TILE_SLICES", slices)
        monkeypatch.delenv("TRAYHIP_EMU_XF_TABLE", raising=False)
        img0, st0 = E.render_tiles(flat, tile_queue(w, h), spp, 5, blocks=blocks)
        monkeypatch.setenv("TRAYHIP_EMU_XF_TABLE", "1")
        img1, st1 = E.render_tiles(flat, tile_queue(w, h), spp, 5, blocks=blocks)
        assert st0 == st1 and st0[0] == w * h * spp
        assert img0.tobytes() == img1.tobytes()
        ref, st = O.render_tiles(flat, spp, seed=5)
        assert float(np.sqrt(np.mean((rgb(img1) - rgb(ref)) ** 2))) < 2e-3


def test_flat_loop_reproduces_the_samples_round_1_got_wrong(tmp_path, built):
    """The flat instance loop tests the box of the instance's BVH<Instance> leaf, as the reference's traversal does on the way to
    it (dev_geom.h: trace_flat). The two samples of the 300-scene sweep that differed in round 1's build (seeds 148 and 323: an
    occlusion ray grazing the wall it starts on) come out bit-identical, and so does everything else."""
    import _random_scenes as R
    d = str(tmp_path)
    for seed, (x, y, s_) in ((148, (32, 18, 6)), (323, (28, 5, 0))):
        scene, *_ = T.Scene.load_file(R.write_random_scene(d, seed))
        flat = scene.flatten(0)
        rng = np.random.default_rng(seed)
        n = 1500
        px = rng.integers(0, 64, n).astype(np.uint32); py = rng.integers(0, 48, n).astype(np.uint32); si = rng.integers(0, 8, n).astype(np.uint32)
        px[0], py[0], si[0] = x, y, s_
        a = O.sample_radiance(flat, px, py, si, 8, seed=seed + 1)
        assert a.tobytes() == E.sample_radiance(flat, px, py, si, 8, seed + 1).tobytes()
    w, h, spp = 32, 24, 8
    scenes.write_assets(d, cornell=(w, h, spp), small=(w, h, spp))
    scene, *_ = T.Scene.load_file(str(tmp_path / "cornell_box.json"))
    flat = scene.flatten(0)
    E.retraced()
    img, st = E.render_tiles(flat, tile_queue(w, h), spp, 7)      # with the cooperative small-mesh test behind the box test
    assert E.retraced() <= 2   # on cornell_box itself ties are (almost) unheard of: the gates decide, not the fallback
    ref, ost = O.render_tiles(flat, spp, seed=7)
    assert st[:3] == (ost.samples, ost.vertices, ost.rays) and float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6


def tied_scene(w, h, spp):
    """cornell_box with every wall doubled at the same place (a different material on the copy) and the short block doubled
    too: every ray that hits a wall or that block has two candidates with exactly the same t, and which of them the reference
    returns depends on the order of its BVH<Instance> / BVH<Triangle> traversal. The flat loop and the cooperative small-mesh
    test cannot know that order: they must notice the tie and hand the ray to the reference's traversal (trace_bvh)."""
    import copy
    d = scenes.cornell_box(w, h, spp)
    extra = []
    def walk(objs):
        for o in objs:
            if o.get("type") == "group":
                walk(o["objects"])
            elif o.get("type") == "receiver":
                c = copy.deepcopy(o)
                c["name"] = o["name"] + "_twin"
                c["material"] = next(m["name"] for m in d["materials"] if m["name"] != o["material"])
                extra.append((objs, c))
    walk(d["objects"])
    for objs, c in extra[:7]:
        objs.append(c)
    return d


def test_tied_candidates_are_resolved_by_the_reference_traversal(tmp_path, built):
    """hit records and whole samples on a scene made of coincident surfaces: bit-identical to the oracle, in the per-lane form
    (k_debug_*: flat loop + per-lane BVH<Triangle> traversal) and in the tile kernel (cooperative small-mesh test)"""
    import json
    w, h, spp = 32, 24, 8
    scenes.write_assets(str(tmp_path))
    json.dump(tied_scene(w, h, spp), open(tmp_path / "tied.json", "w"))
    scene, *_ = T.Scene.load_file(str(tmp_path / "tied.json"))
    flat = scene.flatten(0)
    assert 8 < flat.contents.n_instances <= 16
    rays = rays_for(flat, 11, 20000, 0, [0, 10, 0], 8.0)
    E.retraced()
    a, b = O.intersect(flat, rays), E.debug_intersect(flat, rays, O.HIT_DTYPE)
    assert (a["inst"] != 0xffffffff).mean() > 0.8
    assert E.retraced() > 0.8 * len(rays)   # nearly every ray ends on a doubled surface
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f], equal_nan=True), f
    rng = np.random.default_rng(12)
    n = 6000
    px = rng.integers(0, w, n).astype(np.uint32); py = rng.integers(0, h, n).astype(np.uint32); si = rng.integers(0, spp, n).astype(np.uint32)
    assert O.sample_radiance(flat, px, py, si, spp, seed=5).tobytes() == E.sample_radiance(flat, px, py, si, spp, 5).tobytes()
    img, st = E.render_tiles(flat, tile_queue(w, h), spp, 5, blocks=2)
    ref, ost = O.render_tiles(flat, spp, seed=5)
    assert st[:3] == (ost.samples, ost.vertices, ost.rays) and float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6


def test_sharded_launches_of_the_tile_kernel_partition_the_frame(tmp_path, built):
    """tray_render_shard_device's launch (chunks shard, shard + n, ... of the Morton queue; work item -> queue entry mapping inside
    k_path_tiles) for 3 ranks with 5-tile chunks: the shards render exactly the tiles tray_shard_tiles enumerates and add up to
    the whole frame -- the multi-GPU data path minus the RCCL sum."""
    from tray_rust_amd import multi
    w, h, spp = 48, 32, 4
    scenes.write_assets(str(tmp_path), cornell=(w, h, spp), small=(w, h, spp))
    scene, *_ = T.Scene.load_file(str(tmp_path / "cornell_box.json"))
    flat = scene.flatten(0)
    q = tile_queue(w, h)
    full, sf = E.render_tiles(flat, q, spp, 9, blocks=2)
    total = np.zeros_like(full)
    samples = 0
    for rank in range(3):
        part, sp_ = E.render_tiles(flat, q, spp, 9, blocks=2, shard=(rank, 3, 5))
        mine = multi.shard_tiles(len(q), rank, 3, 5)
        assert sp_[0] == len(mine) * 64 * spp
        touched = np.zeros((h, w), bool)
        for t in mine:
            x, y = int(q[t][0]), int(q[t][1])
            touched[max(0, y * 8 - 4):y * 8 + 13, max(0, x * 8 - 4):x * 8 + 13] = True     # tile + filter halo
        assert not part[~touched].any()
        total += part; samples += sp_[0]
    assert samples == sf[0] == w * h * spp
    assert np.abs(total - full).max() < 1e-4 * full.max()


@pytest.mark.parametrize("w,h,spp,max_depth", [(8, 8, 1, 0), (16, 8, 2, 1), (8, 16, 1, 5), (24, 8, 4, 0)])
def test_smallest_films_sample_counts_and_depths(w, h, spp, max_depth, tmp_path, built):
    """Edge sizes: one-tile films, 1 / 2 samples per pixel (waves of the tile kernel without a sample, pool slots without one),
    max_depth 0 (the path ends at its first vertex) -- both schedules against the oracle."""
    import json
    scenes.write_assets(str(tmp_path))
    for make in (scenes.cornell_box, scenes.smallpt):
        desc = make(w, h, spp)
        desc["integrator"] = {"type": "pathtracer", "min_depth": 0, "max_depth": max_depth}
        json.dump(desc, open(tmp_path / "x.json", "w"))
        scene, *_ = T.Scene.load_file(str(tmp_path / "x.json"))
        flat = scene.flatten(0)
        ref, st = O.render_tiles(flat, spp, seed=3)
        for img, s_ in (E.render_tiles(flat, tile_queue(w, h), spp, 3, blocks=2), E.render_wavefront(flat, tile_queue(w, h), spp, 3, trace=0, n_chunks=2)):
            assert s_[:3] == (st.samples, st.vertices, st.rays)
            assert float(np.sqrt(np.mean((rgb(img) - rgb(ref)) ** 2))) < 2e-6
