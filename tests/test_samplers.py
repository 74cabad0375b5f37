"""The Samplers thread_work does not construct -- sampler/uniform.rs and sampler/adaptive.rs -- behind
tray_scene_set_sampler: the oracle's restatement (oracle.cpp: UniformSampler / AdaptiveSampler driven by the same generic thread_work as
LowDiscrepancy) against an independent numpy reading of adaptive.rs's decision rule and against analytic properties, and the DEVICE code
(k_sampler_pass / k_sampler_decide, compiled for the host in tests/emu) against the oracle, bit for bit. GPU counterparts:
tests/test_gpu_parity.py::test_gpu_other_samplers_*."""
import numpy as np
import pytest

import tray_rust_amd as T
from tray_rust_amd import scenes
import _emu as E
import _oracle as O

f32 = np.float32


def luminance(c):       # Colorf::luminance (film/color.rs:43-45), evaluated left to right in f32
    return f32(f32(f32(0.2126) * c[0]) + f32(f32(0.7152) * c[1])) + f32(f32(0.0722) * c[2])


def adaptive_reference(lum, min_spp, max_spp):
    """adaptive.rs:92-143 read on its own: how many of the grey samples lum[...] a pixel consumes (0: more than given)"""
    def npot(v):
        p = 1
        while p < v:
            p *= 2
        return p
    lo, hi = npot(min_spp), npot(max_spp)
    step = npot((hi - lo) // 5)
    grey = [luminance((f32(v), f32(v), f32(v))) for v in lum]
    taken, avg = 0, f32(0)
    while True:
        taken += lo if taken == 0 else step                              # get_samples
        if taken > len(lum):
            return 0
        s = grey[:taken]                                                  # &block_samples[pixel_samples..]
        if taken >= hi:                                                   # report_results: the rule is not consulted
            return taken
        if taken == lo:
            acc = f32(0)
            for v in s:
                acc = f32(acc + v)
            avg = f32(acc / f32(len(s)))
        else:
            for i in range(len(s) - step, len(s)):
                avg = f32(f32(s[i] + f32(f32(i - 1) * avg)) / f32(i))
        with np.errstate(divide="ignore", invalid="ignore"):
            if not any(f32(abs(f32(v - avg))) / avg > f32(0.5) for v in s):
                return taken


def test_adaptive_parameters_round_like_the_reference():
    # adaptive.rs:36-48: both bounds to powers of two (0 -> 1), step = ((max - min) / 5).next_power_of_two()
    for lo, hi, want in [(4, 32, (4, 32, 8)), (3, 33, (4, 64, 16)), (1, 1, (1, 1, 1)), (0, 5, (1, 8, 1)), (8, 64, (8, 64, 16)), (16, 16, (16, 16, 1)),
                         (2, 1024, (2, 1024, 256))]:
        assert O.adaptive_params(lo, hi) == want
        assert int(T.lib().tray_adaptive_step(lo, hi)) == want[2]
        a = T.sampler.Adaptive((8, 8), lo, hi)
        assert (a.min_spp, a.max_spp(), a.step_size) == want and a.dimensions() == (8, 8)
    assert T.sampler.Uniform((8, 8)).max_spp() == 1 and T.sampler.LowDiscrepancy((8, 8), 5).max_spp() == 8
    with pytest.raises(ValueError):
        T.sampler.Adaptive((8, 8), 64, 4)


def test_adaptive_decision_rule_against_an_independent_reading():
    """Random grey sequences -- flat ones, ones with an outlier early / late, black pixels (the 0 / 0 of the contrast test is NaN: no more
    samples) and pixels whose average is 0 apart from one sample -- through Adaptive as thread_work drives it, against adaptive.rs read
    directly in numpy f32, incl. the (i - 1) / i weighting of the running average."""
    rng = np.random.default_rng(5)
    seen = set()
    for case in range(600):
        lo, hi = [(4, 32), (1, 16), (8, 64), (2, 4), (16, 16)][case % 5]
        n = 160
        kind = case % 7
        if kind == 0:
            lum = np.full(n, rng.uniform(0.05, 1.0))
        elif kind == 1:
            lum = rng.uniform(0.4, 0.6, n)
        elif kind == 2:
            lum = rng.uniform(0.0, 1.0, n)
        elif kind == 3:
            lum = np.full(n, 0.5); lum[rng.integers(0, 12)] = rng.uniform(0.0, 1.0)
        elif kind == 4:
            lum = np.zeros(n)
        elif kind == 5:
            lum = np.zeros(n); lum[rng.integers(0, 8)] = 0.8
        else:
            lum = np.full(n, 0.3); lum[1] = 0.55; lum[max(lo, 2):] = rng.uniform(0.42, 0.48, n - max(lo, 2))     # an outlier the later samples' average catches up with
        got, want = O.adaptive_samples_for(lum, lo, hi), adaptive_reference(lum.astype(f32), lo, hi)
        assert got == want, (case, lo, hi, got, want)
        seen.add((lo, hi, got))
    # the cases reach the minimum, intermediate counts and the overshoot past max_spp (4 + 4 * 8 = 36 > 32)
    assert (4, 32, 4) in seen and (4, 32, 36) in seen and any(lo == 4 and 4 < g < 36 for lo, _, g in seen) and (16, 16, 16) in seen


@pytest.fixture(scope="module")
def cornell(tmp_path_factory, built):
    d = tmp_path_factory.mktemp("samplers")
    scenes.write_assets(str(d), cornell=(64, 48, 16), small=(64, 48, 16))
    out = {}
    for name in ("cornell_box", "smallpt"):
        scene, *_ = T.Scene.load_file(str(d / (name + ".json")))
        out[name] = (scene, scene.flatten(0))
    return out


def test_uniform_takes_one_centred_sample_per_pixel(cornell):
    flat = cornell["cornell_box"][1]
    img, st, counts = O.render_tiles_sampler(flat, O.SAMPLER_UNIFORM, seed=3)
    assert (counts == 1).all() and st.samples == 64 * 48
    img1, st1, _ = O.render_tiles_sampler(flat, O.SAMPLER_UNIFORM, seed=3, threads=1)
    assert np.array_equal(img, img1)
    # every sample sits at a pixel centre: the weight plane is the film's filter summed over a regular grid, the same at every
    # interior pixel
    w = img[8:-8, 8:-8, 3]
    assert w.min() > 0 and np.ptp(w) < 1e-5 * w.mean()
    # a one-sample image is the LowDiscrepancy image plus noise: same mean colour within a few per cent
    ld, _ = O.render_tiles(flat, 16, seed=3)
    a = img[..., :3].sum(axis=(0, 1)) / img[..., 3].sum(); b = ld[..., :3].sum(axis=(0, 1)) / ld[..., 3].sum()
    assert np.abs(a - b).max() < 0.08 * b.max(), (a, b)
    other, _, _ = O.render_tiles_sampler(flat, O.SAMPLER_UNIFORM, seed=4)
    assert not np.array_equal(img, other) and np.array_equal(img[..., 3], other[..., 3])      # other numbers, same positions


def test_adaptive_spends_its_samples_where_the_contrast_is(cornell):
    flat = cornell["cornell_box"][1]
    lo, hi, step = O.adaptive_params(4, 32)
    img, st, counts = O.render_tiles_sampler(flat, O.SAMPLER_ADAPTIVE, 4, 32, seed=2)
    allowed = {lo + k * step for k in range(0, 5)}
    assert set(np.unique(counts)) <= allowed and counts.sum() == st.samples
    assert (counts == lo).any() and (counts > lo).mean() > 0.2 and counts.max() == 36      # 4 + 4 * 8: past max_spp, as the reference
    # the weight plane follows the counts: more filter weight where more samples were taken
    w = img[..., 3]
    assert np.corrcoef(w[4:-4, 4:-4].ravel(), counts[4:-4, 4:-4].ravel())[0, 1] > 0.6
    # pixels that see the light source directly saturate (clamped to 1, quirk Q3) and are flat; the penumbra / edges are not
    img1, _, counts1 = O.render_tiles_sampler(flat, O.SAMPLER_ADAPTIVE, 4, 32, seed=2, threads=1)
    assert np.array_equal(img, img1) and np.array_equal(counts, counts1)
    # min_spp == max_spp: every pixel takes exactly that many (samples_taken >= max_spp ends the pixel, adaptive.rs:136)
    _, st16, c16 = O.render_tiles_sampler(flat, O.SAMPLER_ADAPTIVE, 16, 16, seed=2)
    assert (c16 == 16).all() and st16.samples == 16 * 64 * 48
    # the image converges to LowDiscrepancy's
    ld, _ = O.render_tiles(flat, 64, seed=9)
    a = img[..., :3] / np.maximum(img[..., 3:], 1e-20); b = ld[..., :3] / np.maximum(ld[..., 3:], 1e-20)
    assert np.sqrt(np.mean((a - b) ** 2)) < 0.08


def test_tile_ranges_of_the_other_samplers_add_up(cornell):
    flat = cornell["smallpt"][1]
    for kind, args in ((O.SAMPLER_UNIFORM, (1, 1)), (O.SAMPLER_ADAPTIVE, (2, 16))):
        whole, st, _ = O.render_tiles_sampler(flat, kind, *args, seed=6)
        a, sa, _ = O.render_tiles_sampler(flat, kind, *args, seed=6, tile_start=0, tile_count=20)
        b, sb, _ = O.render_tiles_sampler(flat, kind, *args, seed=6, tile_start=20, tile_count=0)
        assert sa.samples + sb.samples == st.samples
        np.testing.assert_allclose(a + b, whole, rtol=0, atol=2e-5)


@pytest.mark.parametrize("name", ["cornell_box", "smallpt"])
@pytest.mark.parametrize("kind,args", [(O.SAMPLER_UNIFORM, (1, 1)), (O.SAMPLER_ADAPTIVE, (4, 32)), (O.SAMPLER_ADAPTIVE, (1, 8)), (O.SAMPLER_ADAPTIVE, (8, 8))])
def test_device_code_of_the_other_samplers_is_bit_identical(cornell, name, kind, args):
    """k_sampler_pass + k_sampler_decide (host emulation: same libm as the oracle) over a third of the film's tiles, in one batch and in
    batches of five tiles: same samples, same decisions, same sample totals; the films agree to the last bits of the splat order."""
    flat = cornell[name][1]
    q = np.array(list(T.BlockQueue((64, 48))), np.uint32)[7:23]
    ref = np.zeros((48, 64, 4), f32)
    total = 0
    for t in range(7, 23):
        r, st, _ = O.render_tiles_sampler(flat, kind, *args, seed=8, tile_start=t, tile_count=1, threads=1)
        ref += r; total += st.samples
    for batch in (0, 5):
        img, (samples, vertices, rays) = E.render_sampler(flat, q, kind, *args, seed=8, batch_tiles=batch)
        assert samples == total
        assert np.array_equal(img[..., 3] > 0, ref[..., 3] > 0)
        np.testing.assert_allclose(img, ref, rtol=5e-5, atol=5e-5)      # (summation order of the splats)


def test_sampler_pass_groups_of_tiles_that_are_not_squares(cornell, monkeypatch):
    """k_sampler_pass hands a GROUP of consecutive tiles to a workgroup and keeps the group's film in one LDS window when the tiles form a square of
    at most 32 x 32 pixels; any other group is walked tile by tile. A shuffled tile queue in groups of 3 and of 16 (tiles far apart: every group takes
    the tile-by-tile path) and the Z-order queue in groups of 16 (squares): the oracle's sample totals and film under Adaptive and Uniform."""
    flat = cornell["cornell_box"][1]
    q = np.array(list(T.BlockQueue((64, 48))), np.uint32)
    shuffled = q[np.random.default_rng(4).permutation(len(q))]
    for kind, args in ((O.SAMPLER_ADAPTIVE, (4, 16)), (O.SAMPLER_UNIFORM, ())):
        ref, st, _ = O.render_tiles_sampler(flat, kind, *args, seed=8)
        for queue, group in ((shuffled, "3"), (shuffled, "16"), (q, "16")):
            monkeypatch.setenv("TRAYHIP_SAMPLER_GROUP", group)
            img, (samples, vertices, rays) = E.render_sampler(flat, queue, kind, *args, seed=8)
            assert samples == st.samples and vertices == st.vertices
            np.testing.assert_allclose(img, ref, rtol=5e-5, atol=5e-5)


def test_whitted_under_the_other_samplers(tmp_path, built):
    """the one-element arrays of Whitted's activations (whitted.rs:46-47, mod.rs:59-60): index samples_taken under Adaptive, plain draws under
    Uniform -- device code against the oracle"""
    import json
    scenes.write_assets(str(tmp_path), cornell=(32, 24, 4), small=(32, 24, 4))
    doc = json.load(open(tmp_path / "smallpt.json"))
    doc["integrator"]["type"] = "whitted"
    (tmp_path / "w.json").write_text(json.dumps(doc))
    scene, *_ = T.Scene.load_file(str(tmp_path / "w.json"))
    flat = scene.flatten(0)
    q = np.array(list(T.BlockQueue((32, 24))), np.uint32)
    for kind, args in ((O.SAMPLER_UNIFORM, (1, 1)), (O.SAMPLER_ADAPTIVE, (2, 8))):
        ref, st, _ = O.render_tiles_sampler(flat, kind, *args, seed=5, threads=1)
        img, (samples, _, _) = E.render_sampler(flat, q, kind, *args, seed=5)
        assert samples == st.samples
        np.testing.assert_allclose(img, ref, rtol=5e-5, atol=5e-5)      # (summation order of the splats)
        ld, _ = O.render_tiles(flat, 8, seed=5)
        assert np.abs(img[..., :3].sum() / img[..., 3].sum() - ld[..., :3].sum() / ld[..., 3].sum()) < 0.05


def test_device_code_of_adaptive_on_a_moving_scene(tmp_path, built):
    """a moving instance, a moving camera: the per-use spline evaluation of the ANIM = 2 instantiation, times from the round's max_spp-long array"""
    scene, *_ = T.Scene.load_file(scenes.write_moving_box(str(tmp_path), width=48, height=32, samples=4))
    flat = scene.flatten(1)
    q = np.array(list(T.BlockQueue((48, 32))), np.uint32)
    ref, st, counts = O.render_tiles_sampler(flat, O.SAMPLER_ADAPTIVE, 2, 16, seed=3, threads=1)
    img, (samples, _, _) = E.render_sampler(flat, q, O.SAMPLER_ADAPTIVE, 2, 16, seed=3)
    assert samples == st.samples and counts.max() > 2
    np.testing.assert_allclose(img, ref, rtol=5e-5, atol=5e-5)


def test_oracle_reproduces_the_rank_4_golden(tmp_path, built):
    """tests/golden/rank4_48x32_seed9.npz (make_golden.py): the AnimatedMesh scene under LowDiscrepancy, cornell_box under Uniform and Adaptive"""
    import json
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rank4_48x32_seed9.npz"))
    scene, *_ = T.Scene.load_file(scenes.write_waving_flag(str(tmp_path), grid=12, n_keys=4, width=48, height=32, samples=16))
    img, st = O.render_tiles(scene.flatten(int(g["flag_frame"])), 16, seed=9)
    assert st.vertices == int(g["flag_vertices"]) and st.rays == int(g["flag_rays"])
    assert np.abs(img - g["flag"]).max() <= 1e-5 * np.abs(g["flag"]).max()
    scenes.write_assets(str(tmp_path))
    p = os.path.join(str(tmp_path), "s.json")
    json.dump(scenes.cornell_box(48, 32, 16), open(p, "w"))
    cornell, *_ = T.Scene.load_file(p)
    flat = cornell.flatten(0)
    uni, st_u, _ = O.render_tiles_sampler(flat, O.SAMPLER_UNIFORM, seed=9)
    ada, st_a, counts = O.render_tiles_sampler(flat, O.SAMPLER_ADAPTIVE, 4, 32, seed=9)
    assert st_u.vertices == int(g["uniform_vertices"]) and st_a.samples == int(g["adaptive_samples"]) and np.array_equal(counts, g["adaptive_counts"])
    assert np.abs(uni - g["uniform"]).max() <= 1e-5 * np.abs(g["uniform"]).max() and np.abs(ada - g["adaptive"]).max() <= 1e-5 * np.abs(g["adaptive"]).max()


def test_device_code_of_adaptive_behind_bvh_of_instances(tmp_path, built):
    """59 instances, moving, many meshes (the C5 stand-in at low detail): the sampler kernels run the reference's two-level traversal there
    (trace_bvh), whatever schedule LowDiscrepancy renders of the scene use"""
    p = scenes.write_tr15_like_assets(str(tmp_path), film=(32, 24, 4), detail=0.02)
    scene, *_ = T.Scene.load_file(p if isinstance(p, str) else p[0])
    flat = scene.flatten(330)
    assert flat.contents.n_instances > 16
    q = np.array(list(T.BlockQueue((32, 24))), np.uint32)
    ref, st, counts = O.render_tiles_sampler(flat, O.SAMPLER_ADAPTIVE, 2, 8, seed=4, threads=1)
    img, (samples, _, _) = E.render_sampler(flat, q, O.SAMPLER_ADAPTIVE, 2, 8, seed=4)
    assert samples == st.samples and counts.max() > 2
    np.testing.assert_allclose(img, ref, rtol=5e-5, atol=5e-5)
