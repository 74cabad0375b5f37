"""TRAY-CBRNG v2 draws the six per-path shuffles of the LD arrays (integrator/path.rs:55-60, sampler/ld.rs:58,63: a fresh Fisher-Yates per
array) from a pool of 256 permutations per scene -- a change of the sampler's definition made for speed, in product
and oracle alike. The marginals of every (bounce, array) point are untouched by ANY permutation; what the choice shapes is how the points
of ONE path relate across its bounces. This file compares the pool with per-array Fisher-Yates shuffles (the oracle's ORC_FRESH_SHUFFLES
mode, test-only) where that would show: the joint second moments of the first bounces' sample values, and whole images."""
import ctypes as C

import numpy as np

import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O


def path_values(n_paths, n_bounces, fresh, seed):
    o = O.oracle()
    o.oracle_path_samples_mode.restype = None
    o.oracle_path_samples_mode.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    out = np.zeros((n_paths, n_bounces, 9), np.float32)
    rr = np.zeros(n_bounces, np.float32)
    k = 0
    for py in range(n_paths // 1024 + 1):
        for px in range(64):
            for s in range(16):
                if k == n_paths:
                    return out
                o.oracle_path_samples_mode(seed, 0, 64, px, py, s, n_bounces, fresh, out[k].ctypes.data, rr.ctypes.data)
                k += 1
    return out


def test_pool_and_fresh_shuffles_have_the_same_second_moments():
    """60 000 paths, arrays of 9 (max_depth 8): the 27 values of bounces 0..2 -- L2 / B2 / P2 (x, y), L1 / B1 / P1 each. Means, variances and the
    full 27 x 27 covariance (within a bounce: the arrays are independently scrambled; across bounces: the points of one array are
    strata of one (0, 2)-net, negatively correlated by construction) agree between the pool and per-array Fisher-Yates within the
    sampling error of 60 000 paths."""
    n, nb = 60000, 9
    pool = path_values(n, nb, 0, 11)[:, :3, :].reshape(n, 27).astype(np.float64)
    fresh = path_values(n, nb, 1, 11)[:, :3, :].reshape(n, 27).astype(np.float64)
    other = path_values(n, nb, 0, 12)[:, :3, :].reshape(n, 27).astype(np.float64)      # the pool under another seed: the yardstick
    # the two modes differ in the permutations only: same scrambles, so each path holds the same SET of 9 points per array
    assert not np.array_equal(pool, fresh)
    for x in (pool, fresh):
        assert np.abs(x.mean(axis=0) - 0.5).max() < 5 * np.sqrt(1 / 12 / n)
        assert np.abs(x.var(axis=0) - 1 / 12).max() < 2e-3
    cp, cf, co = np.cov(pool.T), np.cov(fresh.T), np.cov(other.T)
    sigma = (1 / 12) / np.sqrt(n)                      # sampling error of a covariance of two U(0, 1)-like variables
    yard = np.abs(cp - co).max()                       # pool against pool under another seed: pure sampling error
    assert yard < 6 * sigma
    assert np.abs(cp - cf).max() < max(1.5 * yard, 6 * sigma), (np.abs(cp - cf).max(), yard, sigma)
    # across bounces of the same array the strata show up, and equally in both modes (x coordinate of P2 at bounces 0 and 1)
    i0, i1 = 0 * 9 + 4, 1 * 9 + 4
    assert cp[i0, i1] < -3 * sigma and cf[i0, i1] < -3 * sigma and abs(cp[i0, i1] - cf[i0, i1]) < 6 * sigma
    # different arrays never correlate, same or different bounce (B2.x at bounce 0 against P2.x at bounce 1, L1 against P1 ...)
    for a, b in ((0 * 9 + 2, 1 * 9 + 4), (0 * 9 + 6, 0 * 9 + 8), (1 * 9 + 0, 2 * 9 + 7)):
        assert abs(cp[a, b]) < 6 * sigma and abs(cf[a, b]) < 6 * sigma


def test_images_under_pool_and_fresh_shuffles_agree_within_monte_carlo_error(tmp_path, built):
    """cornell_box 64 x 48 at 64 spp: the image with per-array Fisher-Yates shuffles lies as close to the pool's image as the pool's image
    under another seed does (two-seed Monte-Carlo error), and the frame's mean radiance agrees"""
    import json
    import os
    p = os.path.join(str(tmp_path), "c.json")
    scenes.write_assets(str(tmp_path))
    json.dump(scenes.cornell_box(64, 48, 64), open(p, "w"))
    scene, *_ = T.Scene.load_file(p)
    flat = scene.flatten(0)

    def rgb(img):
        return img[..., :3] / np.maximum(img[..., 3:], 1e-20)
    a, sa = O.render_tiles(flat, 64, seed=1)
    b, sb = O.render_tiles(flat, 64, seed=2)
    f, sf = O.render_tiles(flat, 64, seed=1, flags=O.FRESH_SHUFFLES)
    assert sa.samples == sf.samples and sa.vertices != sf.vertices          # another sampler: other paths
    two_seed = float(np.sqrt(np.mean((rgb(a) - rgb(b)) ** 2)))
    pool_vs_fresh = float(np.sqrt(np.mean((rgb(a) - rgb(f)) ** 2)))
    assert 0.3 * two_seed < pool_vs_fresh < 1.25 * two_seed, (pool_vs_fresh, two_seed)
    assert abs(rgb(a).mean() - rgb(f).mean()) < 3 * abs(rgb(a).mean() - rgb(b).mean()) + 2e-3
    assert abs(sa.vertices / sa.samples - sf.vertices / sf.samples) < 0.02
