"""Known-answer tests pinning the oracle: the reference's own 17 unit tests restated (they are the
only vectors the reference holds: linalg + partition), plus sampler / RNG properties.
The hot path itself is parity-unpinned by the reference (oracle/oracle_math.hpp header)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O


def f32(*v):
    return np.array(v, dtype=np.float32)


def mat(*rows):
    return np.array(rows, dtype=np.float32).reshape(16)


def matop(name, a, b):
    out = np.zeros(16, dtype=np.float32)
    getattr(O.oracle(), name)(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


IDENT = mat(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1)


def transform(kind, v=None, angle=0.0):
    m, i = np.zeros(16, np.float32), np.zeros(16, np.float32)
    vv = None if v is None else f32(*v)
    O.oracle().oracle_transform(kind, None if vv is None else vv.ctypes.data, C.c_float(angle), m.ctypes.data, i.ctypes.data)
    return m, i


def apply(t, what, v):
    out = np.zeros(3, np.float32)
    vv = f32(*v)
    O.oracle().oracle_transform_apply(t[0].ctypes.data, t[1].ctypes.data, what, vv.ctypes.data, out.ctypes.data)
    return out


POINT, VECTOR, NORMAL = 0, 1, 2


# ---- src/linalg/mod.rs:130-143
def test_cross():
    out = np.zeros(3, np.float32)
    a, b = f32(1, 0, 0), f32(0, 1, 0)
    O.oracle().oracle_cross(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    assert (out == f32(0, 0, 1)).all()


def test_dot():
    a, b = f32(1, 2, 3), f32(4, 5, 6)
    assert O.oracle().oracle_dot(a.ctypes.data, b.ctypes.data) == np.float32(1 * 4 + 2 * 5 + 3 * 6)


# ---- src/linalg/matrix4.rs:266-305 (literal answers)
def test_matrix_add():
    a = IDENT.copy(); a[1] = 1
    b = IDENT.copy(); b[2 * 4 + 3] = 3
    c = mat(2, 1, 0, 0, 0, 2, 0, 0, 0, 0, 2, 3, 0, 0, 0, 2)
    assert (matop("oracle_mat4_add", a, b) == c).all()


def test_matrix_sub():
    a = IDENT.copy(); a[1] = 1
    b = IDENT.copy(); b[2 * 4 + 3] = 3
    c = mat(0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, -3, 0, 0, 0, 0)
    assert (matop("oracle_mat4_sub", a, b) == c).all()


def test_matrix_mul():
    assert (matop("oracle_mat4_mul", IDENT, IDENT) == IDENT).all()
    a = mat(1, 2, 1, 0, 3, 1, 4, 2, 1, 2, -5, 4, 3, 2, 4, 1)
    b = mat(8, 0, 2, 3, -2, 1, 0, 1, 5, -2, 3, 1, 0, 0, 4, 1)
    c = mat(9, 0, 5, 6, 42, -7, 26, 16, -21, 12, 3, 4, 40, -6, 22, 16)
    assert (matop("oracle_mat4_mul", a, b) == c).all()


def test_matrix_inverse_matches_numpy():
    rng = np.random.default_rng(1)
    for _ in range(20):
        a = rng.uniform(-2, 2, 16).astype(np.float32)
        a[12:] = (0, 0, 0, 1)
        out = np.zeros(16, np.float32)
        O.oracle().oracle_mat4_inverse(a.ctypes.data, out.ctypes.data)
        ref = np.linalg.inv(a.reshape(4, 4).astype(np.float64))
        assert np.allclose(out.reshape(4, 4), ref, rtol=2e-4, atol=2e-5)


# ---- src/linalg/transform.rs:285-380
def test_mult_sanity():
    t = transform(0)
    for what in (POINT, VECTOR, NORMAL):
        assert (apply(t, what, (1, 2, 3)) == f32(1, 2, 3)).all()


def test_translate():
    t = transform(1, (1, 2, 3))
    assert (apply(t, POINT, (1, 2, 3)) == f32(2, 4, 6)).all()
    assert (apply(t, VECTOR, (1, 2, 3)) == f32(1, 2, 3)).all()
    assert (apply(t, NORMAL, (1, 2, 3)) == f32(1, 2, 3)).all()


def test_scale():
    t = transform(2, (1, 2, 3))
    assert (apply(t, POINT, (1, 2, 3)) == f32(1, 4, 9)).all()
    assert (apply(t, VECTOR, (1, 2, 3)) == f32(1, 4, 9)).all()
    # normals transform by the inverse transpose
    assert np.allclose(apply(t, NORMAL, (1, 2, 3)), f32(1, 1, 1), atol=1e-6)


@pytest.mark.parametrize("kind,axis_in,plus,minus", [
    (3, (0, 1, 0), (0, 0, 1), (0, 0, -1)),    # rotate_x: +y -> +z at 90 deg
    (4, (0, 0, 1), (1, 0, 0), (-1, 0, 0)),    # rotate_y: +z -> +x
    (5, (1, 0, 0), (0, 1, 0), (0, -1, 0)),    # rotate_z: +x -> +y
])
def test_rotate_axes(kind, axis_in, plus, minus):
    assert np.allclose(apply(transform(kind, None, 90.0), VECTOR, axis_in), plus, atol=1e-4)
    assert np.allclose(apply(transform(kind, None, -90.0), VECTOR, axis_in), minus, atol=1e-4)
    assert np.allclose(apply(transform(kind, None, 90.0), NORMAL, axis_in), plus, atol=1e-4)


def test_rotate_axis_matches_rotate_xyz():
    for kind, axis in ((3, (1, 0, 0)), (4, (0, 1, 0)), (5, (0, 0, 1))):
        for ang in (30.0, -75.0, 90.0):
            a, b = transform(kind, None, ang), transform(6, axis, ang)
            assert np.allclose(a[0], b[0], atol=1e-6) and np.allclose(a[1], b[1], atol=1e-6)


# ---- src/linalg/point.rs:166, vector.rs / normal.rs trivia
def test_distance_sqr():
    a, b = f32(0, 0, 0), f32(3, 4, 0)
    d = b - a
    assert O.oracle().oracle_dot(d.ctypes.data, d.ctypes.data) == np.float32(25)


# ---- sampler/ld.rs
def test_van_der_corput_is_bit_reversal():
    o = O.oracle()
    for n in (0, 1, 2, 3, 5, 255, 1023):
        expect = int(f"{n:032b}"[::-1], 2) >> 8
        assert o.oracle_van_der_corput(n, 0) == np.float32(min(expect / 2 ** 24, 1 - 2 ** -23))


def test_02_sequence_is_stratified():
    """A (0,2)-sequence: every elementary interval of area 1/16 holds exactly one of the first 16 points,
    for any scramble."""
    o = O.oracle()
    for sx, sy in ((0, 0), (0x9e3779b9, 0x7feb352d), (123456789, 987654321)):
        pts = np.array([(o.oracle_van_der_corput(i, sx), o.oracle_sobol(i, sy)) for i in range(16)])
        for a, b in ((16, 1), (8, 2), (4, 4), (2, 8), (1, 16)):
            cells = (np.floor(pts[:, 0] * a) * b + np.floor(pts[:, 1] * b)).astype(int)
            assert len(set(cells)) == 16


def test_permute_is_a_bijection():
    o = O.oracle()
    for l, key in ((1024, 12345), (64, 0xdeadbeef), (9, 77), (1, 5)):
        vals = sorted(o.oracle_permute(i, l, key) for i in range(l))
        assert vals == list(range(l))
    assert [o.oracle_permute(i, 1024, 1) for i in range(8)] != [o.oracle_permute(i, 1024, 2) for i in range(8)]


def test_shuffle_small_is_a_uniformish_permutation():
    o = O.oracle()
    counts = np.zeros((9, 9))
    for key in range(4000):
        buf = (C.c_uint8 * 16)()
        o.oracle_shuffle_small(o.oracle_mix32(key), 9, buf)
        p = list(buf[:9])
        assert sorted(p) == list(range(9))
        for pos, v in enumerate(p):
            counts[pos, v] += 1
    assert np.abs(counts / 4000 - 1 / 9).max() < 0.03


def test_pixel_samples_cover_the_pixel_and_are_seed_dependent():
    o = O.oracle()
    out = np.zeros(3, np.float32)
    pts = []
    for s in range(64):
        o.oracle_pixel_sample(1, 0, 400, 17, 33, s, 64, out.ctypes.data)
        assert 17 <= out[0] < 18 and 33 <= out[1] < 34 and 0 <= out[2] < 1
        pts.append(tuple(out[:2]))
    assert len(set(pts)) == 64
    # stratified over the pixel in both axes
    assert len({int((x - 17) * 64) for x, _ in pts}) == 64
    o.oracle_pixel_sample(2, 0, 400, 17, 33, 0, 64, out.ctypes.data)
    assert tuple(out[:2]) != pts[0]


def test_path_sample_arrays_are_shuffled_02_points():
    o = O.oracle()
    n = 9
    out, rr = np.zeros(9 * n, np.float32), np.zeros(n, np.float32)
    o.oracle_path_samples(5, 0, 400, 3, 4, 7, n, out.ctypes.data, rr.ctypes.data)
    out = out.reshape(n, 9)
    assert ((out >= 0) & (out < 1)).all() and ((rr >= 0) & (rr < 1)).all()
    for col in range(9):
        assert len(set(out[:, col])) == n   # a permutation of distinct sequence points
