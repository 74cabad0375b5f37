// Device-side AnimatedTransform / AnimatedColor: what a ray with its own shutter time needs from a moving scene.
//   AnimatedTransform::transform   linalg/animated_transform.rs:40-56   (object-first stack of B-splines of TRS keyframes)
//   Keyframe::transform/interpolate linalg/keyframe.rs:60-73            quaternion slerp / to_matrix  quaternion.rs:67-113
//   Transform ops                   linalg/transform.rs:24-56,191-197   Matrix4::inverse matrix4.rs:48-172
//   AnimatedColor::color            film/animated_color.rs:52-78
//   BSpline::point                  bspline 0.2.2 (crates.io, not vendored by the reference): upper-bound span search
//                                   clamped to [degree, n_knots - degree - 1], iterative de Boor
// eval_xform_stack returns rows 0..2 of mat and of inv (row 3 of a TRS product is (0,0,0,1)). It needs ~100 registers for
// the 4x4 products and the general inverse, so the hot loops never contain it: they read the per-path cache that the start
// of a camera sample fills (dev_geom.h). It is inlined there (a call gives the calling kernel the callee's full register
// allocation: measured 112 vs 128 Msamples/s on the moving_box scene).
#pragma once
#include "../../../include/trayhip.h"
#include "dev_math.h"
#include "dev_libm.h"

namespace tr {

#ifndef TR_ANIM_EVAL
#define TR_ANIM_EVAL __device__ __forceinline__
#endif

struct DevKey {
    f3 t;
    float q[4];
    f3 s;
};

TR_DEV float quat_dot(const float a[4], const float b[4]) { return (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) + a[3] * b[3]; }

TR_DEV DevKey key_load(const TrayKeyframe* __restrict__ k) {
    DevKey r;
    r.t = mk(k->translation[0], k->translation[1], k->translation[2]);
    r.q[0] = k->rotation[0]; r.q[1] = k->rotation[1]; r.q[2] = k->rotation[2]; r.q[3] = k->rotation[3];
    r.s = mk(k->scaling[0], k->scaling[1], k->scaling[2]);
    return r;
}

// keyframe.rs:66-73 with quaternion.rs:101-113
TR_DEV DevKey key_interpolate(const DevKey& a, const DevKey& b, float t) {
    DevKey r;
    r.t = (1.0f - t) * a.t + t * b.t;
    r.s = (1.0f - t) * a.s + t * b.s;
    float cos_theta = quat_dot(a.q, b.q);
    if (cos_theta > 0.9995f) {
        float q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (1.0f - t) * a.q[i] + t * b.q[i];
        float len = sqrtf(quat_dot(q, q));
#pragma unroll
        for (int i = 0; i < 4; ++i) r.q[i] = q[i] / len;
    } else {
        // acos / cos / sin as the host libm the reference (and the oracle) calls computes them, bit for bit (ref_acosf / ref_sincosf above)
        float theta = ref_acosf(clampf(cos_theta, -1.0f, 1.0f));
        float theta_t = theta * t;
        float perp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) perp[i] = b.q[i] - a.q[i] * cos_theta;
        float len = sqrtf(quat_dot(perp, perp));
        float c, sn;
        ref_sincosf2(theta_t, sn, c);   // (cos and sin of one angle: reduction and squares shared; bit for bit the two calls, tools/libm_port_check.cpp)
#pragma unroll
        for (int i = 0; i < 4; ++i) r.q[i] = a.q[i] * c + (perp[i] / len) * sn;
    }
    return r;
}

// BSpline::point for degree <= 3 (the loader's default is 3; tray_scene_create rejects moving levels of higher degree)
TR_DEV DevKey spline_point(const TrayKeyframe* __restrict__ kfs, const float* __restrict__ knots, uint32_t n_knots, uint32_t degree, float t) {
    uint32_t first = 0u;
    int count = (int)n_knots;
    while (count > 0) {   // first knot greater than t
        int step = count / 2;
        uint32_t it = first + (uint32_t)step;
        if (!(t < knots[it])) { first = it + 1u; count -= step + 1; }
        else count = step;
    }
    const uint32_t hi = n_knots - degree - 1u;
    uint32_t i_start = first;
    if (first == n_knots || first >= hi) i_start = hi;
    if (first == 0u) i_start = degree;
    DevKey tmp[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j)
        if (j <= degree) tmp[j] = key_load(kfs + (j + i_start - degree - 1u));
#pragma unroll
    for (uint32_t lvl = 0; lvl < 3u; ++lvl) {
        if (lvl < degree) {
            const uint32_t k = lvl + 1u;
#pragma unroll
            for (uint32_t j = 0; j < 3u; ++j) {
                if (j < degree - lvl) {
                    uint32_t i = j + k + i_start - degree;
                    float alpha = (t - knots[i - 1u]) / (knots[i + degree - k] - knots[i - 1u]);
                    tmp[j] = key_interpolate(tmp[j], tmp[j + 1u], alpha);
                }
            }
        }
    }
    return tmp[0];
}

// row-major 4x4 product, terms summed left to right (matrix4.rs Mul)
TR_DEV void m4_mul(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ r) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            r[4 * i + j] = a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j] + a[4 * i + 2] * b[8 + j] + a[4 * i + 3] * b[12 + j];
}

// matrix4.rs:48-172: adjugate entry (row, col) = signed 3x3 minor that drops source row `col` and source column `row`; the
// six products of each minor are accumulated in the order +a1b2c3 -a1c2b3 -a2b1c3 +a2c1b3 +a3b1c2 -a3c1b2 (a, b, c = its columns)
TR_DEV void m4_inverse(const float* __restrict__ m, float* __restrict__ inv) {
#pragma unroll
    for (int row = 0; row < 4; ++row)
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            const int r0 = col == 0 ? 1 : 0, r1 = col <= 1 ? 2 : 1, r2 = col <= 2 ? 3 : 2;
            const int c0 = row == 0 ? 1 : 0, c1 = row <= 1 ? 2 : 1, c2 = row <= 2 ? 3 : 2;
            const float a0 = m[4 * r0 + c0], a1 = m[4 * r1 + c0], a2 = m[4 * r2 + c0];
            const float b0 = m[4 * r0 + c1], b1 = m[4 * r1 + c1], b2 = m[4 * r2 + c1];
            const float e0 = m[4 * r0 + c2], e1 = m[4 * r1 + c2], e2 = m[4 * r2 + c2];
            float v;
            if (((row + col) & 1) == 0) v = a0 * b1 * e2 - a0 * e1 * b2 - a1 * b0 * e2 + a1 * e0 * b2 + a2 * b0 * e1 - a2 * e0 * b1;
            else v = -a0 * b1 * e2 + a0 * e1 * b2 + a1 * b0 * e2 - a1 * e0 * b2 - a2 * b0 * e1 + a2 * e0 * b1;
            inv[4 * row + col] = v;
        }
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    det = 1.0f / det;
#pragma unroll
    for (int i = 0; i < 16; ++i) inv[i] *= det;
}

TR_DEV void m4_identity(float* __restrict__ m) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}

// Keyframe::transform = translate * from_mat(rotation.to_matrix()) * scale (keyframe.rs:60-63), Transform{mat, inv}
TR_DEV void key_transform(const DevKey& k, float* __restrict__ mat, float* __restrict__ inv) {
    const float x = k.q[0], y = k.q[1], z = k.q[2], w = k.q[3];
    float rot[16], rinv[16];
    m4_identity(rot);
    // quaternion.rs:67-88 writes the transpose and transposes it back; powf(v, 2.0) is v * v
    rot[0] = 1.0f - 2.0f * (y * y + z * z);  rot[4] = 2.0f * (x * y + z * w);         rot[8] = 2.0f * (x * z - y * w);
    rot[1] = 2.0f * (x * y - z * w);         rot[5] = 1.0f - 2.0f * (x * x + z * z);  rot[9] = 2.0f * (y * z + x * w);
    rot[2] = 2.0f * (x * z + y * w);         rot[6] = 2.0f * (y * z - x * w);         rot[10] = 1.0f - 2.0f * (x * x + y * y);
    m4_inverse(rot, rinv);
    float tm[16], ti[16], sm[16], si[16], a[16], b[16];
    m4_identity(tm); m4_identity(ti); m4_identity(sm); m4_identity(si);
    tm[3] = k.t.x; tm[7] = k.t.y; tm[11] = k.t.z;
    ti[3] = -k.t.x; ti[7] = -k.t.y; ti[11] = -k.t.z;
    sm[0] = k.s.x; sm[5] = k.s.y; sm[10] = k.s.z;
    si[0] = 1.0f / k.s.x; si[5] = 1.0f / k.s.y; si[10] = 1.0f / k.s.z;
    m4_mul(tm, rot, a);     // (translate * rot).mat
    m4_mul(rinv, ti, b);    // (translate * rot).inv = rot.inv * translate.inv
    m4_mul(a, sm, mat);     // (.. * scale).mat
    m4_mul(si, b, inv);     // (.. * scale).inv = scale.inv * (..).inv
}

// AnimatedTransform::transform(time) as TR_XF_WORDS floats: rows 0..2 of mat [0..11], rows 0..2 of inv [12..23], then the two elements
// [3][3]: of inv [24] and of mat [25] ([26], [27] pad the record to whole 16-byte pieces). Row 3 of a product of TRS keyframes is
// (0, 0, 0, w): the zeros are exact, but w of the INVERSE is not always 1 -- Matrix4::inverse (matrix4.rs:48-172) divides a 3x3 determinant
// by the same determinant accumulated in another order, so it comes out as 1 +- an ulp for some scalings / rotations -- and then
// Transform * Point divides by it (quirk Q5, transform.rs:211-215: it divides exactly when |w - 1| < eps). xf_point_affine_w below does.
#define TR_XF_WORDS 28
// Where such records lie one after the other -- the frame's table by shutter-time index, the wavefront schedule's per-path cache -- each takes TR_XF_REC words:
// 128 bytes, so a record is ONE cache line (and its inverse half lies in one) instead of 1.9 on average at 112 bytes; the tile kernel's cache columns have
// TR_XF_WORDS rows. (Round 6: the table's records are fetched from random places of 10 - 26 GB, every line is a trip to HBM.)
#ifndef TR_XF_REC
#define TR_XF_REC 32
#endif
TR_ANIM_EVAL void eval_xform_stack(const TrayXformLevel* __restrict__ levels, const TrayKeyframe* __restrict__ kfs,
                                              const float* __restrict__ knots, uint32_t xf_first, uint32_t xf_count, float time,
                                              float* out24) {
    float mat[16], inv[16];
    m4_identity(mat); m4_identity(inv);
    for (uint32_t l = 0; l < xf_count; ++l) {
        const TrayXformLevel* __restrict__ lv = levels + xf_first + l;
        float km[16], ki[16];
        if (lv->is_const != 0u) {   // one control point, or the open shutter misses the knot domain: evaluated on the host
#pragma unroll
            for (int i = 0; i < 16; ++i) { km[i] = lv->mat[i]; ki[i] = lv->inv[i]; }
        } else {
            const float* __restrict__ kn = knots + lv->knot_first;
            const uint32_t degree = lv->degree, nk = lv->knot_count;
            float t_val = clampf(time, kn[degree], kn[nk - 1u - degree]);   // BSpline::knot_domain
            DevKey k = spline_point(kfs + lv->kf_first, kn, nk, degree, t_val);
            key_transform(k, km, ki);
        }
        float nm[16], ni[16];
        m4_mul(km, mat, nm);   // transform = t * transform (transform.rs:191-197)
        m4_mul(inv, ki, ni);
#pragma unroll
        for (int i = 0; i < 16; ++i) { mat[i] = nm[i]; inv[i] = ni[i]; }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) { out24[i] = mat[i]; out24[12 + i] = inv[i]; }
    out24[24] = inv[15]; out24[25] = mat[15]; out24[26] = 0.0f; out24[27] = 0.0f;
}

// Transform * Point for an affine matrix given by its rows 0..2 (w == 1, so the w test of transform.rs:211-215 is a no-op)
TR_DEV f3 xf_point_affine(const float* __restrict__ m, f3 p) {
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
// ... for a matrix whose row 3 is (0, 0, 0, w): the reference's w = 0 x + 0 y + 0 z + w is that element itself, and quirk Q5 divides by it
// when it is almost -- but not exactly -- one (xf_point in dev_math.h is the general form)
TR_DEV f3 xf_point_affine_w(const float* __restrict__ m, float w, f3 p) {
    f3 r = xf_point_affine(m, p);
    if (w != 1.0f && fabsf(w - 1.0f) < kEps) r = r / w;
    return r;
}

// AnimatedColor::color (film/animated_color.rs:52-78) over keys sorted by time; n >= 2
TR_DEV f3 color_keys_at(const TrayColorKey* __restrict__ keys, uint32_t n, float time) {
    uint32_t i = 0u;
    while (i < n && keys[i].time < time) ++i;   // i = number of leading keys with key.time < time
    if (i == 0u) return mk(keys[0].color[0], keys[0].color[1], keys[0].color[2]);
    if (i == n) return mk(keys[n - 1u].color[0], keys[n - 1u].color[1], keys[n - 1u].color[2]);
    const TrayColorKey* __restrict__ a = keys + (i - 1u);
    const TrayColorKey* __restrict__ b = keys + i;
    float t = (time - a->time) / (b->time - a->time);
    return mk(lerpf(t, a->color[0], b->color[0]), lerpf(t, a->color[1], b->color[1]), lerpf(t, a->color[2], b->color[2]));
}

}  // namespace tr
