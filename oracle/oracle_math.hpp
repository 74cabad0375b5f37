// TEST INFRASTRUCTURE ONLY — CPU oracle for the tray_rust per-sample hot path.
//
// This directory is a plain C++ restatement of the reference algorithm (Twinklebear/tray_rust,
// /root/reference) used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
// checker. Nothing in the product path (tray_rust_amd/, libtrayhip.so) includes, links or calls it.
//
// PARITY UNPINNED: the reference's own tests hold no golden vector for this path (17 unit tests on
// linalg/partition only) and its RNG is OS-seeded (exec/multithreaded.rs:79), so no
// output of the reference can be reproduced. The restated reference unit tests (tests/test_oracle_kat.py)
// and analytic checks are the only pins; hot-path parity is HIP-vs-this-oracle on identical inputs.
//
// This header: f32 value types with the reference's operation order (build with -ffp-contract=off),
// Matrix4 / Transform / Quaternion / Keyframe, the counter-based RNG that replaces rand::StdRng
// (definition in DESIGN.md "TRAY-CBRNG"), and the (0,2)-sequence sampler (sampler/ld.rs).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

static constexpr float PI = 3.14159265358979323846f;        // f32::consts::PI
static constexpr float FRAC_1_PI = 0.318309886183790671538f; // f32::consts::FRAC_1_PI
static constexpr float FRAC_PI_4 = 0.785398163397448309616f; // f32::consts::FRAC_PI_4
static constexpr float EPS = std::numeric_limits<float>::epsilon();
static constexpr float INF = std::numeric_limits<float>::infinity();

// ---- linalg/vector.rs, point.rs, normal.rs: one struct, the three only differ in how they transform
struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    static Vec3 broadcast(float a) { return Vec3(a, a, a); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float length_sqr() const { return x * x + y * y + z * z; }
    float length() const { return std::sqrt(length_sqr()); }
    Vec3 normalized() const { float l = length(); return Vec3(x / l, y / l, z / l); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// linalg/mod.rs:34-127
inline float to_radians(float d) { return PI / 180.0f * d; }
inline Vec3 cross(Vec3 a, Vec3 b) { return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float lerp(float t, float a, float b) { return a * (1.0f - t) + b * t; }
inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline Vec3 spherical_dir(float sin_theta, float cos_theta, float phi) {
    return Vec3(sin_theta * std::cos(phi), sin_theta * std::sin(phi), cos_theta);
}
inline float spherical_theta(Vec3 v) { return std::acos(clampf(v.z, -1.0f, 1.0f)); }
inline float spherical_phi(Vec3 v) {
    float p = std::atan2(v.y, v.x);
    return p < 0.0f ? p + PI * 2.0f : p;
}
inline bool solve_quadratic(float a, float b, float c, float& t0, float& t1) {
    float discrim_sqr = b * b - 4.0f * a * c;
    if (discrim_sqr < 0.0f) return false;
    float discrim = std::sqrt(discrim_sqr);
    float q = b < 0.0f ? -0.5f * (b - discrim) : -0.5f * (b + discrim);
    float x = q / a, y = c / q;
    if (x > y) { t0 = y; t1 = x; } else { t0 = x; t1 = y; }
    return true;
}
inline void coordinate_system(Vec3 e1, Vec3& e2, Vec3& e3) {
    if (std::fabs(e1.x) > std::fabs(e1.y)) {
        float inv_len = 1.0f / std::sqrt(e1.x * e1.x + e1.z * e1.z);
        e2 = Vec3(-e1.z * inv_len, 0.0f, e1.x * inv_len);
    } else {
        float inv_len = 1.0f / std::sqrt(e1.y * e1.y + e1.z * e1.z);
        e2 = Vec3(0.0f, e1.z * inv_len, -e1.y * inv_len);
    }
    e3 = cross(e1, e2);
}
inline Vec3 reflect(Vec3 w, Vec3 v) { return 2.0f * dot(w, v) * v - w; }
inline bool refract(Vec3 w, Vec3 n, float eta, Vec3& out) {
    float cos_t1 = dot(n, w);
    float sin_t1_sqr = std::fmax(0.0f, 1.0f - cos_t1 * cos_t1);   // powf(x, 2.0) == x*x exactly rounded
    float sin_t2_sqr = eta * eta * sin_t1_sqr;
    if (sin_t2_sqr >= 1.0f) return false;
    float cos_t2 = std::sqrt(1.0f - sin_t2_sqr);
    out = eta * -w + (eta * cos_t1 - cos_t2) * n;
    return true;
}

// ---- film/color.rs
struct Colorf {
    float r, g, b, a;
    Colorf() : r(0), g(0), b(0), a(0) {}
    Colorf(float r_, float g_, float b_) : r(r_), g(g_), b(b_), a(1.0f) {}
    Colorf(float r_, float g_, float b_, float a_) : r(r_), g(g_), b(b_), a(a_) {}
    static Colorf broadcast(float v) { return Colorf(v, v, v, v); }
    static Colorf black() { return broadcast(0.0f); }
    Colorf clamp() const { return Colorf(clampf(r, 0, 1), clampf(g, 0, 1), clampf(b, 0, 1), clampf(a, 0, 1)); }
    float luminance() const { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }
    bool is_black() const { return r == 0.0f && g == 0.0f && b == 0.0f; }
};
inline Colorf operator+(Colorf x, Colorf y) { return Colorf(x.r + y.r, x.g + y.g, x.b + y.b, x.a + y.a); }
inline Colorf operator-(Colorf x, Colorf y) { return Colorf(x.r - y.r, x.g - y.g, x.b - y.b, x.a - y.a); }
inline Colorf operator*(Colorf x, Colorf y) { return Colorf(x.r * y.r, x.g * y.g, x.b * y.b, x.a * y.a); }
inline Colorf operator*(Colorf x, float s) { return Colorf(x.r * s, x.g * s, x.b * s, x.a * s); }
inline Colorf operator*(float s, Colorf x) { return Colorf(s * x.r, s * x.g, s * x.b, s * x.a); }
inline Colorf operator/(Colorf x, Colorf y) { return Colorf(x.r / y.r, x.g / y.g, x.b / y.b, x.a / y.a); }
inline Colorf operator/(Colorf x, float s) { return Colorf(x.r / s, x.g / s, x.b / s, x.a / s); }

// ---- linalg/matrix4.rs
struct Mat4 {
    float m[16];
    static Mat4 zero() { Mat4 r; for (float& x : r.m) x = 0.0f; return r; }
    static Mat4 identity() { Mat4 r = zero(); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
    static Mat4 from(const float* p) { Mat4 r; std::memcpy(r.m, p, sizeof r.m); return r; }
    float at(int i, int j) const { return m[4 * i + j]; }
    float& at(int i, int j) { return m[4 * i + j]; }
    Mat4 transpose() const {
        Mat4 r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.at(i, j) = at(j, i);
        return r;
    }
    // matrix4.rs:48-172: adjugate / determinant. Each adjugate entry is a signed 3x3 minor whose six
    // products are accumulated in the order the reference's (MESA-derived) expressions list them:
    // +a1 b2 c3 -a1 c2 b3 -a2 b1 c3 +a2 c1 b3 +a3 b1 c2 -a3 c1 b2 with a,b,c the minor's columns.
    static float minor3(const float a[3], const float b[3], const float c[3], bool neg) {
        if (!neg)
            return a[0] * b[1] * c[2] - a[0] * c[1] * b[2] - a[1] * b[0] * c[2] + a[1] * c[0] * b[2] + a[2] * b[0] * c[1] - a[2] * c[0] * b[1];
        return -a[0] * b[1] * c[2] + a[0] * c[1] * b[2] + a[1] * b[0] * c[2] - a[1] * c[0] * b[2] - a[2] * b[0] * c[1] + a[2] * c[0] * b[1];
    }
    Mat4 inverse() const {
        Mat4 inv;
        for (int row = 0; row < 4; ++row)
            for (int col = 0; col < 4; ++col) {
                // entry (row, col) of the inverse drops source row `col` and source column `row`
                float a[3], b[3], c[3];
                int rr[3], cc[3], n = 0, k = 0;
                for (int t = 0; t < 4; ++t) { if (t != col) rr[n++] = t; if (t != row) cc[k++] = t; }
                for (int t = 0; t < 3; ++t) { a[t] = at(rr[t], cc[0]); b[t] = at(rr[t], cc[1]); c[t] = at(rr[t], cc[2]); }
                inv.at(row, col) = minor3(a, b, c, ((row + col) & 1) != 0);
            }
        float det = m[0] * inv.m[0] + m[1] * inv.m[4] + m[2] * inv.m[8] + m[3] * inv.m[12];
        det = 1.0f / det;
        for (float& x : inv.m) x *= det;
        return inv;
    }
};
inline Mat4 operator+(const Mat4& a, const Mat4& b) { Mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
inline Mat4 operator-(const Mat4& a, const Mat4& b) { Mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
inline Mat4 operator*(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.at(i, j) = a.at(i, 0) * b.at(0, j) + a.at(i, 1) * b.at(1, j) + a.at(i, 2) * b.at(2, j) + a.at(i, 3) * b.at(3, j);
    return r;
}

struct Ray {   // linalg/ray.rs:9-22
    Vec3 o, d;
    float min_t = 0.0f, max_t = INF;
    float time = 0.0f;
    Vec3 at(float t) const { return o + d * t; }
};

// ---- linalg/transform.rs
struct Transform {
    Mat4 mat, inv;
    static Transform identity() { return Transform{Mat4::identity(), Mat4::identity()}; }
    static Transform from_mat(const Mat4& m) { return Transform{m, m.inverse()}; }
    static Transform from_pair(const Mat4& m, const Mat4& i) { return Transform{m, i}; }
    static Transform translate(Vec3 v) {
        Transform t = identity();
        t.mat.at(0, 3) = v.x; t.mat.at(1, 3) = v.y; t.mat.at(2, 3) = v.z;
        t.inv.at(0, 3) = -v.x; t.inv.at(1, 3) = -v.y; t.inv.at(2, 3) = -v.z;
        return t;
    }
    static Transform scale(Vec3 v) {
        Transform t = identity();
        t.mat.at(0, 0) = v.x; t.mat.at(1, 1) = v.y; t.mat.at(2, 2) = v.z;
        t.inv.at(0, 0) = 1.0f / v.x; t.inv.at(1, 1) = 1.0f / v.y; t.inv.at(2, 2) = 1.0f / v.z;
        return t;
    }
    static Transform rotate_x(float deg) {
        float r = to_radians(deg), s = std::sin(r), c = std::cos(r);
        Mat4 m = Mat4::identity();
        m.at(1, 1) = c; m.at(1, 2) = -s; m.at(2, 1) = s; m.at(2, 2) = c;
        return Transform{m, m.transpose()};
    }
    static Transform rotate_y(float deg) {
        float r = to_radians(deg), s = std::sin(r), c = std::cos(r);
        Mat4 m = Mat4::identity();
        m.at(0, 0) = c; m.at(0, 2) = s; m.at(2, 0) = -s; m.at(2, 2) = c;
        return Transform{m, m.transpose()};
    }
    static Transform rotate_z(float deg) {
        float r = to_radians(deg), s = std::sin(r), c = std::cos(r);
        Mat4 m = Mat4::identity();
        m.at(0, 0) = c; m.at(0, 1) = -s; m.at(1, 0) = s; m.at(1, 1) = c;
        return Transform{m, m.transpose()};
    }
    static Transform rotate(Vec3 axis, float deg) {
        Vec3 a = axis.normalized();
        float r = to_radians(deg), s = std::sin(r), c = std::cos(r);
        Mat4 m = Mat4::identity();
        m.at(0, 0) = a.x * a.x + (1.0f - a.x * a.x) * c;
        m.at(0, 1) = a.x * a.y * (1.0f - c) - a.z * s;
        m.at(0, 2) = a.x * a.z * (1.0f - c) + a.y * s;
        m.at(1, 0) = a.x * a.y * (1.0f - c) + a.z * s;
        m.at(1, 1) = a.y * a.y + (1.0f - a.y * a.y) * c;
        m.at(1, 2) = a.y * a.z * (1.0f - c) - a.x * s;
        m.at(2, 0) = a.x * a.z * (1.0f - c) - a.y * s;
        m.at(2, 1) = a.y * a.z * (1.0f - c) + a.x * s;
        m.at(2, 2) = a.z * a.z + (1.0f - a.z * a.z) * c;
        return Transform{m, m.transpose()};
    }
    Transform inverse() const { return Transform{inv, mat}; }

    static Vec3 mul_point(const Mat4& m, Vec3 p) {   // transform.rs:199-216 / 152-163 (quirk Q5)
        Vec3 res;
        for (int i = 0; i < 3; ++i) res[i] = m.at(i, 0) * p.x + m.at(i, 1) * p.y + m.at(i, 2) * p.z + m.at(i, 3);
        float w = m.at(3, 0) * p.x + m.at(3, 1) * p.y + m.at(3, 2) * p.z + m.at(3, 3);
        if (std::fabs(w - 1.0f) < EPS) return res / w;
        return res;
    }
    static Vec3 mul_vector(const Mat4& m, Vec3 v) {
        Vec3 res;
        for (int i = 0; i < 3; ++i) res[i] = m.at(i, 0) * v.x + m.at(i, 1) * v.y + m.at(i, 2) * v.z;
        return res;
    }
    static Vec3 mul_normal_t(const Mat4& m, Vec3 n) {   // uses the transpose of `m`
        Vec3 res;
        for (int i = 0; i < 3; ++i) res[i] = m.at(0, i) * n.x + m.at(1, i) * n.y + m.at(2, i) * n.z;
        return res;
    }
    Vec3 point(Vec3 p) const { return mul_point(mat, p); }
    Vec3 vector(Vec3 v) const { return mul_vector(mat, v); }
    Vec3 normal(Vec3 n) const { return mul_normal_t(inv, n); }          // transform.rs:231-243
    Vec3 inv_point(Vec3 p) const { return mul_point(inv, p); }          // :152-163
    Vec3 inv_vector(Vec3 v) const { return mul_vector(inv, v); }        // :165-172
    Ray ray(const Ray& r) const { Ray o = r; o.o = point(r.o); o.d = vector(r.d); return o; }              // :245-254
    Ray inv_ray(const Ray& r) const { Ray o = r; o.o = inv_point(r.o); o.d = inv_vector(r.d); return o; }  // :183-188
};
inline Transform operator*(const Transform& a, const Transform& b) { return Transform{a.mat * b.mat, b.inv * a.inv}; }

// ---- linalg/quaternion.rs:67-88, keyframe.rs:60-63
inline Mat4 quat_to_matrix(const float q[4]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    Mat4 r = Mat4::zero();
    r.m[0] = 1.0f - 2.0f * (y * y + z * z);  r.m[1] = 2.0f * (x * y + z * w);         r.m[2] = 2.0f * (x * z - y * w);
    r.m[4] = 2.0f * (x * y - z * w);         r.m[5] = 1.0f - 2.0f * (x * x + z * z);  r.m[6] = 2.0f * (y * z + x * w);
    r.m[8] = 2.0f * (x * z + y * w);         r.m[9] = 2.0f * (y * z - x * w);         r.m[10] = 1.0f - 2.0f * (x * x + y * y);
    r.m[15] = 1.0f;
    return r.transpose();
}
inline Transform keyframe_transform(const float t[3], const float q[4], const float s[3]) {
    Mat4 m = quat_to_matrix(q);
    return Transform::translate(Vec3(t[0], t[1], t[2])) * Transform::from_mat(m) * Transform::scale(Vec3(s[0], s[1], s[2]));
}

// ---- quaternion.rs:90-113 (dot, slerp), keyframe.rs:66-73 (Interpolate for Keyframe)
struct Key { Vec3 t; float q[4]; Vec3 s; };
inline float quat_dot(const float a[4], const float b[4]) { return (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) + a[3] * b[3]; }
inline void quat_slerp(float t, const float a[4], const float b[4], float out[4]) {
    float cos_theta = quat_dot(a, b);
    if (cos_theta > 0.9995f) {
        float q[4];
        for (int i = 0; i < 4; ++i) q[i] = (1.0f - t) * a[i] + t * b[i];
        float len = std::sqrt(quat_dot(q, q));
        for (int i = 0; i < 4; ++i) out[i] = q[i] / len;
    } else {
        float theta = std::acos(clampf(cos_theta, -1.0f, 1.0f));
        float theta_t = theta * t;
        float perp[4];
        for (int i = 0; i < 4; ++i) perp[i] = b[i] - a[i] * cos_theta;
        float len = std::sqrt(quat_dot(perp, perp));
        for (int i = 0; i < 4; ++i) perp[i] = perp[i] / len;
        float c = std::cos(theta_t), sn = std::sin(theta_t);
        for (int i = 0; i < 4; ++i) out[i] = a[i] * c + perp[i] * sn;
    }
}
inline Key key_interpolate(const Key& a, const Key& b, float t) {
    Key k;
    k.t = (1.0f - t) * a.t + t * b.t;
    quat_slerp(t, a.q, b.q, k.q);
    k.s = (1.0f - t) * a.s + t * b.s;
    return k;
}
inline Key key_from(const TrayKeyframe& f) {
    Key k;
    k.t = Vec3(f.translation[0], f.translation[1], f.translation[2]);
    for (int i = 0; i < 4; ++i) k.q[i] = f.rotation[i];
    k.s = Vec3(f.scaling[0], f.scaling[1], f.scaling[2]);
    return k;
}

// ---- bspline 0.2.2 `BSpline::point` (third-party crate, absent from /root/reference: PARITY UNPINNED). Textbook de Boor
// (Piegl & Tiller A3.1 recurrence in the d[j] form) on the span mu with knots[mu] <= t < knots[mu+1], mu clamped so that a
// t at the end of the knot domain uses the last full span. Written in the span-index form (the product library uses the
// crate's i_start = mu + 1 form): same arithmetic per blend, different bookkeeping.
template <class P, class Blend>
inline P de_boor(const P* ctrl, const float* knots, uint32_t n_knots, uint32_t degree, float t, Blend&& blend) {
    uint32_t n_ctrl = n_knots - degree - 1;
    uint32_t mu = degree;
    while (mu + 1 < n_ctrl && !(t < knots[mu + 1])) ++mu;   // last span whose start is <= t, never past the final one
    P d[8];
    for (uint32_t j = 0; j <= degree; ++j) d[j] = ctrl[mu - degree + j];
    for (uint32_t r = 1; r <= degree; ++r)
        for (uint32_t j = 0; j + r <= degree; ++j) {   // ascending j overwrites d[j] only after it was consumed as the left operand
            uint32_t lo = mu - degree + r + j;         // knot under the left end of this blend's support
            float alpha = (t - knots[lo]) / (knots[lo + degree + 1 - r] - knots[lo]);
            d[j] = blend(d[j], d[j + 1], alpha);
        }
    return d[0];
}

// ------------------------------------------------------------------ TRAY-CBRNG (replaces rand::StdRng)
// Stateless: every draw is a hash of (seed, frame, pixel, sample, dimension). See DESIGN.md.
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
inline uint32_t key_frame(uint64_t seed, uint32_t frame) {
    uint32_t h = mix32((uint32_t)seed + 0x9E3779B9u);
    h = mix32(h ^ (uint32_t)(seed >> 32));
    return mix32(h + frame);
}
inline uint32_t key_pixel(uint32_t kf, uint32_t pixel_index) { return mix32(kf + 0x9E3779B1u * (pixel_index + 1u)); }
inline uint32_t key_sample(uint32_t kp, uint32_t s) { return mix32((kp ^ 0xA511E9B3u) + 0x9E3779B1u * (s + 1u)); }
inline uint32_t key_pass(uint32_t kp, uint32_t j) { return mix32((kp ^ 0x41445054u) + 0x9E3779B1u * (j + 1u)); }   // round j of a pixel under the Adaptive sampler
inline uint32_t draw(uint32_t key, uint32_t dim) { return mix32(key + 0x9E3779B9u * (dim + 1u)); }

// pixel-level dimensions
enum { PD_SCR_X = 0, PD_SCR_Y = 1, PD_PERM_XY = 2, PD_SCR_T = 3, PD_PERM_T = 4 };
// sample-level dimensions: 2-D light / bsdf / path arrays, then the 1-D component arrays
// (order of path.rs:55-60), then Russian roulette per bounce
enum { SD_L2 = 0, SD_B2 = 3, SD_P2 = 6, SD_L1 = 9, SD_B1 = 11, SD_P1 = 13, SD_RR = 16 };

// Random-access keyed permutation of [0, l) (Kensler, "Correlated Multi-Jittered Sampling", 2013):
// stands in for rng.shuffle over the spp-long per-pixel arrays (ld.rs:58,63)
inline uint32_t permute(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893du;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8; i *= 0x0929eb3fu;
        i ^= p >> 23;
        i ^= (i & w) >> 1; i *= 1u | p >> 27;
        i *= 0x6935fa69u;
        i ^= (i & w) >> 11; i *= 0x74dcb303u;
        i ^= (i & w) >> 2; i *= 0x9e501cc3u;
        i ^= (i & w) >> 2; i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}
// Fisher-Yates over [0, n), n <= 16, same loop shape as rand 0.4's Rng::shuffle (i from n-1 down to
// 1, j uniform in [0, i]); the j draws are 16-bit fields of hashed words.
inline void shuffle_small(uint32_t key, uint32_t n, uint8_t perm[16]) {
    for (uint32_t i = 0; i < n; ++i) perm[i] = (uint8_t)i;
    for (uint32_t i = n - 1; i >= 1 && n > 0; --i) {
        uint32_t k = n - 1 - i;
        uint32_t word = draw(key, k >> 1);
        uint32_t r16 = (k & 1u) ? (word >> 16) : (word & 0xffffu);
        uint32_t j = (r16 * (i + 1u)) >> 16;
        uint8_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
    }
}

// ---- sampler/ld.rs:91-119
inline float van_der_corput(uint32_t n, uint32_t scramble) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    n ^= scramble;
    return std::fmin((float)((n >> 8) & 0xffffffu) / (float)(1u << 24), 1.0f - EPS);
}
inline float sobol(uint32_t n, uint32_t scramble) {
    uint32_t i = 1u << 31;
    while (n != 0) {
        if (n & 1u) scramble ^= i;
        n >>= 1;
        i ^= i >> 1;
    }
    return std::fmin((float)((scramble >> 8) & 0xffffffu) / (float)(1u << 24), 1.0f - EPS);
}

}  // namespace orc
