// Host-side linear algebra for the scene loader: f32 value types with the reference's operation
// order (compile with -ffp-contract=off; rustc never fuses), Matrix4 (row major, cofactor inverse in
// MESA's term order, src/linalg/matrix4.rs:48-172), Transform{mat,inv} (transform.rs:15-283),
// Quaternion (quaternion.rs), TRS Keyframe decomposition (keyframe.rs:32-63).
#pragma once
#include <cmath>
#include <cstring>
#include <limits>

namespace trayh {

static constexpr float kPi = 3.14159265358979323846f;   // f32::consts::PI
static constexpr float kEps = std::numeric_limits<float>::epsilon();

struct V3 {
    float x = 0, y = 0, z = 0;
    V3() = default;
    V3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { float l = length(a); return {a.x / l, a.y / l, a.z / l}; }
inline float to_radians(float d) { return kPi / 180.0f * d; }   // linalg/mod.rs:34-36
inline float lerpf(float t, float a, float b) { return a * (1.0f - t) + b * t; }   // linalg/mod.rs:47-49

struct BBox {   // geometry/bbox.rs
    V3 mn{INFINITY, INFINITY, INFINITY}, mx{-INFINITY, -INFINITY, -INFINITY};
    BBox() = default;
    BBox(V3 a, V3 b) : mn(a), mx(b) {}
    BBox box_union(const BBox& b) const {
        return {V3(std::fmin(mn.x, b.mn.x), std::fmin(mn.y, b.mn.y), std::fmin(mn.z, b.mn.z)),
                V3(std::fmax(mx.x, b.mx.x), std::fmax(mx.y, b.mx.y), std::fmax(mx.z, b.mx.z))};
    }
    BBox point_union(V3 p) const {
        return {V3(std::fmin(mn.x, p.x), std::fmin(mn.y, p.y), std::fmin(mn.z, p.z)),
                V3(std::fmax(mx.x, p.x), std::fmax(mx.y, p.y), std::fmax(mx.z, p.z))};
    }
    int max_extent() const {   // bbox.rs:49-58
        V3 d = mx - mn;
        if (d.x > d.y && d.x > d.z) return 0;
        if (d.y > d.z) return 1;
        return 2;
    }
    V3 center() const {   // GeomInfo::new -> bounds.lerp(0.5,0.5,0.5), bvh.rs:311-316
        return {lerpf(0.5f, mn.x, mx.x), lerpf(0.5f, mn.y, mx.y), lerpf(0.5f, mn.z, mx.z)};
    }
    float surface_area() const {   // bbox.rs:69-72
        V3 d = mx - mn;
        return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
};

struct M4 {
    float m[16];
    static M4 zero() { M4 r; std::memset(r.m, 0, sizeof r.m); return r; }
    static M4 identity() { M4 r = zero(); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
    float at(int i, int j) const { return m[4 * i + j]; }
    float& at(int i, int j) { return m[4 * i + j]; }
    M4 transpose() const {
        M4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.at(i, j) = at(j, i);
        return r;
    }
    // Cofactor inverse. Entry (i,j) of the adjugate is (-1)^(i+j) times the 3x3 minor that drops
    // row j and column i; the six triple products are summed in the order MESA's gluInvertMatrix
    // (as transcribed for row-major storage in matrix4.rs:48-160) writes them, so rounding matches.
    M4 inverse(bool* ok = nullptr) const {
        M4 inv;
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < 4; ++j) {
                int rows[3], cols[3], nr = 0, nc = 0;
                for (int k = 0; k < 4; ++k) {
                    if (k != j) rows[nr++] = k;
                    if (k != i) cols[nc++] = k;
                }
                // p,q,r = the three kept columns, restricted to the kept rows
                float p[3], q[3], r[3];
                for (int k = 0; k < 3; ++k) {
                    p[k] = at(rows[k], cols[0]);
                    q[k] = at(rows[k], cols[1]);
                    r[k] = at(rows[k], cols[2]);
                }
                float minor;
                if (((i + j) & 1) == 0)
                    minor = p[0] * q[1] * r[2] - p[0] * r[1] * q[2] - p[1] * q[0] * r[2]
                          + p[1] * r[0] * q[2] + p[2] * q[0] * r[1] - p[2] * r[0] * q[1];
                else
                    minor = -p[0] * q[1] * r[2] + p[0] * r[1] * q[2] + p[1] * q[0] * r[2]
                          - p[1] * r[0] * q[2] - p[2] * q[0] * r[1] + p[2] * r[0] * q[1];
                inv.at(i, j) = minor;
            }
        }
        float det = m[0] * inv.m[0] + m[1] * inv.m[4] + m[2] * inv.m[8] + m[3] * inv.m[12];
        if (ok) *ok = (det != 0.0f);
        det = 1.0f / det;
        for (float& x : inv.m) x *= det;
        return inv;
    }
};
inline M4 operator*(const M4& a, const M4& b) {   // matrix4.rs:225-238
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.at(i, j) = a.at(i, 0) * b.at(0, j) + a.at(i, 1) * b.at(1, j) + a.at(i, 2) * b.at(2, j) + a.at(i, 3) * b.at(3, j);
    return r;
}

struct Xform {   // Transform, transform.rs:11-14
    M4 mat = M4::identity(), inv = M4::identity();
    static Xform identity() { return {}; }
    static Xform from_mat(const M4& m, bool* ok = nullptr) { Xform t; t.mat = m; t.inv = m.inverse(ok); return t; }
    static Xform translate(V3 v) {
        Xform t;
        t.mat.at(0, 3) = v.x; t.mat.at(1, 3) = v.y; t.mat.at(2, 3) = v.z;
        t.inv.at(0, 3) = -v.x; t.inv.at(1, 3) = -v.y; t.inv.at(2, 3) = -v.z;
        return t;
    }
    static Xform scale(V3 v) {
        Xform t;
        t.mat.at(0, 0) = v.x; t.mat.at(1, 1) = v.y; t.mat.at(2, 2) = v.z;
        t.inv.at(0, 0) = 1.0f / v.x; t.inv.at(1, 1) = 1.0f / v.y; t.inv.at(2, 2) = 1.0f / v.z;
        return t;
    }
    static Xform rotate_x(float deg) {
        float r = to_radians(deg), s = std::sin(r), c = std::cos(r);
        Xform t;
        t.mat.at(1, 1) = c; t.mat.at(1, 2) = -s; t.mat.at(2, 1) = s; t.mat.at(2, 2) = c;
        t.inv = t.mat.transpose();
        return t;
    }
    static Xform rotate_y(float deg) {
        float r = to_radians(deg), s = std::sin(r), c = std::cos(r);
        Xform t;
        t.mat.at(0, 0) = c; t.mat.at(0, 2) = s; t.mat.at(2, 0) = -s; t.mat.at(2, 2) = c;
        t.inv = t.mat.transpose();
        return t;
    }
    static Xform rotate_z(float deg) {
        float r = to_radians(deg), s = std::sin(r), c = std::cos(r);
        Xform t;
        t.mat.at(0, 0) = c; t.mat.at(0, 1) = -s; t.mat.at(1, 0) = s; t.mat.at(1, 1) = c;
        t.inv = t.mat.transpose();
        return t;
    }
    static Xform rotate(V3 axis, float deg) {   // transform.rs:96-113
        V3 a = normalized(axis);
        float r = to_radians(deg), s = std::sin(r), c = std::cos(r);
        Xform t;
        M4& m = t.mat;
        m.at(0, 0) = a.x * a.x + (1.0f - a.x * a.x) * c;
        m.at(0, 1) = a.x * a.y * (1.0f - c) - a.z * s;
        m.at(0, 2) = a.x * a.z * (1.0f - c) + a.y * s;
        m.at(1, 0) = a.x * a.y * (1.0f - c) + a.z * s;
        m.at(1, 1) = a.y * a.y + (1.0f - a.y * a.y) * c;
        m.at(1, 2) = a.y * a.z * (1.0f - c) - a.x * s;
        m.at(2, 0) = a.x * a.z * (1.0f - c) - a.y * s;
        m.at(2, 1) = a.y * a.z * (1.0f - c) + a.x * s;
        m.at(2, 2) = a.z * a.z + (1.0f - a.z * a.z) * c;
        t.inv = m.transpose();
        return t;
    }
    static Xform look_at(V3 pos, V3 center, V3 up) {   // transform.rs:116-128
        V3 dir = normalized(center - pos);
        V3 left = normalized(cross(up, dir));
        V3 u = normalized(cross(dir, left));
        M4 m = M4::identity();
        for (int i = 0; i < 3; ++i) {
            m.at(i, 0) = -left[i]; m.at(i, 1) = u[i]; m.at(i, 2) = dir[i]; m.at(i, 3) = pos[i];
        }
        return from_mat(m);
    }
    Xform inverse() const { Xform t; t.mat = inv; t.inv = mat; return t; }
    // Transform * Point with the reference's inverted w test (transform.rs:199-216, quirk Q5)
    V3 point(V3 p) const {
        V3 r;
        for (int i = 0; i < 3; ++i) r[i] = mat.at(i, 0) * p.x + mat.at(i, 1) * p.y + mat.at(i, 2) * p.z + mat.at(i, 3);
        float w = mat.at(3, 0) * p.x + mat.at(3, 1) * p.y + mat.at(3, 2) * p.z + mat.at(3, 3);
        if (std::fabs(w - 1.0f) < kEps) return {r.x / w, r.y / w, r.z / w};
        return r;
    }
    BBox bbox(const BBox& b) const {   // Arvo, transform.rs:256-283
        BBox out;
        for (int i = 0; i < 3; ++i) { out.mn[i] = mat.at(i, 3); out.mx[i] = mat.at(i, 3); }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float x = mat.at(i, j) * b.mn[j], y = mat.at(i, j) * b.mx[j];
                if (x < y) { out.mn[i] += x; out.mx[i] += y; } else { out.mn[i] += y; out.mx[i] += x; }
            }
        return out;
    }
};
inline Xform operator*(const Xform& a, const Xform& b) {   // transform.rs:191-197
    Xform t;
    t.mat = a.mat * b.mat;
    t.inv = b.inv * a.inv;
    return t;
}

struct Quat {
    V3 v;
    float w = 1.0f;
    static Quat from_matrix(const M4& m) {   // Shoemake 1991, quaternion.rs:27-61
        Quat q;
        float trace = m.at(0, 0) + m.at(1, 1) + m.at(2, 2);
        if (trace > 0.0f) {
            float s = std::sqrt(trace + 1.0f);
            q.w = s / 2.0f;
            s = 0.5f / s;
            q.v = V3(s * (m.at(2, 1) - m.at(1, 2)), s * (m.at(0, 2) - m.at(2, 0)), s * (m.at(1, 0) - m.at(0, 1)));
        } else {
            const int next[3] = {1, 2, 0};
            int i = 0;
            if (m.at(1, 1) > m.at(0, 0)) i = 1;
            else if (m.at(2, 2) > m.at(0, 0)) i = 2;
            int j = next[i], k = next[j];
            float s = std::sqrt((m.at(i, i) - (m.at(j, j) + m.at(k, k))) + 1.0f);
            V3 qv;
            qv[i] = s * 0.5f;
            if (s != 0.0f) s = 0.5f / s;
            q.w = (m.at(k, j) - m.at(j, k)) * s;
            qv[j] = (m.at(j, i) + m.at(i, j)) * s;
            qv[k] = (m.at(k, i) + m.at(i, k)) * s;
            q.v = qv;
        }
        return q;
    }
    M4 to_matrix() const {   // quaternion.rs:67-88 (literal then transposed)
        // powf(x, 2.0) is an exactly rounded x*x
        float x = v.x, y = v.y, z = v.z;
        M4 r = M4::zero();
        r.m[0] = 1.0f - 2.0f * (y * y + z * z);
        r.m[1] = 2.0f * (x * y + z * w);
        r.m[2] = 2.0f * (x * z - y * w);
        r.m[4] = 2.0f * (x * y - z * w);
        r.m[5] = 1.0f - 2.0f * (x * x + z * z);
        r.m[6] = 2.0f * (y * z + x * w);
        r.m[8] = 2.0f * (x * z + y * w);
        r.m[9] = 2.0f * (y * z - x * w);
        r.m[10] = 1.0f - 2.0f * (x * x + y * y);
        r.m[15] = 1.0f;
        return r.transpose();
    }
};
inline float qdot(const Quat& a, const Quat& b) { return dot(a.v, b.v) + a.w * b.w; }

struct Keyframe {
    V3 translation;
    Quat rotation;
    V3 scaling;
    Xform transform() const {   // keyframe.rs:60-63
        M4 m = rotation.to_matrix();
        return Xform::translate(translation) * Xform::from_mat(m) * Xform::scale(scaling);
    }
};

inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }   // linalg/mod.rs:51-53

// quaternion.rs:101-113
inline Quat slerp(float t, const Quat& a, const Quat& b) {
    float cos_theta = qdot(a, b);
    Quat q;
    if (cos_theta > 0.9995f) {
        q.v = (1.0f - t) * a.v + t * b.v;
        q.w = (1.0f - t) * a.w + t * b.w;
        float len = std::sqrt(qdot(q, q));
        q.v = V3(q.v.x / len, q.v.y / len, q.v.z / len);
        q.w = q.w / len;
    } else {
        float theta = std::acos(clampf(cos_theta, -1.0f, 1.0f));
        float theta_t = theta * t;
        Quat perp;
        perp.v = b.v - a.v * cos_theta;
        perp.w = b.w - a.w * cos_theta;
        float len = std::sqrt(qdot(perp, perp));
        perp.v = V3(perp.v.x / len, perp.v.y / len, perp.v.z / len);
        perp.w = perp.w / len;
        float c = std::cos(theta_t), sn = std::sin(theta_t);
        q.v = a.v * c + perp.v * sn;
        q.w = a.w * c + perp.w * sn;
    }
    return q;
}

// bspline::Interpolate for Keyframe (keyframe.rs:66-73)
inline Keyframe kf_interpolate(const Keyframe& a, const Keyframe& b, float t) {
    Keyframe k;
    k.translation = (1.0f - t) * a.translation + t * b.translation;
    k.rotation = slerp(t, a.rotation, b.rotation);
    k.scaling = (1.0f - t) * a.scaling + t * b.scaling;
    return k;
}

// bspline 0.2.2 (crates.io; not vendored under /root/reference): BSpline::point = locate the knot span with an
// upper-bound binary search, clamp it to [degree, n_knots - degree - 1], then iterative de Boor with
// alpha = (t - knots[i-1]) / (knots[i+degree-k] - knots[i-1]). Restated from the published algorithm: parity unpinned.
inline size_t bspline_span(const float* knots, size_t n_knots, size_t degree, float t) {
    size_t first = 0;
    long count = (long)n_knots;
    while (count > 0) {   // first index with knots[i] > t
        long step = count / 2;
        size_t it = first + (size_t)step;
        if (!(t < knots[it])) { first = it + 1; count -= step + 1; }
        else count = step;
    }
    size_t hi = n_knots - degree - 1;
    if (first == n_knots) return hi;
    if (first == 0) return degree;
    if (first >= hi) return hi;
    return first;
}
template <class T, class Lerp>
inline T bspline_point(const T* pts, const float* knots, size_t n_knots, size_t degree, float t, Lerp&& interpolate) {
    size_t i_start = bspline_span(knots, n_knots, degree, t);
    T tmp[8];   // degree <= 7 (checked by the loader)
    for (size_t j = 0; j <= degree; ++j) tmp[j] = pts[j + i_start - degree - 1];
    for (size_t lvl = 0; lvl < degree; ++lvl) {
        size_t k = lvl + 1;
        for (size_t j = 0; j < degree - lvl; ++j) {
            size_t i = j + k + i_start - degree;
            float alpha = (t - knots[i - 1]) / (knots[i + degree - k] - knots[i - 1]);
            tmp[j] = interpolate(tmp[j], tmp[j + 1], alpha);
        }
    }
    return tmp[0];
}

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi, f64): a = V diag(d) V^T
inline void jacobi_eig3(double a[3][3], double v[3][3], double d[3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
        double diag = std::fabs(a[0][0]) + std::fabs(a[1][1]) + std::fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {   // A <- A J
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {   // A <- J^T A
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) d[i] = a[i][i];
}

// Keyframe::decompose (keyframe.rs:32-58): M = T * Q * P with Q = U V^T (rotation, det > 0) and
// P = V S V^T (symmetric stretch) from the SVD of the upper 3x3 in f64; only diag(P) is kept as the
// scaling. The reference uses the `la 0.2.0` crate's SVD (not in /root/reference); the polar factors
// are unique for a non-singular matrix, so any f64 SVD gives the same Q, P up to f64 rounding.
inline Keyframe decompose(const Xform& t) {
    const M4& m = t.mat;
    Keyframe k;
    k.translation = V3(m.at(0, 3), m.at(1, 3), m.at(2, 3));
    double a[3][3], ata[3][3], v[3][3], d[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i][j] = (double)m.at(i, j);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            ata[i][j] = 0.0;
            for (int r = 0; r < 3; ++r) ata[i][j] += a[r][i] * a[r][j];
        }
    jacobi_eig3(ata, v, d);
    double sig[3];
    for (int i = 0; i < 3; ++i) sig[i] = std::sqrt(d[i] > 0.0 ? d[i] : 0.0);
    // P = V S V^T ; P^-1 = V S^-1 V^T ; Q = A P^-1
    double p[3][3], pinv[3][3], q[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            p[i][j] = 0.0; pinv[i][j] = 0.0;
            for (int r = 0; r < 3; ++r) {
                p[i][j] += v[i][r] * sig[r] * v[j][r];
                pinv[i][j] += v[i][r] * (sig[r] > 0.0 ? 1.0 / sig[r] : 0.0) * v[j][r];
            }
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            q[i][j] = 0.0;
            for (int r = 0; r < 3; ++r) q[i][j] += a[i][r] * pinv[r][j];
        }
    double det = q[0][0] * (q[1][1] * q[2][2] - q[1][2] * q[2][1]) - q[0][1] * (q[1][0] * q[2][2] - q[1][2] * q[2][0])
               + q[0][2] * (q[1][0] * q[2][1] - q[1][1] * q[2][0]);
    if (det < 0.0)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { q[i][j] = -q[i][j]; p[i][j] = -p[i][j]; }
    M4 qm = M4::identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) qm.at(i, j) = (float)q[i][j];
    k.rotation = Quat::from_matrix(qm);
    k.scaling = V3((float)p[0][0], (float)p[1][1], (float)p[2][2]);
    return k;
}

}  // namespace trayh
