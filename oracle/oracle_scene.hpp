// TEST INFRASTRUCTURE ONLY — see oracle_math.hpp. Geometry, lights and BxDFs of the CPU oracle:
// restatement of src/geometry/*, src/light/mod.rs, src/bxdf/*, src/material/*, src/mc.rs over the
// TrayFlatScene POD (include/trayhip.h). Parity unpinned (no reference golden vectors exist).
#pragma once
#include <cstdint>
#include <vector>

#include "../include/trayhip.h"
#include "oracle_math.hpp"

namespace orc {

enum { ORC_FAITHFUL_XF = 1, ORC_BRUTE_FORCE = 2,
       ORC_FRESH_SHUFFLES = 4 };   // every per-path LD array shuffled by its OWN Fisher-Yates (ld.rs:58,63 as written) instead of a pool permutation:
                                   // a different sampler definition (not the product's), for the statistical comparison in tests/test_sampler_pool.py

struct Stats { uint64_t samples = 0, vertices = 0, rays = 0; };

// geometry/differential_geometry.rs:9-27 (+ instance / material of intersection.rs:11-18)
struct Hit {
    Vec3 p, n, ng;
    float u = 0, v = 0, time = 0;
    Vec3 dp_du, dp_dv;
    uint32_t inst = 0xffffffffu;
    uint32_t prim = 0;
};

// DifferentialGeometry::new (differential_geometry.rs:32-47)
inline void dg_new(Hit& h, Vec3 p, Vec3 ng, float u, float v, float time, Vec3 dp_du, Vec3 dp_dv) {
    h.p = p; h.n = cross(dp_du, dp_dv).normalized(); h.ng = ng.normalized();
    h.u = u; h.v = v; h.time = time; h.dp_du = dp_du; h.dp_dv = dp_dv;
}
// DifferentialGeometry::with_normal (:49-64): ng = shading normal
inline void dg_with_normal(Hit& h, Vec3 p, Vec3 n, float u, float v, float time, Vec3 dp_du, Vec3 dp_dv) {
    Vec3 nn = n.normalized();
    h.p = p; h.n = nn; h.ng = nn; h.u = u; h.v = v; h.time = time; h.dp_du = dp_du; h.dp_dv = dp_dv;
}

// ------------------------------------------------------------------ mc.rs
inline void concentric_sample_disk(float u0, float u1, float& dx, float& dy) {
    float sx = 2.0f * u0 - 1.0f, sy = 2.0f * u1 - 1.0f;
    float radius, theta;
    if (sx == 0.0f && sy == 0.0f) { dx = sx; dy = sy; return; }
    if (sx >= -sy) {
        if (sx > sy) {
            radius = sx;
            theta = sy > 0.0f ? sy / sx : 8.0f + sy / sx;
        } else {
            radius = sy;
            theta = 2.0f - sx / sy;
        }
    } else if (sx <= sy) {
        radius = -sx;
        theta = 4.0f + sy / sx;
    } else {
        radius = -sy;
        theta = 6.0f - sx / sy;
    }
    theta = theta * FRAC_PI_4;
    dx = radius * std::cos(theta);
    dy = radius * std::sin(theta);
}
inline Vec3 cos_sample_hemisphere(float u0, float u1) {
    float dx, dy;
    concentric_sample_disk(u0, u1, dx, dy);
    return Vec3(dx, dy, std::sqrt(std::fmax(0.0f, 1.0f - dx * dx - dy * dy)));
}
inline float power_heuristic(float n_f, float pdf_f, float n_g, float pdf_g) {
    float f = n_f * pdf_f, g = n_g * pdf_g;
    return (f * f) / (f * f + g * g);
}
inline float uniform_cone_pdf(float cos_theta) { return 1.0f / (PI * 2.0f * (1.0f - cos_theta)); }
inline Vec3 uniform_sample_cone_frame(float u0, float u1, float cos_theta_max, Vec3 wx, Vec3 wy, Vec3 wz) {
    float cos_theta = lerp(u0, cos_theta_max, 1.0f);
    float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
    float phi = u1 * PI * 2.0f;
    return std::cos(phi) * sin_theta * wx + std::sin(phi) * sin_theta * wy + cos_theta * wz;
}
inline Vec3 uniform_sample_sphere(float u0, float u1) {
    float z = 1.0f - 2.0f * u0;
    float r = std::sqrt(std::fmax(0.0f, 1.0f - z * z));
    float phi = PI * 2.0f * u1;
    return Vec3(std::cos(phi) * r, std::sin(phi) * r, z);
}

// ------------------------------------------------------------------ primitives (object space)
// Sphere::intersect (sphere.rs:33-81)
inline bool sphere_intersect(float radius, Ray& ray, Hit& h) {
    float a = ray.d.length_sqr();
    float b = 2.0f * dot(ray.d, ray.o);
    float c = dot(ray.o, ray.o) - radius * radius;
    float t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return false;
    if (t0 > ray.max_t || t1 < ray.min_t) return false;
    float t_hit = t0;
    if (t_hit < ray.min_t) {
        t_hit = t1;
        if (t_hit > ray.max_t) return false;
    }
    ray.max_t = t_hit;
    Vec3 p = ray.at(t_hit);
    Vec3 n = p;
    float theta = std::acos(clampf(p.z / radius, -1.0f, 1.0f));
    float inv_z = 1.0f / std::sqrt(p.x * p.x + p.y * p.y);
    float cos_phi = p.x * inv_z, sin_phi = p.y * inv_z;
    float u = std::atan2(p.x, p.y) / (2.0f * PI);   // argument order as written (sphere.rs:71)
    if (u < 0.0f) u = u + 1.0f;
    float v = theta / PI;
    Vec3 dp_du(-PI * 2.0f * p.y, PI * 2.0f * p.x, 0.0f);
    Vec3 dp_dv = Vec3(p.z * cos_phi, p.z * sin_phi, -radius * std::sin(theta)) * PI;
    dg_with_normal(h, p, n, u, v, ray.time, dp_du, dp_dv);
    return true;
}
// Rectangle::intersect (rectangle.rs:38-64)
inline bool rect_intersect(float width, float height, Ray& ray, Hit& h) {
    if (std::fabs(ray.d.z) < 1e-8f) return false;
    float t = -ray.o.z / ray.d.z;
    if (t < ray.min_t || t > ray.max_t) return false;
    Vec3 p = ray.at(t);
    float hw = width / 2.0f, hh = height / 2.0f;
    if (p.x >= -hw && p.x <= hw && p.y >= -hh && p.y <= hh) {
        ray.max_t = t;
        float u = (p.x + hw) / (2.0f * hw), v = (p.y + hh) / (2.0f * hh);
        dg_new(h, p, Vec3(0, 0, 1), u, v, ray.time, Vec3(hw * 2.0f, 0, 0), Vec3(0, hh * 2.0f, 0));
        return true;
    }
    return false;
}
// Disk::intersect (disk.rs:42-76)
inline bool disk_intersect(float radius, float inner_radius, Ray& ray, Hit& h) {
    if (std::fabs(ray.d.z) == 0.0f) return false;
    float t = -ray.o.z / ray.d.z;
    if (t < ray.min_t || t > ray.max_t) return false;
    Vec3 p = ray.at(t);
    float dist_sqr = p.x * p.x + p.y * p.y;
    if (dist_sqr > radius * radius || dist_sqr < inner_radius * inner_radius) return false;
    float phi = std::atan2(p.y, p.x);
    if (phi < 0.0f) phi += PI * 2.0f;
    if (phi > PI * 2.0f) return false;
    ray.max_t = t;
    float hit_radius = std::sqrt(dist_sqr);
    float u = phi / (2.0f * PI);
    float v = 1.0f - (hit_radius - inner_radius) / (radius - inner_radius);
    Vec3 dp_du(-PI * 2.0f * p.y, PI * 2.0f * p.x, 0.0f);
    Vec3 dp_dv = ((inner_radius - radius) / hit_radius) * Vec3(p.x, p.y, 0.0f);
    dg_new(h, p, Vec3(0, 0, 1), u, v, ray.time, dp_du, dp_dv);
    return true;
}
// intersect_triangle (mesh.rs:136-198)
inline bool triangle_intersect(const TrayTriVerts& tv, const TrayTriAttrs& ta, Ray& ray, Hit& h) {
    Vec3 pa(tv.pa[0], tv.pa[1], tv.pa[2]), pb(tv.pb[0], tv.pb[1], tv.pb[2]), pc(tv.pc[0], tv.pc[1], tv.pc[2]);
    Vec3 e0 = pb - pa, e1 = pc - pa;
    Vec3 s0 = cross(ray.d, e1);
    float dv = dot(s0, e0);
    if (dv == 0.0f) return false;
    float div = 1.0f / dv;
    Vec3 d = ray.o - pa;
    float b1 = dot(d, s0) * div;
    if (b1 < 0.0f || b1 > 1.0f) return false;
    Vec3 s1 = cross(d, e0);
    float b2 = dot(ray.d, s1) * div;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return false;
    float t = dot(e1, s1) * div;
    if (t < ray.min_t || t > ray.max_t) return false;
    float b0 = 1.0f - b1 - b2;
    ray.max_t = t;
    Vec3 p = ray.at(t);
    Vec3 na(ta.na[0], ta.na[1], ta.na[2]), nb(ta.nb[0], ta.nb[1], ta.nb[2]), nc(ta.nc[0], ta.nc[1], ta.nc[2]);
    Vec3 n = (b0 * na + b1 * nb + b2 * nc).normalized();
    // texcoords are Points with z = 0 (mesh.rs:67-68)
    Vec3 tA(ta.ta[0], ta.ta[1], 0.0f), tB(ta.tb[0], ta.tb[1], 0.0f), tC(ta.tc[0], ta.tc[1], 0.0f);
    Vec3 texcoord = b0 * tA + b1 * tB + b2 * tC;
    float du[2] = {tA.x - tC.x, tB.x - tC.x};
    float dvv[2] = {tA.y - tC.y, tB.y - tC.y};
    float det = du[0] * dvv[1] - dvv[0] * du[1];
    Vec3 dp_du, dp_dv;
    if (det == 0.0f) {
        coordinate_system(cross(e1, e0).normalized(), dp_du, dp_dv);
    } else {
        det = 1.0f / det;
        Vec3 dp0 = pa - pc, dp1 = pb - pc;
        dp_du = (dvv[1] * dp0 - dvv[0] * dp1) * det;
        dp_dv = (-du[1] * dp0 + du[0] * dp1) * det;
    }
    dg_with_normal(h, p, n, texcoord.x, texcoord.y, ray.time, dp_du, dp_dv);
    return true;
}

// BBox::fast_intersect (bbox.rs:75-104)
inline bool bbox_fast_intersect(const TrayBvhNode& nd, const Ray& r, Vec3 inv_dir, const int neg_dir[3]) {
    const float* b[2] = {nd.bmin, nd.bmax};
    float tmin = (b[neg_dir[0]][0] - r.o.x) * inv_dir.x;
    float tmax = (b[1 - neg_dir[0]][0] - r.o.x) * inv_dir.x;
    float tymin = (b[neg_dir[1]][1] - r.o.y) * inv_dir.y;
    float tymax = (b[1 - neg_dir[1]][1] - r.o.y) * inv_dir.y;
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (b[neg_dir[2]][2] - r.o.z) * inv_dir.z;
    float tzmax = (b[1 - neg_dir[2]][2] - r.o.z) * inv_dir.z;
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return tmin < r.max_t && tmax > r.min_t;
}

// BVH::intersect (bvh.rs:81-130): `leaf(first, count)` tests the primitives of a leaf in order and
// returns through `f`; every success shrinks ray.max_t so the last success is the closest.
template <class LeafFn>
inline void bvh_traverse(const TrayBvhNode* tree, Ray& ray, LeafFn&& leaf) {
    Vec3 inv_dir(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int neg_dir[3] = {ray.d.x < 0.0f, ray.d.y < 0.0f, ray.d.z < 0.0f};
    uint32_t stack[64];
    int sp = 0;
    uint32_t current = 0;
    for (;;) {
        const TrayBvhNode& node = tree[current];
        if (bbox_fast_intersect(node, ray, inv_dir, neg_dir)) {
            if (node.count > 0) {
                leaf(node.offset, (uint32_t)node.count);
                if (sp == 0) break;
                current = stack[--sp];
            } else {
                if (neg_dir[node.axis]) {
                    stack[sp++] = current + 1;
                    current = node.offset;
                } else {
                    stack[sp++] = node.offset;
                    current = current + 1;
                }
            }
        } else {
            if (sp == 0) break;
            current = stack[--sp];
        }
    }
}

struct SceneView {
    const TrayFlatScene* fs;
    int flags;
    Stats* stats;
    // the Sampler the integrator draws from (TRAY_SAMPLER_*; Adaptive: its samples_taken at the moment) -- see PathSamples in oracle.cpp
    uint32_t smp_kind = TRAY_SAMPLER_LOW_DISCREPANCY, smp_offset = 0;
    // AnimatedTransform::transform (animated_transform.rs:40-56) from the TRS keyframes of the spline stack
    Transform stack_transform(uint32_t xf_first, uint32_t xf_count, float time) const {
        Transform t = Transform::identity();
        for (uint32_t l = 0; l < xf_count; ++l) {
            const TrayXformLevel& lv = fs->xf_levels[xf_first + l];
            if (lv.kf_count == 1) {
                const TrayKeyframe& k = fs->keyframes[lv.kf_first];
                t = keyframe_transform(k.translation, k.rotation, k.scaling) * t;
                continue;
            }
            const float* knots = fs->knots + lv.knot_first;
            float lo = knots[lv.degree], hi = knots[lv.knot_count - 1 - lv.degree];   // BSpline::knot_domain
            float t_val = clampf(time, lo, hi);
            Key ctrl[64];
            uint32_t n = lv.kf_count < 64 ? lv.kf_count : 64;
            for (uint32_t i = 0; i < n; ++i) ctrl[i] = key_from(fs->keyframes[lv.kf_first + i]);
            Key k = de_boor(ctrl, knots, lv.knot_count, lv.degree, t_val, key_interpolate);
            float tr[3] = {k.t.x, k.t.y, k.t.z}, sc[3] = {k.s.x, k.s.y, k.s.z};
            t = keyframe_transform(tr, k.q, sc) * t;
        }
        return t;
    }
    // per-instance transforms; in faithful mode rebuilt on every use like receiver.rs:30. An instance whose stack has a
    // moving level is always rebuilt at the ray's time.
    Transform instance_transform(uint32_t i, float time) const {
        const TrayInstance& in = fs->instances[i];
        if ((flags & ORC_FAITHFUL_XF) || in.animated) return stack_transform(in.xf_first, in.xf_count, time);
        return Transform::from_pair(Mat4::from(in.mat), Mat4::from(in.inv));
    }
};


// Mesh::intersect (mesh.rs:82-84)
inline bool mesh_intersect(const SceneView& sv, const TrayMesh& m, Ray& ray, Hit& h) {
    const TrayFlatScene& fs = *sv.fs;
    bool any = false;
    auto leaf = [&](uint32_t first, uint32_t count) {
        for (uint32_t k = 0; k < count; ++k) {
            uint32_t slot = m.tri_offset + first + k;
            Hit cand;
            if (triangle_intersect(fs.tri_verts[slot], fs.tri_attrs[slot], ray, cand)) { h = cand; h.prim = slot; any = true; }
        }
    };
    if (sv.flags & ORC_BRUTE_FORCE) leaf(0, m.tri_count);
    else bvh_traverse(fs.mesh_nodes + m.node_offset, ray, leaf);
    return any;
}

// AnimatedMesh::intersect (animated_mesh.rs:130-134): BVH<AnimatedTriangle>::intersect over the ONE tree AnimatedMesh::new built for
// times[0] .. times[1] (:124-126; Boundable::update_deformation has no caller -- quirk Q13), every triangle at the ray's time:
// AnimatedMeshData::active_keyframes (:56-70), position / normal / texcoord = lerp of the two keyframes' values (:72-107, linalg/mod.rs:47-49),
// then mesh.rs's intersect_triangle (AnimatedTriangle::intersect, :160-172).
inline bool animated_mesh_intersect(const SceneView& sv, uint32_t mesh_id, Ray& ray, Hit& h) {
    const TrayFlatScene& fs = *sv.fs;
    const TrayMesh& m = fs.meshes[mesh_id];
    const TrayMeshKeys& mk = fs.mesh_keys[mesh_id];
    const float* times = fs.key_times + mk.time_first;
    // times.binary_search_by(|t| t.partial_cmp(&time).unwrap()) over ascending times
    uint32_t i = 0;
    while (i < mk.n_keys && times[i] < ray.time) ++i;
    uint32_t lo, hi;
    bool two = false;
    if (i < mk.n_keys && times[i] == ray.time) { lo = hi = i; }            // Ok(i) => (i, None)
    else if (i == mk.n_keys) { lo = hi = i - 1; }                           // Err(len) => (len - 1, None)
    else if (i == 0) { lo = hi = 0; }                                       // Err(0) => (0, None)
    else { lo = i - 1; hi = i; two = true; }                                // Err(i) => (i - 1, Some(i))
    const float x = two ? (ray.time - times[lo]) / (times[hi] - times[lo]) : 0.0f;
    auto lerp3 = [&](const float* a, const float* b, float* out) { for (int c = 0; c < 3; ++c) out[c] = a[c] * (1.0f - x) + b[c] * x; };
    auto lerp2 = [&](const float* a, const float* b, float* out) { for (int c = 0; c < 2; ++c) out[c] = a[c] * (1.0f - x) + b[c] * x; };
    bool any = false;
    auto leaf = [&](uint32_t first, uint32_t count) {
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t slot = m.tri_offset + first + k;   // the triangle's record in keyframe 0
            const TrayTriVerts& v0 = fs.tri_verts[slot + (size_t)lo * m.tri_count];
            const TrayTriAttrs& a0 = fs.tri_attrs[slot + (size_t)lo * m.tri_count];
            TrayTriVerts v = v0;
            TrayTriAttrs a = a0;
            if (two) {
                const TrayTriVerts& v1 = fs.tri_verts[slot + (size_t)hi * m.tri_count];
                const TrayTriAttrs& a1 = fs.tri_attrs[slot + (size_t)hi * m.tri_count];
                lerp3(v0.pa, v1.pa, v.pa); lerp3(v0.pb, v1.pb, v.pb); lerp3(v0.pc, v1.pc, v.pc);
                lerp3(a0.na, a1.na, a.na); lerp3(a0.nb, a1.nb, a.nb); lerp3(a0.nc, a1.nc, a.nc);
                lerp2(a0.ta, a1.ta, a.ta); lerp2(a0.tb, a1.tb, a.tb); lerp2(a0.tc, a1.tc, a.tc);
            }
            Hit cand;
            if (triangle_intersect(v, a, ray, cand)) { h = cand; h.prim = slot; any = true; }
        }
    };
    if (sv.flags & ORC_BRUTE_FORCE) leaf(0, m.tri_count);
    else bvh_traverse(fs.mesh_nodes + m.node_offset, ray, leaf);
    return any;
}

inline bool geom_intersect(const SceneView& sv, const TrayInstance& in, Ray& local, Hit& h) {
    switch (in.geom_type) {
        case TRAY_GEOM_SPHERE: return sphere_intersect(in.geom_params[0], local, h);
        case TRAY_GEOM_DISK: return disk_intersect(in.geom_params[0], in.geom_params[1], local, h);
        case TRAY_GEOM_RECT: return rect_intersect(in.geom_params[0], in.geom_params[1], local, h);
        case TRAY_GEOM_MESH: return mesh_intersect(sv, sv.fs->meshes[in.mesh_id], local, h);
        case TRAY_GEOM_ANIMATED_MESH: return animated_mesh_intersect(sv, in.mesh_id, local, h);
        default: return false;
    }
}

// Instance::intersect -> Receiver::intersect / Emitter::intersect (receiver.rs:29-44, emitter.rs:118-137)
inline bool instance_intersect(const SceneView& sv, uint32_t i, Ray& ray, Hit& h) {
    const TrayInstance& in = sv.fs->instances[i];
    if (in.kind == TRAY_INST_POINT_EMITTER) return false;
    Transform t = sv.instance_transform(i, ray.time);
    Ray local = t.inv_ray(ray);
    Hit dg;
    if (!geom_intersect(sv, in, local, dg)) return false;
    ray.max_t = local.max_t;
    dg.p = t.point(dg.p);
    dg.n = t.normal(dg.n);
    dg.ng = t.normal(dg.ng);
    dg.dp_du = t.vector(dg.dp_du);
    dg.dp_dv = t.vector(dg.dp_dv);
    dg.inst = i;
    h = dg;
    return true;
}

// Scene::intersect (scene.rs:148-150)
inline bool scene_intersect(const SceneView& sv, Ray& ray, Hit& h) {
    if (sv.stats) sv.stats->rays++;
    const TrayFlatScene& fs = *sv.fs;
    bool any = false;
    auto leaf = [&](uint32_t first, uint32_t count) {
        for (uint32_t k = 0; k < count; ++k) {
            uint32_t i = fs.top_order[first + k];
            Hit cand;
            if (instance_intersect(sv, i, ray, cand)) { h = cand; any = true; }
        }
    };
    if (sv.flags & ORC_BRUTE_FORCE) {
        for (uint32_t i = 0; i < fs.n_instances; ++i) {
            Hit cand;
            if (instance_intersect(sv, i, ray, cand)) { h = cand; any = true; }
        }
    } else {
        bvh_traverse(fs.top_nodes, ray, leaf);
    }
    return any;
}

// ------------------------------------------------------------------ Sampleable (rectangle.rs:74-104, disk.rs:84-110, sphere.rs:91-140)
inline void geom_sample(const TrayInstance& in, Vec3 p, float u0, float u1, Vec3& ps, Vec3& ns) {
    switch (in.geom_type) {
        case TRAY_GEOM_RECT: {
            float w = in.geom_params[0], hgt = in.geom_params[1];
            ps = Vec3(u0 * w - w / 2.0f, u1 * hgt - hgt / 2.0f, 0.0f);
            ns = Vec3(0, 0, 1);
            return;
        }
        case TRAY_GEOM_DISK: {
            float dx, dy;
            concentric_sample_disk(u0, u1, dx, dy);
            ps = Vec3(dx * in.geom_params[0], dy * in.geom_params[0], 0.0f);
            ns = Vec3(0, 0, 1);
            return;
        }
        default: {   // sphere
            float radius = in.geom_params[0];
            float dist_sqr = (p - Vec3(0, 0, 0)).length_sqr();
            if (dist_sqr - radius * radius < 0.0001f) {
                ps = Vec3(0, 0, 0) + radius * uniform_sample_sphere(u0, u1);
                ns = ps.normalized();
                return;
            }
            Vec3 w_z = (Vec3(0, 0, 0) - p).normalized();
            Vec3 w_x, w_y;
            coordinate_system(w_z, w_x, w_y);
            float cos_theta_max = std::sqrt(std::fmax(0.0f, 1.0f - radius * radius / dist_sqr));
            Ray ray;
            ray.o = p;
            ray.d = uniform_sample_cone_frame(u0, u1, cos_theta_max, w_x, w_y, w_z).normalized();
            ray.time = 0.0f;
            Hit dg;
            if (sphere_intersect(radius, ray, dg)) { ps = dg.p; ns = dg.ng; return; }
            float t = dot(Vec3(0, 0, 0) - p, ray.d);
            ps = ray.at(t);
            ns = ps.normalized();
            return;
        }
    }
}
inline float geom_pdf(const TrayInstance& in, Vec3 p, Vec3 w_i) {
    if (in.geom_type == TRAY_GEOM_SPHERE) {
        float radius = in.geom_params[0];
        float dist_sqr = (p - Vec3(0, 0, 0)).length_sqr();
        if (dist_sqr - radius * radius < 0.0001f) return 1.0f / (4.0f * PI * radius);   // surface_area = 4*pi*r (quirk Q7)
        float cos_theta_max = std::sqrt(std::fmax(0.0f, 1.0f - radius * radius / dist_sqr));
        return uniform_cone_pdf(cos_theta_max);
    }
    Ray ray;
    ray.o = p; ray.d = w_i; ray.min_t = 0.001f; ray.max_t = INF; ray.time = 0.0f;
    Hit d;
    bool hit;
    float area;
    if (in.geom_type == TRAY_GEOM_RECT) {
        hit = rect_intersect(in.geom_params[0], in.geom_params[1], ray, d);
        area = in.geom_params[0] * in.geom_params[1];
    } else {
        hit = disk_intersect(in.geom_params[0], in.geom_params[1], ray, d);
        area = PI * (in.geom_params[0] * in.geom_params[0] - in.geom_params[1] * in.geom_params[1]);
    }
    if (!hit) return 0.0f;
    Vec3 w = -w_i;
    float pdf = (p - ray.at(ray.max_t)).length_sqr() / (std::fabs(dot(d.n, w)) * area);
    return std::isfinite(pdf) ? pdf : 0.0f;
}

// ------------------------------------------------------------------ Light for Emitter (emitter.rs:140-203)
// AnimatedColor::color (film/animated_color.rs:52-78)
inline Colorf inst_emission(const SceneView& sv, const TrayInstance& in, float time) {
    if (in.emis_count < 2) return Colorf(in.emission[0], in.emission[1], in.emission[2], in.emission[3]);
    const TrayColorKey* keys = sv.fs->color_keys + in.emis_first;
    const TrayColorKey* first = nullptr;
    const TrayColorKey* second = nullptr;
    uint32_t i = 0;
    while (i < in.emis_count && keys[i].time < time) { first = &keys[i]; ++i; }   // take_while(..).last()
    if (i < in.emis_count) second = &keys[i];                                       // skip_while(..).next()
    auto col = [](const TrayColorKey* k) { return Colorf(k->color[0], k->color[1], k->color[2], k->color[3]); };
    if (!first) return col(&keys[0]);
    if (!second) return col(&keys[in.emis_count - 1]);
    float t = (time - first->time) / (second->time - first->time);
    return Colorf(lerp(t, first->color[0], second->color[0]), lerp(t, first->color[1], second->color[1]),
                  lerp(t, first->color[2], second->color[2]), lerp(t, first->color[3], second->color[3]));
}
inline Colorf emitter_radiance(const SceneView& sv, const TrayInstance& in, Vec3 w, Vec3 n, float time) {   // emitter.rs:139-141
    return dot(w, n) > 0.0f ? inst_emission(sv, in, time) : Colorf::black();
}
struct LightSample { Colorf li; Vec3 w_i; float pdf; Ray occlusion; };
inline LightSample light_sample_incident(const SceneView& sv, uint32_t inst, Vec3 p, float u0, float u1, float time) {
    const TrayInstance& in = sv.fs->instances[inst];
    LightSample ls;
    Transform t = sv.instance_transform(inst, time);
    auto test_points = [&](Vec3 a, Vec3 b) {   // OcclusionTester::test_points (light/mod.rs:21-23)
        Ray r; r.o = a; r.d = b - a; r.min_t = 0.001f; r.max_t = 0.999f; r.time = time; return r;
    };
    if (in.kind == TRAY_INST_POINT_EMITTER) {
        Vec3 pos = t.point(Vec3(0, 0, 0));
        ls.w_i = (pos - p).normalized();
        ls.li = inst_emission(sv, in, time) / (pos - p).length_sqr();
        ls.pdf = 1.0f;
        ls.occlusion = test_points(p, pos);
        return ls;
    }
    Vec3 p_l = t.inv_point(p);
    Vec3 p_sampled, normal;
    geom_sample(in, p_l, u0, u1, p_sampled, normal);
    Vec3 w_il = (p_sampled - p_l).normalized();
    ls.pdf = geom_pdf(in, p_l, w_il);
    ls.li = emitter_radiance(sv, in, -w_il, normal, time);
    Vec3 p_w = t.point(p_sampled);
    ls.w_i = t.vector(w_il);
    ls.occlusion = test_points(p, p_w);
    return ls;
}
inline float light_pdf(const SceneView& sv, uint32_t inst, Vec3 p, Vec3 w_i, float time) {
    const TrayInstance& in = sv.fs->instances[inst];
    if (in.kind == TRAY_INST_POINT_EMITTER) return 0.0f;
    Transform t = sv.instance_transform(inst, time);
    Vec3 p_l = t.inv_point(p);
    Vec3 w = t.inv_vector(w_i).normalized();
    return geom_pdf(in, p_l, w);
}

// ------------------------------------------------------------------ BxDFs (src/bxdf/*)
enum { BX_REFLECTION = 1, BX_TRANSMISSION = 2, BX_DIFFUSE = 4, BX_GLOSSY = 8, BX_SPECULAR = 16 };
enum { BX_ALL = 31, BX_NON_SPECULAR = 15 };   // bxdf/mod.rs:37-88

inline float cos_theta(Vec3 v) { return v.z; }
inline float cos_theta_sqr(Vec3 v) { return v.z * v.z; }
inline float sin_theta_sqr(Vec3 v) { return std::fmax(0.0f, 1.0f - v.z * v.z); }
inline float sin_theta(Vec3 v) { return std::sqrt(sin_theta_sqr(v)); }
inline float tan_theta(Vec3 v) {
    float s2 = sin_theta_sqr(v);
    return s2 <= 0.0f ? 0.0f : std::sqrt(s2) / cos_theta(v);
}
inline float tan_theta_sqr(Vec3 v) { return sin_theta_sqr(v) / cos_theta_sqr(v); }
inline float cos_phi(Vec3 v) { float s = sin_theta(v); return s == 0.0f ? 1.0f : clampf(v.x / s, -1.0f, 1.0f); }
inline float sin_phi(Vec3 v) { float s = sin_theta(v); return s == 0.0f ? 0.0f : clampf(v.y / s, -1.0f, 1.0f); }
inline bool same_hemisphere(Vec3 a, Vec3 b) { return a.z * b.z > 0.0f; }

// fresnel.rs
enum FresnelKind { FR_DIELECTRIC, FR_CONDUCTOR };
struct Fresnel {
    FresnelKind kind;
    float eta_i, eta_t;   // dielectric
    Colorf eta, k;        // conductor
    Colorf eval(float cos_i) const {
        if (kind == FR_CONDUCTOR) {
            float ci = std::fabs(cos_i);
            Colorf a = (eta * eta + k * k) * ci * ci;
            Colorf col = Colorf::broadcast(1.0f);
            Colorf r_par = (a - eta * ci * 2.0f + col) / (a + eta * ci * 2.0f + col);
            Colorf b = eta * eta + k * k;
            col = Colorf::broadcast(ci * ci);
            Colorf r_perp = (b - eta * ci * 2.0f + col) / (b + eta * ci * 2.0f + col);
            return (r_par + r_perp) * 0.5f;
        }
        float ci = clampf(cos_i, -1.0f, 1.0f);
        float ei = ci > 0.0f ? eta_i : eta_t, et = ci > 0.0f ? eta_t : eta_i;
        float sin_t = ei / et * std::sqrt(std::fmax(0.0f, 1.0f - ci * ci));
        if (sin_t >= 1.0f) return Colorf::broadcast(1.0f);
        float ct = std::sqrt(std::fmax(0.0f, 1.0f - sin_t * sin_t));
        float c = std::fabs(ci);
        float r_par = (et * c - ei * ct) / (et * c + ei * ct);
        float r_perp = (ei * c - et * ct) / (ei * c + et * ct);
        return Colorf::broadcast(0.5f * (r_par * r_par + r_perp * r_perp));
    }
};

// microfacet/beckmann.rs, microfacet/ggx.rs: the two MicrofacetDistribution impls behind one value type (the reference holds a
// `&MicrofacetDistribution`; its materials only ever construct Beckmann -- GGX is reachable here through the loader's
// `"microfacet": "ggx"` extension key)
struct Beckmann {
    float width;
    bool ggx = false;
    static Beckmann make(float w, bool ggx_ = false) { Beckmann m{std::fmax(w, 0.000001f)}; m.ggx = ggx_; return m; }
    float normal_distribution(Vec3 w_h) const {
        if (ggx) {   // ggx.rs:27-36; powf(x, 2.0) is x * x exactly (and what LLVM makes of it), powf(x, 4.0) is libm's powf: ONE rounding of x^4
            if (cos_theta(w_h) > 0.0f) {
                float width_sqr = width * width;
                float c = cos_theta(w_h);
                float t = tan_theta(w_h);
                float s = width_sqr + t * t;
                float denom = PI * std::pow(c, 4.0f) * (s * s);
                return width_sqr / denom;
            }
            return 0.0f;
        }
        float tan_sqr = tan_theta_sqr(w_h);
        if (std::isinf(tan_sqr)) return 0.0f;
        float c2 = cos_theta_sqr(w_h);
        float cos_theta_4 = c2 * c2, width_sqr = width * width;
        return std::exp(-tan_sqr / width_sqr) / (PI * width_sqr * cos_theta_4);
    }
    Vec3 sample(float u0, float u1) const {
        float tan_theta_sqr;
        if (ggx) {   // ggx.rs:37-43
            float t = width * std::sqrt(u0) / std::sqrt(1.0f - u0);
            tan_theta_sqr = t * t;
        } else {
            float log_sample = std::log(1.0f - u0);
            if (std::isinf(log_sample)) log_sample = 0.0f;
            tan_theta_sqr = -(width * width) * log_sample;
        }
        if (ggx) {
            float cos_t = 1.0f / std::sqrt(1.0f + tan_theta_sqr);
            float sin_t = std::sqrt(std::fmax(0.0f, 1.0f - cos_t * cos_t));
            float phi = 2.0f * PI * u1;
            return spherical_dir(sin_t, cos_t, phi);
        }
        float phi = 2.0f * PI * u1;
        float cos_t = 1.0f / std::sqrt(1.0f + tan_theta_sqr);
        float sin_t = std::sqrt(std::fmax(0.0f, 1.0f - cos_t * cos_t));
        return spherical_dir(sin_t, cos_t, phi);
    }
    float pdf(Vec3 w_h) const { return std::fabs(w_h.z) * normal_distribution(w_h); }
    float monodir_shadowing(Vec3 v) const {
        if (ggx) {   // ggx.rs:53-56
            float t = width * std::fabs(tan_theta(v));
            return 2.0f / (1.0f + std::sqrt(1.0f + t * t));
        }
        float a = 1.0f / (width * std::fabs(tan_theta(v)));
        if (a < 1.6f) {
            float a_sqr = a * a;
            return (3.535f * a + 2.181f * a_sqr) / (1.0f + 2.276f * a + 2.577f * a_sqr);
        }
        return 1.0f;
    }
    float shadowing_masking(Vec3 w_i, Vec3 w_o) const { return monodir_shadowing(w_i) * monodir_shadowing(w_o); }
};

enum LobeKind { LB_LAMBERTIAN, LB_OREN_NAYAR, LB_SPEC_REFL, LB_SPEC_TRANS, LB_TORRANCE_SPARROW, LB_MICROFACET_TRANS, LB_MERL };

struct Lobe {
    LobeKind kind;
    int type;          // BX_* bits
    Colorf color;      // reflectance / albedo / transmission
    float a = 0, b = 0;   // Oren-Nayar constants
    Fresnel fresnel;
    Beckmann mf;
    const float* merl = nullptr;
    uint32_t n_theta_h = 0, n_theta_d = 0, n_phi_d = 0;

    bool matches(int flags) const { return (type & ~flags) == 0; }   // is_subset (bxdf/mod.rs:108-110)

    Colorf eval(Vec3 w_o, Vec3 w_i) const {
        switch (kind) {
            case LB_LAMBERTIAN: return color * FRAC_1_PI;   // lambertian.rs:32-34
            case LB_OREN_NAYAR: {   // oren_nayar.rs:42-60
                float sin_o = sin_theta(w_o), sin_i = sin_theta(w_i);
                float max_cos = 0.0f;
                if (sin_i > 1e-4f && sin_o > 1e-4f)
                    max_cos = std::fmax(0.0f, cos_phi(w_i) * cos_phi(w_o) + sin_phi(w_i) * sin_phi(w_o));
                float sin_alpha, tan_beta;
                if (std::fabs(cos_theta(w_i)) > std::fabs(cos_theta(w_o))) {
                    sin_alpha = sin_o; tan_beta = sin_i / std::fabs(cos_theta(w_i));
                } else {
                    sin_alpha = sin_i; tan_beta = sin_o / std::fabs(cos_theta(w_o));
                }
                return color * FRAC_1_PI * (a + b * max_cos * sin_alpha * tan_beta);
            }
            case LB_SPEC_REFL:
            case LB_SPEC_TRANS: return Colorf::broadcast(0.0f);
            case LB_TORRANCE_SPARROW: {   // torrance_sparrow.rs:40-56
                float cos_to = std::fabs(cos_theta(w_o)), cos_ti = std::fabs(cos_theta(w_i));
                if (cos_to == 0.0f || cos_ti == 0.0f) return Colorf(0, 0, 0);
                Vec3 w_h = w_i + w_o;
                if (w_h == Vec3(0, 0, 0)) return Colorf(0, 0, 0);
                w_h = w_h.normalized();
                float d = mf.normal_distribution(w_h);
                Colorf f = fresnel.eval(dot(w_i, w_h));
                float g = mf.shadowing_masking(w_i, w_o);
                return color * f * d * g / (4.0f * cos_ti * cos_to);
            }
            case LB_MICROFACET_TRANS: {   // microfacet_transmission.rs:69-86
                if (same_hemisphere(w_o, w_i)) return Colorf::black();
                float cos_to = cos_theta(w_o), cos_ti = cos_theta(w_i);
                if (cos_to == 0.0f || cos_ti == 0.0f) return Colorf::black();
                float e0, e1;
                eta_for_interaction(w_o, e0, e1);
                Vec3 w_h = mt_half_vector(w_o, w_i, e0, e1);
                float d = mf.normal_distribution(w_h);
                Colorf f = Colorf::broadcast(1.0f) - fresnel.eval(dot(w_i, w_h));
                float g = mf.shadowing_masking(w_i, w_o);
                float wi_dot_h = dot(w_i, w_h);
                float jac = mt_jacobian(w_o, w_i, w_h, e0, e1);
                return color * (std::fabs(wi_dot_h) / (std::fabs(w_i.z) * std::fabs(w_o.z))) * (f * g * d) * jac;
            }
            case LB_MERL: return merl_eval(w_o, w_i);
        }
        return Colorf::black();
    }

    // default BxDF::pdf (bxdf/mod.rs:112-121) or the overrides
    float pdf(Vec3 w_o, Vec3 w_i) const {
        switch (kind) {
            case LB_TORRANCE_SPARROW: {   // torrance_sparrow.rs:71-81
                if (!same_hemisphere(w_o, w_i)) return 0.0f;
                Vec3 w_h = (w_o + w_i).normalized();
                float jac = 1.0f / (4.0f * std::fabs(dot(w_o, w_h)));
                return mf.pdf(w_h) * jac;
            }
            case LB_MICROFACET_TRANS: {   // microfacet_transmission.rs:101-108
                if (same_hemisphere(w_o, w_i)) return 0.0f;
                float e0, e1;
                eta_for_interaction(w_o, e0, e1);
                Vec3 w_h = mt_half_vector(w_o, w_i, e0, e1);
                return mf.pdf(w_h) * mt_jacobian(w_o, w_i, w_h, e0, e1);
            }
            default:   // SpecularReflection / SpecularTransmission do not override pdf either
                return same_hemisphere(w_o, w_i) ? std::fabs(cos_theta(w_i)) * FRAC_1_PI : 0.0f;
        }
    }

    // returns f; writes w_i and pdf
    Colorf sample(Vec3 w_o, float u0, float u1, Vec3& w_i, float& pdf_out) const {
        switch (kind) {
            case LB_SPEC_REFL: {   // specular_reflection.rs:39-50
                w_i = Vec3(-w_o.x, -w_o.y, w_o.z);
                if (w_i.z != 0.0f) {
                    pdf_out = 1.0f;
                    return fresnel.eval(cos_theta(w_o)) * color / std::fabs(cos_theta(w_i));
                }
                pdf_out = 0.0f;
                return Colorf::black();
            }
            case LB_SPEC_TRANS: {   // specular_transmission.rs:39-56
                bool entering = cos_theta(w_o) > 0.0f;
                float ei = entering ? fresnel.eta_i : fresnel.eta_t, et = entering ? fresnel.eta_t : fresnel.eta_i;
                Vec3 n = entering ? Vec3(0, 0, 1) : Vec3(0, 0, -1);
                if (refract(w_o, n, ei / et, w_i)) {
                    Colorf f = Colorf::broadcast(1.0f) - fresnel.eval(cos_theta(w_i));
                    pdf_out = 1.0f;
                    return f * color / std::fabs(cos_theta(w_i));
                }
                w_i = Vec3(0, 0, 0);
                pdf_out = 0.0f;
                return Colorf::black();
            }
            case LB_TORRANCE_SPARROW: {   // torrance_sparrow.rs:57-70
                if (w_o.z == 0.0f) { w_i = Vec3(0, 0, 0); pdf_out = 0.0f; return Colorf::black(); }
                Vec3 w_h = mf.sample(u0, u1);
                if (!same_hemisphere(w_o, w_h)) w_h = -w_h;
                w_i = reflect(w_o, w_h);
                if (!same_hemisphere(w_o, w_i)) { w_i = Vec3(0, 0, 0); pdf_out = 0.0f; return Colorf::black(); }
                pdf_out = pdf(w_o, w_i);
                return eval(w_o, w_i);
            }
            case LB_MICROFACET_TRANS: {   // microfacet_transmission.rs:87-100
                Vec3 w_h = mf.sample(u0, u1);
                if (!same_hemisphere(w_o, w_h)) w_h = -w_h;
                float e0, e1;
                eta_for_interaction(w_o, e0, e1);
                Vec3 wi;
                if (refract(w_o, w_h, e0 / e1, wi)) {
                    if (same_hemisphere(w_o, wi)) { w_i = Vec3(0, 0, 0); pdf_out = 0.0f; return Colorf::black(); }
                    w_i = wi;
                    pdf_out = pdf(w_o, w_i);
                    return eval(w_o, w_i);
                }
                w_i = Vec3(0, 0, 0);
                pdf_out = 0.0f;
                return Colorf::black();
            }
            default: {   // BxDF::sample default (bxdf/mod.rs:102-109): cosine hemisphere on w_o's side
                w_i = cos_sample_hemisphere(u0, u1);
                if (w_o.z < 0.0f) w_i.z *= -1.0f;
                pdf_out = pdf(w_o, w_i);
                return eval(w_o, w_i);
            }
        }
    }

    // microfacet_transmission.rs:34-60
    void eta_for_interaction(Vec3 w_o, float& e0, float& e1) const {
        if (cos_theta(w_o) > 0.0f) { e0 = fresnel.eta_i; e1 = fresnel.eta_t; } else { e0 = fresnel.eta_t; e1 = fresnel.eta_i; }
    }
    static float mt_jacobian(Vec3 w_o, Vec3 w_i, Vec3 w_h, float e0, float e1) {
        float wi_dot_h = dot(w_i, w_h), wo_dot_h = dot(w_o, w_h);
        float s = e1 * wi_dot_h + e0 * wo_dot_h;
        float denom = s * s;
        if (denom != 0.0f) return std::fabs(e0 * e0 * std::fabs(wo_dot_h) / denom);
        return 0.0f;
    }
    static Vec3 mt_half_vector(Vec3 w_o, Vec3 w_i, float e0, float e1) { return (-e1 * w_i - e0 * w_o).normalized(); }

    // bxdf/merl.rs:47-83
    static uint32_t map_index(float val, float max, uint32_t n_vals) {
        float f = val / max * (float)n_vals;
        uint64_t idx = f > 0.0f ? (f >= 1.8446744e19f ? ~0ull : (uint64_t)f) : 0ull;   // `as usize` saturates, NaN -> 0
        uint64_t hi = n_vals - 1;
        return (uint32_t)(idx > hi ? hi : idx);
    }
    Colorf merl_eval(Vec3 w_oi, Vec3 w_ii) const {
        Vec3 w_i = w_ii;
        Vec3 w_h = w_oi + w_i;
        if (w_h.z < 0.0f) { w_i = -w_i; w_h = -w_h; }
        if (w_h.length_sqr() == 0.0f) return Colorf::black();
        w_h = w_h.normalized();
        float theta_h = spherical_theta(w_h);
        float cos_phi_h = cos_phi(w_h), sin_phi_h = sin_phi(w_h);
        float cos_theta_h = cos_theta(w_h), sin_theta_h = sin_theta(w_h);
        Vec3 w_hx(cos_phi_h * cos_theta_h, sin_phi_h * cos_theta_h, -sin_theta_h);
        Vec3 w_hy(-sin_phi_h, cos_phi_h, 0.0f);
        Vec3 w_d(dot(w_i, w_hx), dot(w_i, w_hy), dot(w_i, w_h));
        float theta_d = spherical_theta(w_d);
        float phi_d = spherical_phi(w_d);
        if (phi_d > PI) phi_d = phi_d - PI;
        uint32_t theta_h_idx = map_index(std::sqrt(std::fmax(0.0f, 2.0f * theta_h / PI)), 1.0f, n_theta_h);
        uint32_t theta_d_idx = map_index(theta_d, PI / 2.0f, n_theta_d);
        uint32_t phi_d_idx = map_index(phi_d, PI, n_phi_d);
        size_t i = phi_d_idx + (size_t)n_phi_d * (theta_d_idx + (size_t)theta_h_idx * n_theta_d);
        return Colorf(merl[3 * i], merl[3 * i + 1], merl[3 * i + 2]);
    }
};

// bxdf/bsdf.rs
struct BSDF {
    Vec3 p, n, ng, tan, bitan;
    float eta = 1.0f;
    Lobe lobes[2];
    int n_lobes = 0;

    void set_frame(const Hit& dg) {   // BSDF::new (bsdf.rs:38-44), quirk Q8
        n = dg.n.normalized();
        Vec3 bt = dg.dp_du.normalized();
        tan = cross(n, bt);
        bitan = cross(tan, n);
        p = dg.p;
        ng = dg.ng;
    }
    int num_matching(int flags) const { int c = 0; for (int i = 0; i < n_lobes; ++i) c += lobes[i].matches(flags); return c; }
    Vec3 to_shading(Vec3 v) const { return Vec3(dot(v, bitan), dot(v, tan), dot(v, n)); }
    Vec3 from_shading(Vec3 v) const {
        return Vec3(bitan.x * v.x + tan.x * v.y + n.x * v.z, bitan.y * v.x + tan.y * v.y + n.y * v.z, bitan.z * v.x + tan.z * v.y + n.z * v.z);
    }
    Colorf eval(Vec3 wo_world, Vec3 wi_world, int flags) const {   // bsdf.rs:66-79
        Vec3 w_o = to_shading(wo_world).normalized(), w_i = to_shading(wi_world).normalized();
        if (w_o.z * w_i.z > 0.0f) flags &= ~BX_TRANSMISSION; else flags &= ~BX_REFLECTION;
        Colorf sum = Colorf::broadcast(0.0f);
        for (int i = 0; i < n_lobes; ++i)
            if (lobes[i].matches(flags)) sum = sum + lobes[i].eval(w_o, w_i);
        return sum;
    }
    float pdf(Vec3 wo_world, Vec3 wi_world, int flags) const {   // bsdf.rs:114-125
        Vec3 w_o = to_shading(wo_world).normalized(), w_i = to_shading(wi_world).normalized();
        float pdf_val = 0.0f;
        int n_comps = 0;
        for (int i = 0; i < n_lobes; ++i)
            if (lobes[i].matches(flags)) { pdf_val = pdf_val + lobes[i].pdf(w_o, w_i); ++n_comps; }
        return n_comps > 0 ? pdf_val / (float)n_comps : 0.0f;
    }
    // bsdf.rs:85-111: returns f, writes wi_world, pdf and the sampled lobe's type bits
    Colorf sample(Vec3 wo_world, int flags, float u0, float u1, float one_d, Vec3& wi_world, float& pdf_out, int& sampled_type) const {
        int n_matching = num_matching(flags);
        if (n_matching == 0) { wi_world = Vec3(0, 0, 0); pdf_out = 0.0f; sampled_type = 0; return Colorf::broadcast(0.0f); }
        float fc = one_d * (float)n_matching;
        int comp = fc > 0.0f ? (int)fc : 0;
        if (comp > n_matching - 1) comp = n_matching - 1;
        const Lobe* bxdf = nullptr;
        for (int i = 0, k = 0; i < n_lobes; ++i)
            if (lobes[i].matches(flags)) { if (k == comp) { bxdf = &lobes[i]; break; } ++k; }
        Vec3 w_o = to_shading(wo_world).normalized();
        Vec3 w_i;
        float pdf_v;
        Colorf f = bxdf->sample(w_o, u0, u1, w_i, pdf_v);
        if (w_i.length_sqr() == 0.0f) { wi_world = Vec3(0, 0, 0); pdf_out = 0.0f; sampled_type = 0; return Colorf::broadcast(0.0f); }
        wi_world = from_shading(w_i).normalized();
        bool specular = (bxdf->type & BX_SPECULAR) != 0;
        if (!specular && n_matching > 1) pdf_v = pdf(wo_world, wi_world, flags);
        if (!specular) f = eval(wo_world, wi_world, flags);
        pdf_out = pdf_v;
        sampled_type = bxdf->type;
        return f;
    }
};

// ---- image textures (src/texture/mod.rs:22-40, image.rs:9-47, animated_image.rs:7-58) --------------------------------
// `x as u32` of Rust: saturating, NaN -> 0
inline uint32_t f32_as_u32(float x) { return x > 0.0f ? (x >= 4294967296.0f ? 0xffffffffu : (uint32_t)x) : 0u; }
inline Colorf tex_get_color(const TrayFlatScene& fs, const TrayTexFrame& fr, uint32_t x, uint32_t y) {   // Image::get_color
    x = x > fr.width - 1 ? fr.width - 1 : x;   // clamp(x, 0, dims.0 - 1) on u32
    y = y > fr.height - 1 ? fr.height - 1 : y;
    const uint8_t* px = fs.tex_data + fr.offset + ((size_t)y * fr.width + x) * 4;
    return Colorf((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f, (float)px[3] / 255.0f);
}
// bilinear_interpolate (texture/mod.rs:22-40): the four texels around (x, y), NOT centred on texel centres
template <class T, class Get>
inline T tex_bilinear(float x, float y, Get get) {
    const uint32_t x0 = f32_as_u32(x), y0 = f32_as_u32(y);
    const T s00 = get(x0, y0), s10 = get(x0 + 1u, y0), s01 = get(x0, y0 + 1u), s11 = get(x0 + 1u, y0 + 1u);
    const float sx = x - (float)x0, sy = y - (float)y0;
    return s00 * (1.0f - sx) * (1.0f - sy) + s10 * sx * (1.0f - sy) + s01 * (1.0f - sx) * sy + s11 * sx * sy;
}
inline Colorf image_sample_color(const TrayFlatScene& fs, const TrayTexFrame& fr, float u, float v) {
    const float x = u * (float)fr.width, y = v * (float)fr.height;
    return tex_bilinear<Colorf>(x, y, [&](uint32_t px, uint32_t py) { return tex_get_color(fs, fr, px, py); });
}
inline float image_sample_f32(const TrayFlatScene& fs, const TrayTexFrame& fr, float u, float v) {   // get_float: data[0] / 255
    const float x = u * (float)fr.width, y = v * (float)fr.height;
    return tex_bilinear<float>(x, y, [&](uint32_t px, uint32_t py) { return tex_get_color(fs, fr, px, py).r; });
}
// AnimatedImage::active_keyframes (binary_search_by on the frame times)
inline void active_keyframes(const TrayTexFrame* fr, uint32_t n, float time, uint32_t& lo, bool& two) {
    uint32_t a = 0, b = n;   // first index with fr[i].time >= time, or an exact match
    two = false;
    while (a < b) {
        const uint32_t mid = a + (b - a) / 2;
        if (fr[mid].time == time) { lo = mid; return; }
        if (fr[mid].time < time) a = mid + 1; else b = mid;
    }
    if (a == n) lo = n - 1;
    else if (a == 0) lo = 0;
    else { lo = a - 1; two = true; }
}
inline Colorf texture_sample_color(const TrayFlatScene& fs, uint32_t tex, float u, float v, float time) {
    const TrayTexture& t = fs.textures[tex];
    const TrayTexFrame* fr = fs.tex_frames + t.first_frame;
    if (t.n_frames < 2) return image_sample_color(fs, fr[0], u, v);
    uint32_t lo; bool two;
    active_keyframes(fr, t.n_frames, time, lo, two);
    if (!two) return image_sample_color(fs, fr[lo], u, v);
    const float x = (time - fr[lo].time) / (fr[lo + 1].time - fr[lo].time);
    return image_sample_color(fs, fr[lo], u, v) * (1.0f - x) + image_sample_color(fs, fr[lo + 1], u, v) * x;   // linalg::lerp
}
inline float texture_sample_f32(const TrayFlatScene& fs, uint32_t tex, float u, float v, float time) {
    const TrayTexture& t = fs.textures[tex];
    const TrayTexFrame* fr = fs.tex_frames + t.first_frame;
    if (t.n_frames < 2) return image_sample_f32(fs, fr[0], u, v);
    uint32_t lo; bool two;
    active_keyframes(fr, t.n_frames, time, lo, two);
    if (!two) return image_sample_f32(fs, fr[lo], u, v);
    const float x = (time - fr[lo].time) / (fr[lo + 1].time - fr[lo].time);
    return image_sample_f32(fs, fr[lo], u, v) * (1.0f - x) + image_sample_f32(fs, fr[lo + 1], u, v) * x;
}

// Material::bsdf for the seven materials (src/material/*.rs): every parameter is texture.sample_color / sample_f32 at
// (hit.dg.u, hit.dg.v, hit.dg.time) (e.g. matte.rs:55-56); constants are ConstantColor / ConstantScalar
inline BSDF material_bsdf(const TrayFlatScene& fs, const Hit& hit) {
    TrayMaterial m = fs.materials[fs.instances[hit.inst].material_id];
    BSDF b;
    b.set_frame(hit);
    Colorf c0(m.c0[0], m.c0[1], m.c0[2], m.c0[3]), c1(m.c1[0], m.c1[1], m.c1[2], m.c1[3]);
    if (m.tex_c0 != TRAY_NO_TEXTURE) c0 = texture_sample_color(fs, m.tex_c0, hit.u, hit.v, hit.time);
    if (m.tex_c1 != TRAY_NO_TEXTURE) c1 = texture_sample_color(fs, m.tex_c1, hit.u, hit.v, hit.time);
    if (m.tex_f0 != TRAY_NO_TEXTURE) m.f0 = texture_sample_f32(fs, m.tex_f0, hit.u, hit.v, hit.time);
    if (m.tex_f1 != TRAY_NO_TEXTURE) m.f1 = texture_sample_f32(fs, m.tex_f1, hit.u, hit.v, hit.time);
    auto dielectric = [](float ei, float et) { Fresnel f{}; f.kind = FR_DIELECTRIC; f.eta_i = ei; f.eta_t = et; return f; };
    auto conductor = [](Colorf eta, Colorf k) { Fresnel f{}; f.kind = FR_CONDUCTOR; f.eta = eta; f.k = k; return f; };
    switch (m.kind) {
        case TRAY_MAT_MATTE: {   // matte.rs:52-65
            Lobe l{};
            l.color = c0;
            l.type = BX_DIFFUSE | BX_REFLECTION;
            if (m.f0 == 0.0f) {
                l.kind = LB_LAMBERTIAN;
            } else {   // OrenNayar::new (oren_nayar.rs:26-34)
                l.kind = LB_OREN_NAYAR;
                float sigma = to_radians(m.f0);
                sigma *= sigma;
                l.a = 1.0f - 0.5f * sigma / (sigma + 0.33f);
                l.b = 0.45f * sigma / (sigma + 0.09f);
            }
            b.lobes[b.n_lobes++] = l;
            b.eta = 1.0f;
            break;
        }
        case TRAY_MAT_PLASTIC: {   // plastic.rs:59-88
            if (!c0.is_black()) { Lobe l{}; l.kind = LB_LAMBERTIAN; l.type = BX_DIFFUSE | BX_REFLECTION; l.color = c0; b.lobes[b.n_lobes++] = l; }
            if (!c1.is_black()) {
                Lobe l{}; l.kind = LB_TORRANCE_SPARROW; l.type = BX_GLOSSY | BX_REFLECTION; l.color = c1;
                l.fresnel = dielectric(1.0f, 1.5f); l.mf = Beckmann::make(m.f0, m.microfacet == TRAY_MF_GGX);
                b.lobes[b.n_lobes++] = l;
            }
            b.eta = 1.0f;
            break;
        }
        case TRAY_MAT_METAL: {   // metal.rs:56-67
            Lobe l{}; l.kind = LB_TORRANCE_SPARROW; l.type = BX_GLOSSY | BX_REFLECTION; l.color = Colorf::broadcast(1.0f);
            l.fresnel = conductor(c0, c1); l.mf = Beckmann::make(m.f0, m.microfacet == TRAY_MF_GGX);
            b.lobes[b.n_lobes++] = l;
            b.eta = 1.0f;
            break;
        }
        case TRAY_MAT_GLASS: {   // glass.rs:51-78
            Fresnel fr = dielectric(1.0f, m.f0);
            if (!c0.is_black()) { Lobe l{}; l.kind = LB_SPEC_REFL; l.type = BX_SPECULAR | BX_REFLECTION; l.color = c0; l.fresnel = fr; b.lobes[b.n_lobes++] = l; }
            if (!c1.is_black()) { Lobe l{}; l.kind = LB_SPEC_TRANS; l.type = BX_SPECULAR | BX_TRANSMISSION; l.color = c1; l.fresnel = fr; b.lobes[b.n_lobes++] = l; }
            b.eta = m.f0;
            break;
        }
        case TRAY_MAT_ROUGH_GLASS: {   // rough_glass.rs:57-85
            Fresnel fr = dielectric(1.0f, m.f0);
            Beckmann mf = Beckmann::make(m.f1, m.microfacet == TRAY_MF_GGX);
            if (!c0.is_black()) { Lobe l{}; l.kind = LB_TORRANCE_SPARROW; l.type = BX_GLOSSY | BX_REFLECTION; l.color = c0; l.fresnel = fr; l.mf = mf; b.lobes[b.n_lobes++] = l; }
            if (!c1.is_black()) { Lobe l{}; l.kind = LB_MICROFACET_TRANS; l.type = BX_GLOSSY | BX_TRANSMISSION; l.color = c1; l.fresnel = fr; l.mf = mf; b.lobes[b.n_lobes++] = l; }
            b.eta = m.f0;
            break;
        }
        case TRAY_MAT_SPECULAR_METAL: {   // specular_metal.rs:49-58
            Lobe l{}; l.kind = LB_SPEC_REFL; l.type = BX_SPECULAR | BX_REFLECTION; l.color = Colorf::broadcast(1.0f); l.fresnel = conductor(c0, c1);
            b.lobes[b.n_lobes++] = l;
            b.eta = 1.0f;
            break;
        }
        case TRAY_MAT_MERL: {   // material/merl.rs:88-92
            const TrayMerlTable& t = fs.merl_tables[m.table];
            Lobe l{}; l.kind = LB_MERL; l.type = BX_GLOSSY | BX_REFLECTION;
            l.merl = fs.merl_data + t.offset; l.n_theta_h = t.n_theta_h; l.n_theta_d = t.n_theta_d; l.n_phi_d = t.n_phi_d;
            b.lobes[b.n_lobes++] = l;
            b.eta = 1.0f;
            break;
        }
    }
    return b;
}

}  // namespace orc
