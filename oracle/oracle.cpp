// TEST INFRASTRUCTURE ONLY — CPU oracle entry points (liboracle.so). See oracle_math.hpp.
// Camera (film/camera.rs:150-157), Path integrator (integrator/path.rs:45-120, mod.rs:106-169),
// tile worker (exec/multithreaded.rs:72-114), film (film/render_target.rs:77-165) and the C entry
// points tests/ and bench.py's cpu_baseline use. PARITY UNPINNED (no reference golden vectors).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/trayhip.h"
#include "oracle_math.hpp"
#include "oracle_scene.hpp"

using namespace orc;

namespace {

// Camera::generate_ray (camera.rs:150-157)
Ray camera_generate_ray(const SceneView& sv, float px, float py, float time) {
    const TrayFlatScene& fs = *sv.fs;
    const TrayCamera& c = fs.camera;
    Vec3 q = Transform::mul_point(Mat4::from(c.raster_to_cam), Vec3(px, py, 0.0f));
    Vec3 px_pos = Vec3(c.scaling[0], c.scaling[1], c.scaling[2]) * q;
    Vec3 d = px_pos.normalized();
    float frame_time = (c.shutter_close - c.shutter_open) * time + c.shutter_open;
    // cam_world.transform(frame_time) * Ray (camera.rs:156)
    Mat4 cw = ((sv.flags & ORC_FAITHFUL_XF) || c.animated) ? sv.stack_transform(c.xf_first, c.xf_count, frame_time).mat : Mat4::from(c.cam_world);
    Ray r;
    r.o = Transform::mul_point(cw, Vec3(0, 0, 0));
    r.d = Transform::mul_vector(cw, d);
    r.min_t = 0.0f; r.max_t = INF; r.time = frame_time;
    return r;
}

// Per-camera-sample LD arrays of path.rs:48-60 generated from the counter-based RNG (TRAY-CBRNG v3):
// every array has its scramble word(s) and is shuffled (ld.rs:58,63) by the composition of two of 256 pool permutations of [0, n),
// perm_q2[perm_q1[.]]: q1 = the low byte of the array's first scramble word, q2 = the low byte of its second one (2-D arrays) or
// bits 8..15 of its only one (1-D arrays). Pool permutation q is the Fisher-Yates shuffle shuffle_small(mix32(0x50455250 + q), n).
struct PermPool {
    uint32_t n = 0;
    uint8_t perm[256][16];
    void build(uint32_t num) {
        n = num;
        for (uint32_t q = 0; q < 256u; ++q) shuffle_small(mix32(0x50455250u + q), n, perm[q]);
    }
};
inline const PermPool& perm_pool(uint32_t n) {
    static thread_local PermPool pools[17];
    PermPool& p = pools[n <= 16u ? n : 16u];
    if (p.n != n) p.build(n);
    return p;
}
struct PathSamples {
    uint32_t ks;
    uint32_t n;
    uint32_t scr[9];
    const uint8_t* perm[6];
    uint8_t own[6][16];   // this path's six shuffles (the composed pool permutations; ORC_FRESH_SHUFFLES: six Fisher-Yates shuffles of its own)
    // which Sampler fills the arrays (sampler/mod.rs:20-49): LowDiscrepancy (ld.rs:54-64); Adaptive -- the same with the (0,2) points
    // starting at index samples_taken (adaptive.rs:112-121: ld::sample_2d(samples, scramble, self.samples_taken)); Uniform -- every entry
    // Range::new(0.0, 1.0).ind_sample(rng) (uniform.rs:36-46), here next_f32 of draw(ks, 64 + 32 * dim + 2 * entry (+ 1 for the second coordinate))
    uint32_t kind = TRAY_SAMPLER_LOW_DISCREPANCY, offset = 0;
    void init(uint32_t key_samp, uint32_t num_samples, bool fresh = false, uint32_t sampler_kind = TRAY_SAMPLER_LOW_DISCREPANCY, uint32_t samples_taken = 0) {
        ks = key_samp; n = num_samples; kind = sampler_kind; offset = sampler_kind == TRAY_SAMPLER_ADAPTIVE ? samples_taken : 0u;
        const PermPool& pool = perm_pool(n);
        // 2-D arrays: scramble x, scramble y; 1-D arrays: scramble
        const int d2[3] = {SD_L2, SD_B2, SD_P2}, d1[3] = {SD_L1, SD_B1, SD_P1};
        auto compose = [&](int a, uint32_t q1, uint32_t q2) {
            for (uint32_t b = 0; b < 16u; ++b) own[a][b] = b < (n <= 16u ? n : 16u) ? pool.perm[q2 & 255u][pool.perm[q1 & 255u][b]] : (uint8_t)b;
            perm[a] = own[a];
        };
        for (int a = 0; a < 3; ++a) {
            scr[2 * a] = draw(ks, d2[a]); scr[2 * a + 1] = draw(ks, d2[a] + 1);
            compose(a, scr[2 * a], scr[2 * a + 1]);
        }
        for (int a = 0; a < 3; ++a) {
            scr[6 + a] = draw(ks, d1[a]);
            compose(3 + a, scr[6 + a], scr[6 + a] >> 8);
        }
        if (fresh)   // Rng::shuffle per array (ld.rs:58,63), each under a key of its own
            for (int a = 0; a < 6; ++a) { shuffle_small(draw(ks, 40u + (uint32_t)a), n <= 16u ? n : 16u, own[a]); perm[a] = own[a]; }
    }
    float uniform(uint32_t dim, uint32_t bounce, uint32_t c) const { return (float)(draw(ks, 64u + 32u * dim + 2u * bounce + c) >> 8) / (float)(1u << 24); }
    void two_d(int a, uint32_t bounce, float& u0, float& u1) const {   // sample_02 (ld.rs:91-93)
        const uint32_t d2[3] = {SD_L2, SD_B2, SD_P2};
        if (kind == TRAY_SAMPLER_UNIFORM) { u0 = uniform(d2[a], bounce, 0u); u1 = uniform(d2[a], bounce, 1u); return; }
        uint32_t idx = perm[a][bounce] + offset;
        u0 = van_der_corput(idx, scr[2 * a]);
        u1 = sobol(idx, scr[2 * a + 1]);
    }
    float one_d(int a, uint32_t bounce) const {
        const uint32_t d1[3] = {SD_L1, SD_B1, SD_P1};
        if (kind == TRAY_SAMPLER_UNIFORM) return uniform(d1[a], bounce, 0u);
        return van_der_corput(perm[3 + a][bounce] + offset, scr[6 + a]);
    }
    float rr(uint32_t bounce) const { return (float)(draw(ks, SD_RR + bounce) >> 8) / (float)(1u << 24); }   // Rng::next_f32
};

// Schedule analysis hook (tools/simulate_query_compaction.py): which BSDF queries each path vertex needed
// ORACLE_TRACE=1 (set before the first call): per-vertex lines on stderr, to line a sample up against tests/emu's emu_trace_sample
static bool oracle_trace() { static const bool on = getenv("ORACLE_TRACE") != nullptr; return on; }
struct VertexLog { std::vector<uint8_t>* sink = nullptr; bool light = false, mis = false; };
thread_local VertexLog g_vlog;

// Integrator::estimate_direct (integrator/mod.rs:122-169)
Colorf estimate_direct(const SceneView& sv, Vec3 w_o, Vec3 p, const BSDF& bsdf, const float l2[2], const float b2[2], float b1,
                       uint32_t light_inst, int flags, float time) {
    const TrayInstance& light = sv.fs->instances[light_inst];
    bool delta = light.kind == TRAY_INST_POINT_EMITTER;
    Colorf direct_light = Colorf::black();
    LightSample ls = light_sample_incident(sv, light_inst, bsdf.p, l2[0], l2[1], time);
    bool unoccluded = false;
    if (ls.pdf > 0.0f && !ls.li.is_black()) {
        Ray r = ls.occlusion;
        Hit tmp;
        unoccluded = !scene_intersect(sv, r, tmp);
    }
    g_vlog.light = unoccluded; g_vlog.mis = !delta;
    if (oracle_trace()) fprintf(stderr, "oracle    light sample: pdf %.9g li %.9g %.9g %.9g unoccluded %d w_i %.9g %.9g %.9g\n", ls.pdf, ls.li.r, ls.li.g, ls.li.b, (int)unoccluded, ls.w_i.x, ls.w_i.y, ls.w_i.z);
    if (unoccluded) {
        Colorf f = bsdf.eval(w_o, ls.w_i, flags);
        if (!f.is_black()) {
            if (delta) {
                direct_light = f * ls.li * std::fabs(dot(ls.w_i, bsdf.n)) / ls.pdf;
            } else {
                float pdf_bsdf = bsdf.pdf(w_o, ls.w_i, flags);
                float w = power_heuristic(1.0f, ls.pdf, 1.0f, pdf_bsdf);
                direct_light = f * ls.li * std::fabs(dot(ls.w_i, bsdf.n)) * w / ls.pdf;
                if (oracle_trace()) fprintf(stderr, "oracle    light half: f %.9g %.9g %.9g pdf_bsdf %.9g w %.9g -> %.9g %.9g %.9g\n", f.r, f.g, f.b, pdf_bsdf, w, direct_light.r, direct_light.g, direct_light.b);
            }
        }
    }
    if (!delta) {
        Vec3 w_i;
        float pdf_bsdf;
        int sampled_type;
        Colorf f = bsdf.sample(w_o, flags, b2[0], b2[1], b1, w_i, pdf_bsdf, sampled_type);
        if (pdf_bsdf > 0.0f && !f.is_black()) {
            float w = 1.0f;
            if (!(sampled_type & BX_SPECULAR)) {
                float pdf_light = light_pdf(sv, light_inst, p, w_i, time);
                if (pdf_light == 0.0f) return direct_light;
                w = power_heuristic(1.0f, pdf_bsdf, 1.0f, pdf_light);
            }
            Ray ray;
            ray.o = p; ray.d = w_i; ray.min_t = 0.001f; ray.max_t = INF; ray.time = time;
            Colorf li = Colorf::black();
            Hit h;
            if (scene_intersect(sv, ray, h)) {
                if (h.inst == light_inst)   // same emitter object (mod.rs:157-160)
                    li = emitter_radiance(sv, light, -w_i, h.ng, time);
            }
            if (oracle_trace()) fprintf(stderr, "oracle    bsdf half: pdf_bsdf %.9g w %.9g f %.9g %.9g %.9g li %.9g %.9g %.9g\n", pdf_bsdf, w, f.r, f.g, f.b, li.r, li.g, li.b);
            if (!li.is_black()) direct_light = direct_light + f * li * std::fabs(dot(w_i, bsdf.n)) * w / pdf_bsdf;
        }
    }
    return direct_light;
}

// Path::illumination (integrator/path.rs:45-120)
Colorf path_illumination(const SceneView& sv, const Ray& r, const Hit& hit, const PathSamples& ps) {
    const TrayFlatScene& fs = *sv.fs;
    Colorf illum = Colorf::black();
    Colorf path_throughput = Colorf::broadcast(1.0f);
    bool specular_bounce = false;
    Hit current_hit = hit;
    Ray ray = r;
    uint32_t bounce = 0;
    for (;;) {
        if (sv.stats) sv.stats->vertices++;
        const TrayInstance& inst = fs.instances[current_hit.inst];
        if (bounce == 0 || specular_bounce) {
            if (inst.kind != TRAY_INST_RECEIVER) {
                Vec3 w = -ray.d;
                illum = illum + path_throughput * emitter_radiance(sv, inst, w, hit.ng, ray.time);   // first hit's ng (quirk Q1)
            }
        }
        BSDF bsdf = material_bsdf(fs, current_hit);
        Vec3 w_o = -ray.d;
        float l2[2], b2[2], p2[2];
        ps.two_d(0, bounce, l2[0], l2[1]);
        ps.two_d(1, bounce, b2[0], b2[1]);
        float l1 = ps.one_d(0, bounce), b1 = ps.one_d(1, bounce);
        // sample_one_light (mod.rs:106-111); no 1/p_select factor (quirk Q6)
        float fl = l1 * (float)fs.n_lights;
        uint32_t li_idx = fl > 0.0f ? (uint32_t)std::min<double>((double)fl, 4294967295.0) : 0u;
        if (li_idx > fs.n_lights - 1) li_idx = fs.n_lights - 1;
        Colorf li = estimate_direct(sv, w_o, current_hit.p, bsdf, l2, b2, b1, fs.lights[li_idx], BX_NON_SPECULAR, ray.time);
        illum = illum + path_throughput * li;
        if (oracle_trace()) fprintf(stderr, "oracle  bounce %u inst %u mat %u li %.9g %.9g %.9g  throughput %.9g %.9g %.9g  illum %.9g %.9g %.9g\n", bounce, current_hit.inst,
                                            inst.material_id, li.r, li.g, li.b, path_throughput.r, path_throughput.g, path_throughput.b, illum.r, illum.g, illum.b);
        if (g_vlog.sink)   // material (5 bits) | light query ran (32) | BSDF-half query ran (64)
            g_vlog.sink->push_back((uint8_t)((inst.material_id & 31u) | (g_vlog.light ? 32u : 0u) | (g_vlog.mis ? 64u : 0u)));

        ps.two_d(2, bounce, p2[0], p2[1]);
        float p1 = ps.one_d(2, bounce);
        Vec3 w_i;
        float pdf;
        int sampled_type;
        Colorf f = bsdf.sample(w_o, BX_ALL, p2[0], p2[1], p1, w_i, pdf, sampled_type);
        if (f.is_black() || pdf == 0.0f) break;
        specular_bounce = (sampled_type & BX_SPECULAR) != 0;
        path_throughput = path_throughput * f * std::fabs(dot(w_i, bsdf.n)) / pdf;
        if (bounce > fs.min_depth) {   // quirk Q2
            float cont_prob = std::fmax(0.5f, path_throughput.luminance());
            if (ps.rr(bounce) > cont_prob) break;
            path_throughput = path_throughput / cont_prob;
        }
        if (bounce == fs.max_depth) break;
        Ray child;
        child.o = bsdf.p; child.d = w_i.normalized(); child.min_t = 0.001f; child.max_t = INF; child.time = ray.time;
        ray = child;
        Hit h;
        if (!scene_intersect(sv, ray, h)) break;
        current_hit = h;
        bounce += 1;
    }
    return illum;
}

// Whitted::illumination (integrator/whitted.rs:41-68) with Integrator::specular_reflection / specular_transmission
// (integrator/mod.rs:49-97), recursive as in the reference. Random numbers: every activation is a node with a key; a one-element
// LowDiscrepancy::get_samples_2d / _1d (ld.rs:54-64) is the scrambled (0,2) point of index 0 with fresh scrambles
enum : uint32_t { WD_L2 = 0, WD_R2 = 2, WD_R1 = 4, WD_T2 = 5, WD_T1 = 7, WD_CHILD = 8 };
// (a one-element array under the scene's Sampler: LowDiscrepancy -- the point of index 0; Adaptive -- of index samples_taken; Uniform -- a uniform draw)
inline float whitted_vdc(const SceneView& sv, uint32_t key, uint32_t d) {
    if (sv.smp_kind == TRAY_SAMPLER_UNIFORM) return (float)(draw(key, d) >> 8) / (float)(1u << 24);
    return van_der_corput(sv.smp_kind == TRAY_SAMPLER_ADAPTIVE ? sv.smp_offset : 0u, draw(key, d));
}
inline float whitted_sobol(const SceneView& sv, uint32_t key, uint32_t d) {
    if (sv.smp_kind == TRAY_SAMPLER_UNIFORM) return (float)(draw(key, d) >> 8) / (float)(1u << 24);
    return sobol(sv.smp_kind == TRAY_SAMPLER_ADAPTIVE ? sv.smp_offset : 0u, draw(key, d));
}
Colorf whitted_illumination(const SceneView& sv, const Ray& ray, const Hit& hit, uint32_t key, uint32_t depth);
Colorf whitted_specular(const SceneView& sv, const Ray& ray, const BSDF& bsdf, uint32_t key, uint32_t depth, bool transmission) {
    const Vec3 w_o = -ray.d;
    const uint32_t d2 = transmission ? WD_T2 : WD_R2, d1 = transmission ? WD_T1 : WD_R1;
    const float u0 = whitted_vdc(sv, key, d2), u1 = whitted_sobol(sv, key, d2 + 1u), one_d = whitted_vdc(sv, key, d1);
    Vec3 w_i;
    float pdf;
    int sampled_type;
    const Colorf f = bsdf.sample(w_o, BX_SPECULAR | (transmission ? BX_TRANSMISSION : BX_REFLECTION), u0, u1, one_d, w_i, pdf, sampled_type);
    Colorf out = Colorf::broadcast(0.0f);
    if (pdf > 0.0f && !f.is_black() && std::fabs(dot(w_i, bsdf.n)) != 0.0f) {
        Ray child;   // ray.child(&bsdf.p, &w_i), min_t = 0.001 (mod.rs:65-66)
        child.o = bsdf.p; child.d = w_i; child.min_t = 0.001f; child.max_t = INF; child.time = ray.time;
        Hit h;
        if (scene_intersect(sv, child, h)) {
            const Colorf li = whitted_illumination(sv, child, h, draw(key, WD_CHILD + (transmission ? 1u : 0u)), depth + 1u);
            out = f * li * std::fabs(dot(w_i, bsdf.n)) / pdf;
        }
    }
    return out;
}
Colorf whitted_illumination(const SceneView& sv, const Ray& ray, const Hit& hit, uint32_t key, uint32_t depth) {
    const TrayFlatScene& fs = *sv.fs;
    if (sv.stats) sv.stats->vertices++;
    const BSDF bsdf = material_bsdf(fs, hit);
    const Vec3 w_o = -ray.d;
    const float u0 = whitted_vdc(sv, key, WD_L2), u1 = whitted_sobol(sv, key, WD_L2 + 1u);
    Colorf illum = Colorf::broadcast(0.0f);
    const TrayInstance& inst = fs.instances[hit.inst];
    if (depth == 0u && inst.kind != TRAY_INST_RECEIVER) illum = illum + emitter_radiance(sv, inst, w_o, hit.ng, ray.time);
    for (uint32_t k = 0; k < fs.n_lights; ++k) {   // every light, the same sample (whitted.rs:56-62)
        const LightSample ls = light_sample_incident(sv, fs.lights[k], bsdf.p, u0, u1, ray.time);
        const Colorf f = bsdf.eval(w_o, ls.w_i, BX_ALL);
        if (ls.li.is_black() || f.is_black()) continue;
        Ray r = ls.occlusion;
        Hit tmp;
        if (!scene_intersect(sv, r, tmp)) illum = illum + f * ls.li * std::fabs(dot(ls.w_i, bsdf.n)) / ls.pdf;
    }
    if (depth < fs.max_depth) {
        illum = illum + whitted_specular(sv, ray, bsdf, key, depth, false);
        illum = illum + whitted_specular(sv, ray, bsdf, key, depth, true);
    }
    return illum;
}

struct PixelSampler {   // per-pixel part of LowDiscrepancy (ld.rs:33-64) over the counter RNG
    uint32_t kp, spp, scr_x, scr_y, key_xy, scr_t, key_t;
    void init(uint32_t kf, uint32_t pixel_index, uint32_t spp_) {
        kp = key_pixel(kf, pixel_index); spp = spp_;
        scr_x = draw(kp, PD_SCR_X); scr_y = draw(kp, PD_SCR_Y); key_xy = draw(kp, PD_PERM_XY);
        scr_t = draw(kp, PD_SCR_T); key_t = draw(kp, PD_PERM_T);
    }
    void position(uint32_t s, uint32_t px, uint32_t py, float& x, float& y) const {
        uint32_t idx = permute(s, spp, key_xy);
        x = van_der_corput(idx, scr_x); y = sobol(idx, scr_y);
        x += (float)px; y += (float)py;
    }
    float time(uint32_t s) const { return van_der_corput(permute(s, spp, key_t), scr_t); }
};

// The body of thread_work's inner loop for one camera sample (multithreaded.rs:95-103): the clamped colour of the sample at film
// position (sx, sy) and time t; ks = the key every random number of the path derives from
Colorf sample_radiance(const SceneView& sv, float sx, float sy, float t, uint32_t ks) {
    const TrayFlatScene& fs = *sv.fs;
    if (sv.stats) sv.stats->samples++;
    Ray ray = camera_generate_ray(sv, sx, sy, t);
    Hit hit;
    if (scene_intersect(sv, ray, hit)) {
        if (sv.fs->integrator == TRAY_INTEGRATOR_WHITTED) return whitted_illumination(sv, ray, hit, ks, 0u).clamp();
        PathSamples ps;
        ps.init(ks, fs.max_depth + 1, (sv.flags & ORC_FRESH_SHUFFLES) != 0, sv.smp_kind, sv.smp_offset);
        if (sv.fs->integrator == TRAY_INTEGRATOR_NORMALS_DEBUG) {   // NormalsDebug::illumination (integrator/normals_debug.rs:28-33)
            if (sv.stats) sv.stats->vertices++;
            const BSDF bsdf = material_bsdf(*sv.fs, hit);
            return ((Colorf(bsdf.n.x, bsdf.n.y, bsdf.n.z) + Colorf::broadcast(1.0f)) / 2.0f).clamp();
        }
        return path_illumination(sv, ray, hit, ps).clamp();   // quirk Q3
    }
    return Colorf::black();
}
// ... with LowDiscrepancy's positions and times (ld.rs:33-64): sample s of pixel (px, py)
Colorf trace_sample(const SceneView& sv, uint32_t kf, uint32_t px, uint32_t py, uint32_t s, uint32_t spp, float& sx, float& sy) {
    PixelSampler pix;
    pix.init(kf, py * sv.fs->film.width + px, spp);
    pix.position(s, px, py, sx, sy);
    return sample_radiance(sv, sx, sy, pix.time(s), key_sample(pix.kp, s));
}

struct ImageSample { float x, y; Colorf c; };

// ---- the Sampler trait as thread_work uses it (sampler/mod.rs:20-49), over the counter RNG: every number is a function of the frame
// key, the pixel, the get_samples round and the index inside it, so tiles, threads and shards add up to the same film.
struct Region {   // sampler/mod.rs:62-98
    uint32_t cur_x = 0, cur_y = 0, start_x = 0, start_y = 0, end_x = 0, end_y = 0;
    void select_region(uint32_t x0, uint32_t y0) { start_x = cur_x = x0; start_y = cur_y = y0; end_x = x0 + 8; end_y = y0 + 8; }
    void advance() {   // x fastest (ld.rs:47-51, uniform.rs:29-33, adaptive.rs:137-141)
        cur_x += 1;
        if (cur_x == end_x) { cur_x = start_x; cur_y += 1; }
    }
};
inline uint32_t next_power_of_two(uint32_t v) { uint32_t p = 1; while (p < v && p < 0x80000000u) p <<= 1; return p; }   // usize::next_power_of_two (0 -> 1)

// sampler/ld.rs:20-64: all spp samples of a pixel per get_samples call
struct LowDiscrepancySampler {
    uint32_t kf, width, spp;
    Region region;
    PixelSampler pix;
    uint32_t sx = 0, sy = 0;
    LowDiscrepancySampler(uint32_t kf_, uint32_t width_, uint32_t spp_) : kf(kf_), width(width_), spp(spp_) {}
    uint32_t max_spp() const { return spp; }
    uint32_t kind() const { return TRAY_SAMPLER_LOW_DISCREPANCY; }
    uint32_t samples_taken() const { return 0u; }
    void select_block(uint32_t x0, uint32_t y0) { region.select_region(x0, y0); }
    bool has_samples() const { return region.cur_y != region.end_y; }
    void get_samples(std::vector<std::pair<float, float>>& samples) {
        samples.clear();
        if (!has_samples()) return;
        sx = region.cur_x; sy = region.cur_y;
        pix.init(kf, sy * width + sx, spp);
        samples.resize(spp);
        for (uint32_t s = 0; s < spp; ++s) pix.position(s, sx, sy, samples[s].first, samples[s].second);
        region.advance();
    }
    void get_samples_1d(std::vector<float>& t) { for (uint32_t s = 0; s < spp && s < t.size(); ++s) t[s] = pix.time(s); }
    uint32_t path_key(uint32_t k) const { return key_sample(pix.kp, k); }
    uint32_t sampled_x() const { return sx; }
    uint32_t sampled_y() const { return sy; }
    bool report_results(const ImageSample*, size_t) { return true; }   // sampler/mod.rs:48
};

// sampler/uniform.rs:15-55: one sample at the centre of the pixel; every other number an independent uniform draw
struct UniformSampler {
    uint32_t kf, width;
    Region region;
    uint32_t kp = 0, sx = 0, sy = 0;
    UniformSampler(uint32_t kf_, uint32_t width_) : kf(kf_), width(width_) {}
    uint32_t max_spp() const { return 1u; }
    uint32_t kind() const { return TRAY_SAMPLER_UNIFORM; }
    uint32_t samples_taken() const { return 0u; }
    void select_block(uint32_t x0, uint32_t y0) { region.select_region(x0, y0); }
    bool has_samples() const { return region.cur_y != region.end_y; }
    void get_samples(std::vector<std::pair<float, float>>& samples) {
        samples.clear();
        if (!has_samples()) return;
        sx = region.cur_x; sy = region.cur_y;
        kp = key_pixel(kf, sy * width + sx);
        samples.push_back({(float)sx + 0.5f, (float)sy + 0.5f});   // uniform.rs:28
        region.advance();
    }
    void get_samples_1d(std::vector<float>& t) { if (!t.empty()) t[0] = (float)(draw(kp, PD_SCR_T) >> 8) / (float)(1u << 24); }   // uniform.rs:42-46
    uint32_t path_key(uint32_t) const { return key_sample(kp, 0u); }
    uint32_t sampled_x() const { return sx; }
    uint32_t sampled_y() const { return sy; }
    bool report_results(const ImageSample*, size_t) { return true; }
};

// sampler/adaptive.rs:17-143. Round j of a pixel (the j-th get_samples call before report_results lets go of it) has the key
// key_pass(kp, j): its two position scrambles, its time scramble and its two shuffles (Kensler's hashed permutation stands for
// rng.shuffle, as in PixelSampler) derive from it; sample i of the round has the path key key_sample(key_pass, i).
struct AdaptiveSampler {
    uint32_t kf, width;
    Region region;
    uint32_t min_spp, max_spp_, step_size, taken = 0;
    float avg_luminance = 0.0f;
    uint32_t round = 0, count = 0, kq = 0, sx = 0, sy = 0;
    AdaptiveSampler(uint32_t kf_, uint32_t width_, uint32_t min_spp_, uint32_t max_spp__) : kf(kf_), width(width_) {
        min_spp = next_power_of_two(min_spp_);    // adaptive.rs:36-47 (is_power_of_two is false for 0)
        max_spp_ = next_power_of_two(max_spp__);
        step_size = next_power_of_two((max_spp_ - min_spp) / 5u);   // adaptive.rs:48
    }
    uint32_t max_spp() const { return max_spp_; }
    uint32_t kind() const { return TRAY_SAMPLER_ADAPTIVE; }
    uint32_t samples_taken() const { return taken; }
    void select_block(uint32_t x0, uint32_t y0) { region.select_region(x0, y0); }
    bool has_samples() const { return region.cur_y != region.end_y; }
    void get_samples(std::vector<std::pair<float, float>>& samples) {
        samples.clear();
        if (!has_samples()) return;
        sx = region.cur_x; sy = region.cur_y;
        if (taken == 0) { taken += min_spp; count = min_spp; round = 0; }   // adaptive.rs:97-102
        else { taken += step_size; count = step_size; round += 1; }         // adaptive.rs:103-109
        samples.resize(count);
        kq = key_pass(key_pixel(kf, sy * width + sx), round);
        // get_samples_2d (adaptive.rs:112-116): sample_2d(samples, scramble, samples_taken), then rng.shuffle
        const uint32_t scr_x = draw(kq, PD_SCR_X), scr_y = draw(kq, PD_SCR_Y), key_xy = draw(kq, PD_PERM_XY);
        for (uint32_t i = 0; i < count; ++i) {
            const uint32_t n = permute(i, count, key_xy) + taken;
            samples[i].first = van_der_corput(n, scr_x) + (float)sx;      // adaptive.rs:107-110
            samples[i].second = sobol(n, scr_y) + (float)sy;
        }
    }
    // get_samples_1d (adaptive.rs:117-121) on thread_work's max_spp-long time_samples: every entry is filled and shuffled, the zip
    // of multithreaded.rs:94 then uses the first `count`
    void get_samples_1d(std::vector<float>& t) {
        const uint32_t scr_t = draw(kq, PD_SCR_T), key_t = draw(kq, PD_PERM_T);
        for (uint32_t i = 0; i < t.size(); ++i) t[i] = van_der_corput(permute(i, (uint32_t)t.size(), key_t) + taken, scr_t);
    }
    uint32_t path_key(uint32_t i) const { return key_sample(kq, i); }
    uint32_t sampled_x() const { return sx; }
    uint32_t sampled_y() const { return sy; }
    // adaptive.rs:55-75
    bool needs_supersampling(const ImageSample* samples, size_t n) {
        const float max_contrast = 0.5f;
        if (taken == min_spp) {
            float ac = 0.0f;
            for (size_t k = 0; k < n; ++k) ac = ac + samples[k].c.luminance();
            avg_luminance = ac / (float)n;
        } else {
            const size_t prev_samples = n - step_size;
            for (size_t i = prev_samples; i < n; ++i) avg_luminance = (samples[i].c.luminance() + (float)(i - 1) * avg_luminance) / (float)i;
        }
        for (size_t k = 0; k < n; ++k)
            if (std::fabs(samples[k].c.luminance() - avg_luminance) / avg_luminance > max_contrast) return true;
        return false;
    }
    // adaptive.rs:133-143
    bool report_results(const ImageSample* samples, size_t n) {
        if (taken >= max_spp_ || !needs_supersampling(samples, n)) {
            taken = 0;
            region.advance();
            return true;
        }
        return false;
    }
};

// RenderTarget::write (render_target.rs:77-165) into a dense RGBW image (get_renderf32 layout)
// Destination pixel (ix, iy) lives at dst + ((iy - oy) * stride + (ix - ox)) * 4.
void film_write(const TrayFilm& film, const std::vector<ImageSample>& samples, uint32_t rx0, uint32_t ry0, float* rgbw,
                int ox, int oy, int stride) {
    const int W = (int)film.width, H = (int)film.height;
    const int fpw = film.filter_pixel_w, fph = film.filter_pixel_h;
    const int lock = 2;
    int x_range[2] = {std::max((int)rx0 - fpw, 0), std::min((int)rx0 + 8 + fpw, W - 1)};
    int y_range[2] = {std::max((int)ry0 - fph, 0), std::min((int)ry0 + 8 + fph, H - 1)};
    if (x_range[1] - x_range[0] < 0 || y_range[1] - y_range[0] < 0) return;
    int bx_range[2] = {x_range[0] / lock, x_range[1] / lock}, by_range[2] = {y_range[0] / lock, y_range[1] / lock};
    Colorf filtered[4];
    const int N = TRAY_FILTER_TABLE_SIZE;
    for (int by = by_range[0]; by <= by_range[1]; ++by)
        for (int bx = bx_range[0]; bx <= bx_range[1]; ++bx) {
            int bxs = bx * lock, bys = by * lock;
            int xw[2] = {std::max(x_range[0], bxs), std::min(x_range[1] + 1, bxs + lock)};
            int yw[2] = {std::max(y_range[0], bys), std::min(y_range[1] + 1, bys + lock)};
            for (auto& c : filtered) c = Colorf::broadcast(0.0f);
            for (const ImageSample& c : samples) {
                if (!(c.x >= (float)(xw[0] - fpw) && c.x < (float)(xw[1] + fpw) && c.y >= (float)(yw[0] - fph) && c.y < (float)(yw[1] + fph))) continue;
                float img_x = c.x - 0.5f, img_y = c.y - 0.5f;
                for (int iy = yw[0]; iy < yw[1]; ++iy) {
                    float fy = std::fabs((float)iy - img_y) * film.inv_h;
                    if (fy > film.filter_h) continue;
                    int fy_idx = std::min((int)(fy * (float)N), N - 1);
                    for (int ix = xw[0]; ix < xw[1]; ++ix) {
                        float fx = std::fabs((float)ix - img_x) * film.inv_w;
                        if (fx > film.filter_w) continue;
                        int fx_idx = std::min((int)(fx * (float)N), N - 1);
                        float weight = film.table[fy_idx * N + fx_idx];
                        int pxi = (iy - bys) * lock + ix - bxs;
                        filtered[pxi].r += weight * c.c.r;
                        filtered[pxi].g += weight * c.c.g;
                        filtered[pxi].b += weight * c.c.b;
                        filtered[pxi].a += weight;
                    }
                }
            }
            for (int iy = yw[0]; iy < yw[1]; ++iy)
                for (int ix = xw[0]; ix < xw[1]; ++ix) {
                    int pxi = (iy - bys) * lock + ix - bxs;
                    float* dst = rgbw + ((size_t)(iy - oy) * stride + (ix - ox)) * 4;
                    dst[0] += filtered[pxi].r; dst[1] += filtered[pxi].g; dst[2] += filtered[pxi].b; dst[3] += filtered[pxi].a;
                }
        }
}

// sampler/morton.rs + block_queue.rs:28-48 (own copy: the oracle must not depend on the product library)
uint32_t part1_by1(uint32_t x) {
    x &= 0x0000ffffu; x = (x ^ (x << 8)) & 0x00ff00ffu; x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u; return (x ^ (x << 1)) & 0x55555555u;
}
std::vector<std::pair<uint32_t, uint32_t>> block_queue(uint32_t w, uint32_t h) {
    uint32_t nx = w / 8, ny = h / 8;
    std::vector<std::pair<uint32_t, uint32_t>> b((size_t)nx * ny);
    for (uint32_t i = 0; i < nx * ny; ++i) b[i] = {i % nx, i / nx};
    std::stable_sort(b.begin(), b.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& c) {
        return ((part1_by1(a.second) << 1) + part1_by1(a.first)) < ((part1_by1(c.second) << 1) + part1_by1(c.first));
    });
    return b;
}

}  // namespace

extern "C" {

typedef struct OracleStats { uint64_t samples, vertices, rays; double seconds; } OracleStats;

}  // extern "C"

// thread_work (exec/multithreaded.rs:72-114) over tiles [tile_start, tile_start + tile_count) of the Morton queue with n_threads
// workers pulling tiles from an atomic counter (block_queue.rs:52-59), each with a Sampler of its own from `make_sampler`
// (multithreaded.rs:74). Adds into rgbw (width*height*4 f32). Tile results are merged in queue order so the output does not depend
// on the thread count. flags: ORC_FAITHFUL_XF | ORC_BRUTE_FORCE. `stride` > 1 renders only every stride-th tile of the range (bench
// subset). counts (optional, width*height): camera samples taken per pixel.
template <class MakeSampler>
static int render_tiles_with(const TrayFlatScene* fs, uint32_t tile_start, uint32_t tile_count, uint32_t stride, uint64_t seed, float* rgbw,
                             int n_threads, int flags, OracleStats* stats_out, uint32_t* counts, MakeSampler make_sampler) {
    auto queue = block_queue(fs->film.width, fs->film.height);
    if (tile_start > queue.size()) tile_start = (uint32_t)queue.size();
    if (tile_count == 0 || tile_start + (size_t)tile_count > queue.size()) tile_count = (uint32_t)(queue.size() - tile_start);
    if (stride == 0) stride = 1;
    std::vector<uint32_t> tiles;
    for (uint32_t i = 0; i < tile_count; i += stride) tiles.push_back(tile_start + i);
    const uint32_t kf = key_frame(seed, fs->frame);
    const int W = (int)fs->film.width, H = (int)fs->film.height;
    // every tile touches at most a 16x16 pixel window starting at (x0-4, y0-4) (filter radius 4 px)
    const int fpw = fs->film.filter_pixel_w, fph = fs->film.filter_pixel_h;
    const int win_w = 8 + 2 * fpw + 1, win_h = 8 + 2 * fph + 1;
    std::vector<float> windows((size_t)tiles.size() * win_w * win_h * 4, 0.0f);
    std::atomic<size_t> next(0);
    if (n_threads < 1) n_threads = 1;
    std::vector<Stats> tstats(n_threads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        Stats st;
        SceneView sv{fs, flags, &st};
        auto sampler = make_sampler(kf);                                   // multithreaded.rs:74
        std::vector<std::pair<float, float>> sample_pos;
        std::vector<float> time_samples(sampler.max_spp(), 0.0f);          // multithreaded.rs:76
        std::vector<ImageSample> block_samples;
        block_samples.reserve((size_t)sampler.max_spp() * 64);
        for (;;) {
            size_t qi = next.fetch_add(1);
            if (qi >= tiles.size()) break;
            auto tile = queue[tiles[qi]];
            uint32_t x0 = tile.first * 8, y0 = tile.second * 8;
            block_samples.clear();
            sampler.select_block(x0, y0);                                  // multithreaded.rs:88
            size_t pixel_samples = 0;
            while (sampler.has_samples()) {
                sampler.get_samples(sample_pos);                           // a pixel's (next) samples, multithreaded.rs:92
                sampler.get_samples_1d(time_samples);
                sv.smp_kind = sampler.kind(); sv.smp_offset = sampler.samples_taken();
                for (size_t k = 0; k < sample_pos.size() && k < time_samples.size(); ++k) {   // zip, multithreaded.rs:94
                    ImageSample is;
                    is.x = sample_pos[k].first; is.y = sample_pos[k].second;
                    is.c = sample_radiance(sv, is.x, is.y, time_samples[k], sampler.path_key((uint32_t)k));
                    block_samples.push_back(is);
                }
                if (counts) counts[(size_t)sampler.sampled_y() * W + sampler.sampled_x()] += (uint32_t)sample_pos.size();
                if (sampler.report_results(block_samples.data() + pixel_samples, block_samples.size() - pixel_samples))   // multithreaded.rs:107-109
                    pixel_samples = block_samples.size();
            }
            // RenderTarget::write into this tile's private window
            float* win = windows.data() + qi * (size_t)win_w * win_h * 4;
            film_write(fs->film, block_samples, x0, y0, win, (int)x0 - fpw, (int)y0 - fph, win_w);
        }
        tstats[tid] = st;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    auto t1 = std::chrono::steady_clock::now();
    // merge in queue order (film::Image::add_pixels semantics)
    for (size_t qi = 0; qi < tiles.size(); ++qi) {
        auto tile = queue[tiles[qi]];
        int wx0 = (int)tile.first * 8 - fpw, wy0 = (int)tile.second * 8 - fph;
        const float* win = windows.data() + qi * (size_t)win_w * win_h * 4;
        for (int r = 0; r < win_h; ++r) {
            int iy = wy0 + r;
            if (iy < 0 || iy >= H) continue;
            for (int c = 0; c < win_w; ++c) {
                int ix = wx0 + c;
                if (ix < 0 || ix >= W) continue;
                float* dst = rgbw + ((size_t)iy * W + ix) * 4;
                const float* src = win + ((size_t)r * win_w + c) * 4;
                dst[0] += src[0]; dst[1] += src[1]; dst[2] += src[2]; dst[3] += src[3];
            }
        }
    }
    if (stats_out) {
        Stats sum;
        for (auto& s : tstats) { sum.samples += s.samples; sum.vertices += s.vertices; sum.rays += s.rays; }
        stats_out->samples = sum.samples; stats_out->vertices = sum.vertices; stats_out->rays = sum.rays;
        stats_out->seconds = std::chrono::duration<double>(t1 - t0).count();
    }
    return 0;
}

extern "C" {

int oracle_render_tiles(const TrayFlatScene* fs, uint32_t tile_start, uint32_t tile_count, uint32_t stride, uint32_t spp,
                        uint64_t seed, float* rgbw, int n_threads, int flags, OracleStats* stats_out) {
    if (!fs || !rgbw || fs->n_lights == 0 || (spp & (spp - 1)) != 0 || spp == 0) return -1;
    const uint32_t width = fs->film.width;
    return render_tiles_with(fs, tile_start, tile_count, stride, seed, rgbw, n_threads, flags, stats_out, nullptr,
                             [=](uint32_t kf) { return LowDiscrepancySampler(kf, width, spp); });
}

// The same with sampler::Uniform::new(dim) (kind TRAY_SAMPLER_UNIFORM) or sampler::Adaptive::new(dim, min_spp, max_spp)
// (TRAY_SAMPLER_ADAPTIVE) in place of LowDiscrepancy -- the counterpart of tray_scene_set_sampler. counts: see render_tiles_with.
int oracle_render_tiles_sampler(const TrayFlatScene* fs, uint32_t tile_start, uint32_t tile_count, uint32_t stride, uint32_t kind,
                                uint32_t min_spp, uint32_t max_spp, uint64_t seed, float* rgbw, int n_threads, int flags,
                                OracleStats* stats_out, uint32_t* counts) {
    if (!fs || !rgbw || fs->n_lights == 0) return -1;
    const uint32_t width = fs->film.width;
    if (kind == TRAY_SAMPLER_UNIFORM)
        return render_tiles_with(fs, tile_start, tile_count, stride, seed, rgbw, n_threads, flags, stats_out, counts,
                                 [=](uint32_t kf) { return UniformSampler(kf, width); });
    if (kind == TRAY_SAMPLER_ADAPTIVE) {
        if (next_power_of_two(max_spp) < next_power_of_two(min_spp)) return -1;   // (Adaptive::new would underflow, adaptive.rs:48)
        return render_tiles_with(fs, tile_start, tile_count, stride, seed, rgbw, n_threads, flags, stats_out, counts,
                                 [=](uint32_t kf) { return AdaptiveSampler(kf, width, min_spp, max_spp); });
    }
    return -1;
}
// How many samples Adaptive takes for ONE pixel whose successive samples have the grey colours lum[0], lum[1], ... (n of them available;
// returns 0 if it would need more): get_samples / report_results (adaptive.rs:92-110, 133-143) driven as thread_work drives them.
uint32_t oracle_adaptive_samples_for(const float* lum, uint32_t n, uint32_t min_spp, uint32_t max_spp) {
    AdaptiveSampler a(0u, 8u, min_spp, max_spp);
    a.select_block(0u, 0u);
    std::vector<std::pair<float, float>> pos;
    std::vector<ImageSample> got;
    for (;;) {
        a.get_samples(pos);
        if (got.size() + pos.size() > n) return 0u;
        for (size_t k = 0; k < pos.size(); ++k) {
            ImageSample is;
            is.x = pos[k].first; is.y = pos[k].second;
            const float l = lum[got.size()];
            is.c = Colorf(l, l, l);
            got.push_back(is);
        }
        if (a.report_results(got.data(), got.size())) return (uint32_t)got.size();
    }
}
// Adaptive::new's rounding and step (adaptive.rs:36-48): out = {min_spp, max_spp, step_size}
void oracle_adaptive_params(uint32_t min_spp, uint32_t max_spp, uint32_t* out3) {
    AdaptiveSampler a(0u, 8u, min_spp, max_spp);
    out3[0] = a.min_spp; out3[1] = a.max_spp_; out3[2] = a.step_size;
}

int oracle_intersect(const TrayFlatScene* fs, uint32_t n, const TrayRay* rays, TrayHit* hits, int flags) {
    if (!fs || !rays || !hits) return -1;
    SceneView sv{fs, flags, nullptr};
    for (uint32_t i = 0; i < n; ++i) {
        Ray r;
        r.o = Vec3(rays[i].o[0], rays[i].o[1], rays[i].o[2]);
        r.d = Vec3(rays[i].d[0], rays[i].d[1], rays[i].d[2]);
        r.min_t = rays[i].min_t; r.max_t = rays[i].max_t; r.time = rays[i].time;
        Hit h;
        TrayHit& o = hits[i];
        std::memset(&o, 0, sizeof o);
        if (scene_intersect(sv, r, h)) {
            o.t = r.max_t; o.inst = h.inst; o.prim = h.prim;
            for (int k = 0; k < 3; ++k) { o.p[k] = h.p[k]; o.n[k] = h.n[k]; o.ng[k] = h.ng[k]; o.dp_du[k] = h.dp_du[k]; o.dp_dv[k] = h.dp_dv[k]; }
            o.u = h.u; o.v = h.v;
        } else {
            o.t = r.max_t; o.inst = 0xffffffffu;
        }
    }
    return 0;
}

// Schedule analysis: for n camera samples, the per-vertex query codes (see VertexLog) of each path, up to 16 vertices per sample;
// counts[i] = number of vertices (0 = the camera ray missed)
int oracle_path_profile(const TrayFlatScene* fs, uint32_t n, const uint32_t* px, const uint32_t* py, const uint32_t* si, uint32_t spp, uint64_t seed,
                        uint8_t* codes, uint8_t* counts) {
    if (!fs || !px || !py || !si || !codes || !counts) return -1;
    SceneView sv{fs, 0, nullptr};
    const uint32_t kf = key_frame(seed, fs->frame);
    std::vector<uint8_t> log;
    g_vlog.sink = &log;
    for (uint32_t i = 0; i < n; ++i) {
        log.clear();
        float sx, sy;
        trace_sample(sv, kf, px[i], py[i], si[i], spp, sx, sy);
        const size_t m = std::min<size_t>(log.size(), 16);
        counts[i] = (uint8_t)m;
        std::memcpy(codes + (size_t)i * 16, log.data(), m);
    }
    g_vlog.sink = nullptr;
    return 0;
}

int oracle_camera_rays(const TrayFlatScene* fs, uint32_t n, const float* xy, const float* time, TrayRay* rays) {
    if (!fs || !xy || !rays) return -1;
    SceneView sv{fs, 0, nullptr};
    for (uint32_t i = 0; i < n; ++i) {
        Ray r = camera_generate_ray(sv, xy[2 * i], xy[2 * i + 1], time ? time[i] : 0.0f);
        for (int k = 0; k < 3; ++k) { rays[i].o[k] = r.o[k]; rays[i].d[k] = r.d[k]; }
        rays[i].min_t = r.min_t; rays[i].max_t = r.max_t; rays[i].time = r.time;
    }
    return 0;
}

// Same record layout as tray_debug_sample_radiance
int oracle_sample_radiance(const TrayFlatScene* fs, uint32_t n, const uint32_t* px, const uint32_t* py, const uint32_t* si,
                           uint32_t spp, uint64_t seed, float* out, int flags) {
    if (!fs || !px || !py || !si || !out || fs->n_lights == 0) return -1;
    uint32_t kf = key_frame(seed, fs->frame);
    for (uint32_t i = 0; i < n; ++i) {
        Stats st;
        SceneView sv{fs, flags, &st};
        float sx, sy;
        Colorf c = trace_sample(sv, kf, px[i], py[i], si[i], spp, sx, sy);
        float* o = out + (size_t)i * 8;
        o[0] = c.r; o[1] = c.g; o[2] = c.b; o[3] = sx; o[4] = sy; o[5] = (float)st.vertices; o[6] = (float)st.rays; o[7] = 0.0f;
    }
    return 0;
}

// Same contract as tray_debug_bsdf: canonical frame n = +z, dp_du = +x
int oracle_bsdf(const TrayFlatScene* fs, uint32_t material_id, uint32_t flags_sel, uint32_t n, const float* dirs, const float* u3, float* out) {
    if (!fs || material_id >= fs->n_materials || !dirs || !u3 || !out) return -1;
    // find (or fake) an instance carrying this material
    TrayFlatScene tmp = *fs;
    TrayInstance fake{};
    fake.material_id = material_id;
    tmp.instances = &fake; tmp.n_instances = 1;
    Hit h;
    h.p = Vec3(0, 0, 0); h.n = Vec3(0, 0, 1); h.ng = Vec3(0, 0, 1); h.dp_du = Vec3(1, 0, 0); h.dp_dv = Vec3(0, 1, 0); h.inst = 0;
    BSDF b = material_bsdf(tmp, h);
    int flags = flags_sel == 0 ? BX_ALL : BX_NON_SPECULAR;
    for (uint32_t i = 0; i < n; ++i) {
        Vec3 wo(dirs[6 * i], dirs[6 * i + 1], dirs[6 * i + 2]), wi(dirs[6 * i + 3], dirs[6 * i + 4], dirs[6 * i + 5]);
        float* o = out + (size_t)i * 12;
        Colorf e = b.eval(wo, wi, flags);
        o[0] = e.r; o[1] = e.g; o[2] = e.b; o[3] = b.pdf(wo, wi, flags);
        Vec3 swi;
        float spdf;
        int st;
        Colorf f = b.sample(wo, flags, u3[3 * i], u3[3 * i + 1], u3[3 * i + 2], swi, spdf, st);
        o[4] = f.r; o[5] = f.g; o[6] = f.b; o[7] = swi.x; o[8] = swi.y; o[9] = swi.z; o[10] = spdf; o[11] = (float)st;
    }
    return 0;
}

// Rebuilds every instance's Transform from its TRS stack the way receiver.rs:30 does per ray and
// writes mat(16)+inv(16) per instance: cross-check against the loader's matrices.
int oracle_instance_matrices(const TrayFlatScene* fs, float time, float* out) {
    if (!fs || !out) return -1;
    SceneView sv{fs, ORC_FAITHFUL_XF, nullptr};
    for (uint32_t i = 0; i < fs->n_instances; ++i) {
        Transform t = sv.instance_transform(i, time);
        std::memcpy(out + (size_t)i * 32, t.mat.m, sizeof t.mat.m);
        std::memcpy(out + (size_t)i * 32 + 16, t.inv.m, sizeof t.inv.m);
    }
    return 0;
}

// B-spline of TRS keyframes: ctrl = n x 10 floats (translation 3, rotation xyzw 4, scaling 3) -> out 10 floats
int oracle_bspline_point(const float* ctrl, uint32_t n, const float* knots, uint32_t n_knots, uint32_t degree, float t, float* out) {
    if (!ctrl || !knots || !out || n == 0 || n > 64 || degree > 7 || n_knots != n + degree + 1) return -1;
    Key keys[64];
    for (uint32_t i = 0; i < n; ++i) {
        const float* c = ctrl + (size_t)i * 10;
        keys[i].t = Vec3(c[0], c[1], c[2]);
        for (int k = 0; k < 4; ++k) keys[i].q[k] = c[3 + k];
        keys[i].s = Vec3(c[7], c[8], c[9]);
    }
    Key k = n == 1 ? keys[0] : de_boor(keys, knots, n_knots, degree, clampf(t, knots[degree], knots[n_knots - 1 - degree]), key_interpolate);
    out[0] = k.t.x; out[1] = k.t.y; out[2] = k.t.z;
    for (int i = 0; i < 4; ++i) out[3 + i] = k.q[i];
    out[7] = k.s.x; out[8] = k.s.y; out[9] = k.s.z;
    return 0;
}

// AnimatedTransform::transform(time) of any spline stack of the flat scene (camera or instance): mat(16) + inv(16)
int oracle_stack_transform(const TrayFlatScene* fs, uint32_t xf_first, uint32_t xf_count, float time, float* out) {
    if (!fs || !out || xf_first + xf_count > fs->n_xf_levels) return -1;
    SceneView sv{fs, ORC_FAITHFUL_XF, nullptr};
    Transform t = sv.stack_transform(xf_first, xf_count, time);
    std::memcpy(out, t.mat.m, sizeof t.mat.m);
    std::memcpy(out + 16, t.inv.m, sizeof t.inv.m);
    return 0;
}

// ---- known-answer hooks for the reference's own unit tests (linalg) and the sampler
void oracle_mat4_mul(const float* a, const float* b, float* out) { Mat4 r = Mat4::from(a) * Mat4::from(b); std::memcpy(out, r.m, sizeof r.m); }
void oracle_mat4_add(const float* a, const float* b, float* out) { Mat4 r = Mat4::from(a) + Mat4::from(b); std::memcpy(out, r.m, sizeof r.m); }
void oracle_mat4_sub(const float* a, const float* b, float* out) { Mat4 r = Mat4::from(a) - Mat4::from(b); std::memcpy(out, r.m, sizeof r.m); }
void oracle_mat4_inverse(const float* a, float* out) { Mat4 r = Mat4::from(a).inverse(); std::memcpy(out, r.m, sizeof r.m); }
// kind: 0 identity, 1 translate(v), 2 scale(v), 3 rotate_x(a), 4 rotate_y(a), 5 rotate_z(a), 6 rotate(axis v, a)
void oracle_transform(int kind, const float* v, float angle, float* mat, float* inv) {
    Transform t = Transform::identity();
    Vec3 vv = v ? Vec3(v[0], v[1], v[2]) : Vec3();
    switch (kind) {
        case 1: t = Transform::translate(vv); break;
        case 2: t = Transform::scale(vv); break;
        case 3: t = Transform::rotate_x(angle); break;
        case 4: t = Transform::rotate_y(angle); break;
        case 5: t = Transform::rotate_z(angle); break;
        case 6: t = Transform::rotate(vv, angle); break;
        default: break;
    }
    std::memcpy(mat, t.mat.m, sizeof t.mat.m);
    std::memcpy(inv, t.inv.m, sizeof t.inv.m);
}
// what: 0 point, 1 vector, 2 normal (uses inv), 3 inverse-point, 4 inverse-vector
void oracle_transform_apply(const float* mat, const float* inv, int what, const float* v, float* out) {
    Transform t = Transform::from_pair(Mat4::from(mat), Mat4::from(inv));
    Vec3 a(v[0], v[1], v[2]), r;
    switch (what) {
        case 0: r = t.point(a); break;
        case 1: r = t.vector(a); break;
        case 2: r = t.normal(a); break;
        case 3: r = t.inv_point(a); break;
        default: r = t.inv_vector(a); break;
    }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_cross(const float* a, const float* b, float* out) { Vec3 r = cross(Vec3(a[0], a[1], a[2]), Vec3(b[0], b[1], b[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
float oracle_dot(const float* a, const float* b) { return dot(Vec3(a[0], a[1], a[2]), Vec3(b[0], b[1], b[2])); }
float oracle_van_der_corput(uint32_t n, uint32_t scramble) { return van_der_corput(n, scramble); }
float oracle_sobol(uint32_t n, uint32_t scramble) { return sobol(n, scramble); }
uint32_t oracle_permute(uint32_t i, uint32_t l, uint32_t p) { return permute(i, l, p); }
void oracle_shuffle_small(uint32_t key, uint32_t n, uint8_t* perm) { uint8_t p[16]; shuffle_small(key, n, p); std::memcpy(perm, p, n); }
uint32_t oracle_mix32(uint32_t x) { return mix32(x); }
// pixel sample position + time for (pixel, s)
void oracle_pixel_sample(uint64_t seed, uint32_t frame, uint32_t width, uint32_t px, uint32_t py, uint32_t s, uint32_t spp, float* out3) {
    PixelSampler pix;
    pix.init(key_frame(seed, frame), py * width + px, spp);
    pix.position(s, px, py, out3[0], out3[1]);
    out3[2] = pix.time(s);
}
// the six per-bounce LD values of a camera sample: out[b*9 + {l2x,l2y,b2x,b2y,p2x,p2y,l1,b1,p1}], rr[b]
void oracle_path_samples_mode(uint64_t seed, uint32_t frame, uint32_t width, uint32_t px, uint32_t py, uint32_t s, uint32_t n, int fresh, float* out, float* rr);
void oracle_path_samples(uint64_t seed, uint32_t frame, uint32_t width, uint32_t px, uint32_t py, uint32_t s, uint32_t n, float* out, float* rr) {
    oracle_path_samples_mode(seed, frame, width, px, py, s, n, 0, out, rr);
}
// ... fresh != 0: with per-array Fisher-Yates shuffles (ORC_FRESH_SHUFFLES) instead of the permutation pool
void oracle_path_samples_mode(uint64_t seed, uint32_t frame, uint32_t width, uint32_t px, uint32_t py, uint32_t s, uint32_t n, int fresh, float* out, float* rr) {
    PathSamples ps;
    ps.init(key_sample(key_pixel(key_frame(seed, frame), py * width + px), s), n, fresh != 0);
    for (uint32_t b = 0; b < n; ++b) {
        float* o = out + 9 * b;
        ps.two_d(0, b, o[0], o[1]); ps.two_d(1, b, o[2], o[3]); ps.two_d(2, b, o[4], o[5]);
        o[6] = ps.one_d(0, b); o[7] = ps.one_d(1, b); o[8] = ps.one_d(2, b);
        rr[b] = ps.rr(b);
    }
}
// Mitchell-Netravali / film: splat one sample (x, y, rgb) of tile (tx, ty) into rgbw
// Texture::sample_color / sample_f32 of texture `tex` at n (u, v, time) triples -> n x (r, g, b, a, f32)
int oracle_texture_sample(const TrayFlatScene* fs, uint32_t tex, uint32_t n, const float* uvt, float* out) {
    if (!fs || tex >= fs->n_textures) return 1;
    for (uint32_t i = 0; i < n; ++i) {
        const Colorf c = texture_sample_color(*fs, tex, uvt[3 * i], uvt[3 * i + 1], uvt[3 * i + 2]);
        out[5 * i] = c.r; out[5 * i + 1] = c.g; out[5 * i + 2] = c.b; out[5 * i + 3] = c.a;
        out[5 * i + 4] = texture_sample_f32(*fs, tex, uvt[3 * i], uvt[3 * i + 1], uvt[3 * i + 2]);
    }
    return 0;
}

void oracle_film_write(const TrayFlatScene* fs, uint32_t n, const float* samples5, uint32_t tile_x, uint32_t tile_y, float* rgbw) {
    std::vector<ImageSample> v(n);
    for (uint32_t i = 0; i < n; ++i) { v[i].x = samples5[5 * i]; v[i].y = samples5[5 * i + 1]; v[i].c = Colorf(samples5[5 * i + 2], samples5[5 * i + 3], samples5[5 * i + 4]); }
    film_write(fs->film, v, tile_x * 8, tile_y * 8, rgbw, 0, 0, (int)fs->film.width);
}

}  // extern "C"
