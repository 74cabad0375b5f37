// Device-side camera, per-sample LD sampler and the Path integrator, one path vertex per wave step.
//   Camera::generate_ray            film/camera.rs:150-157
//   LowDiscrepancy::get_samples*    sampler/ld.rs:33-64 (draws replaced by TRAY-CBRNG)
//   Path::illumination              integrator/path.rs:45-120
//   sample_one_light/estimate_direct integrator/mod.rs:106-169
//   Light for Emitter, OcclusionTester  geometry/emitter.rs:164-203, light/mod.rs:14-39
//
// Wave-synchronous schedule. Every live lane of a wave advances by exactly one path vertex per step,
// and the step is cut into three stages that all lanes enter together:
//   stage A  trace the camera / continuation ray           -> vertex_begin  (path.rs:69-79, mod.rs:106-127)
//   stage B  trace the light sample's occlusion ray        -> BSDF queries  (mod.rs:127-153, path.rs:84-110)
//   stage C  trace the BSDF-sampled ray of estimate_direct -> vertex_end    (mod.rs:154-166, path.rs:82)
// so the divergent parts of the reference's control flow (which lobe, which light geometry, whether a
// ray exists at all) never put lanes into different stages, and the one traversal code site is entered
// by all lanes that have a ray of that kind. The only re-ordering w.r.t. the reference: the path
// continuation is sampled (stage B) before the BSDF-sampled light ray is traced (stage C). Both only
// read the vertex; `illum += throughput * direct` still happens after the full estimate_direct value is
// known and with the pre-update throughput, so every float operation and its operands are unchanged.
#pragma once
#include "dev_tex.h"

namespace tr {

template <int ANIM>
TR_DEV Ray camera_ray(const DevScene& sc, float px, float py, float time) {
    const TrayCamera& c = *sc.camera_p;
    f3 q = xf_point(c.raster_to_cam, mk(px, py, 0.0f));
    f3 px_pos = mk(c.scaling[0], c.scaling[1], c.scaling[2]) * q;
    f3 d = normalized(px_pos);
    // with a closed shutter (or nothing moving) frame_time is the same for every ray and only selects transforms
    // the host evaluated already
    const float frame_time = (c.shutter_close - c.shutter_open) * time + c.shutter_open;
    Ray r;
    if (ANIM && c.animated) {   // cam_world.transform(frame_time) * Ray (camera.rs:156)
        float x[TR_XF_WORDS];
        if (ANIM == 1 && sc.xf_tab) {   // the frame's table holds the camera's transform at this time index as the last record of the index (dev_geom.h)
            const float4* __restrict__ rec = reinterpret_cast<const float4*>(sc.xf_tab + ((size_t)xf_time_index(time) * sc.xf_tab_stride + sc.n_moving) * TR_XF_REC);
#pragma unroll
            for (int q = 0; q < TR_XF_WORDS / 4; ++q) { const float4 v = rec[q]; x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
        } else
        eval_xform_stack(sc.xf_levels, sc.keyframes, sc.knots, c.xf_first, c.xf_count, frame_time, x);
        r.o = xf_point_affine_w(x, x[25], mk(0.0f, 0.0f, 0.0f));
        r.d = xf_vector(x, d);
    } else {
        r.o = xf_point(c.cam_world, mk(0.0f, 0.0f, 0.0f));
        r.d = xf_vector(c.cam_world, d);
    }
    r.min_t = 0.0f; r.max_t = TR_INF;
    r.time = frame_time; r.col = 0u;
    return r;
}

// Per-pixel part of the sampler: sub-pixel position and shutter time of sample s
TR_DEV void pixel_sample(uint32_t kp, uint32_t s, uint32_t spp, uint32_t px, uint32_t py, float& x, float& y, float& t) {
    uint32_t idx = permute(s, spp, draw(kp, PD_PERM_XY));
    x = van_der_corput(idx, draw(kp, PD_SCR_X)) + (float)px;   // ld.rs:43-46
    y = sobol(idx, draw(kp, PD_SCR_Y)) + (float)py;
    t = van_der_corput(permute(s, spp, draw(kp, PD_PERM_T)), draw(kp, PD_SCR_T));
}

TR_DEV float rr_draw(uint32_t ks, uint32_t bounce) { return (float)(draw(ks, SD_RR + bounce) >> 8) / 16777216.0f; }   // Rng::next_f32

// self.emission.color(time): the host stores color(shutter_open); keys are only present when the colour moves while the
// shutter is open
template <int ANIM>
TR_DEV f3 inst_emission(const DevScene& sc, const TrayInstance* __restrict__ in, float time) {
    if (ANIM && in->emis_count >= 2u) return color_keys_at(sc.color_keys + in->emis_first, in->emis_count, time);
    return mk(in->emission[0], in->emission[1], in->emission[2]);
}
// Emitter::radiance (emitter.rs:140-142)
template <int ANIM>
TR_DEV f3 emitter_radiance(const DevScene& sc, const TrayInstance* __restrict__ in, f3 w, f3 n, float time) {
    return dot(w, n) > 0.0f ? inst_emission<ANIM>(sc, in, time) : mk(0.0f, 0.0f, 0.0f);
}

enum : uint32_t {
    LF_ALIVE = 1u,       // the lane owns an unfinished camera sample
    LF_SPECULAR = 2u,    // previous bounce sampled a specular lobe
    LF_SHADOW = 4u,      // stage B has an occlusion ray to trace
    LF_MIS = 8u,         // stage C has a BSDF-sampled light ray to trace
    LF_LAST = 16u,       // the path ends at this vertex (decided in stage B, applied in stage C)
    LF_MIS_MISS = 1024u,     // the BSDF-sampled light ray cannot hit the light's own primitive: it counts as a ray, it is not traced
    LF_MIS_UNTESTED = 2048u  // LF_MIS set for a specular sample: no Light::pdf looked at the light's primitive (mis_ray_filter)
};
enum : uint32_t { WANT_NONE = 0, WANT_LIGHT = 1, WANT_MIS = 2, WANT_PATH = 3 };

// Everything a lane keeps between stages
struct Lane {
    uint32_t flags;
    uint32_t bounce;       // index of the path vertex being shaded
    uint32_t ks;           // sample key: scrambles and shuffle entries of the six LD arrays derive from it
    float time;            // ray.time of the camera ray, inherited by every ray of the path (path.rs:110, mod.rs:154)
    uint32_t col;          // the path's column of the transform cache (ANIM): thread id (tile kernel) or pool slot (wavefront)
    f3 d;                  // direction of the stage A ray; its origin is bsdf.p: the previous vertex, or the camera position parked there
                           // until the first hit (three registers less across the whole step; bit-identical on the GPU since the
                           // device code is built without the SLP vectoriser, Makefile)
#define LN_O(ln) ((ln).bsdf.p)
    f3 throughput, illum;
    f3 first_ng;           // hit.dg.ng of the camera ray's hit (quirk Q1)
    Bsdf bsdf;             // shading context of the current vertex (BSDF::new)
    uint32_t light_inst;
    f3 li;                 // light sample (stage B)           | stage C: li = (|cos|, mis weight, pdf_bsdf)
    union { f3 wi_l; f3 mis_f; };   // wi_l (written in vertex_begin, last read by the LIGHT query) and mis_f (f of the BSDF half, written by the
                                    // MIS query, which follows it) never live at the same time
    float pdf_l;
    f3 aux_d;              // stage B: occlusion segment p_w - p | stage C: BSDF-sampled direction
    f3 direct;             // direct_light of estimate_direct
    f3 t_vertex;           // throughput at this vertex, kept for `illum += throughput * direct`
    LdsB perm_lds = nullptr;   // the scene's permutation pool in LDS (tile kernel; wave-uniform, costs no register); null: read sc.perm_pool
    // which Sampler fills the path's arrays (sampler/mod.rs:20-49; see lane_2d) and, for Adaptive, its samples_taken when the path was
    // started (adaptive.rs:115,120). The tile and wavefront kernels never write these: the constants fold and the LowDiscrepancy code is
    // all that is compiled into them; k_sampler_pass sets them from its launch parameters.
    uint32_t smp_kind = TRAY_SAMPLER_LOW_DISCREPANCY, smp_offset = 0u;
#ifdef TR_STAGE_CLOCKS   // instrumented builds only: wave clocks of the parts of a BSDF query (set by k_path_tiles, null elsewhere)
    unsigned long long* qclk = nullptr;   // [0] sample head / light setup, [1] eval + pdf site, [2] epilogue of the query kind
    long long qt = 0;
#endif
};
#ifdef TR_STAGE_CLOCKS
#define TR_QCLK_START(ln) do { if ((ln).qclk) (ln).qt = clock64(); } while (0)
#define TR_QCLK(ln, k) do { if ((ln).qclk) { const long long n_ = clock64(); (ln).qclk[k] += (unsigned long long)(n_ - (ln).qt); (ln).qt = n_; } } while (0)
#else
#define TR_QCLK_START(ln) ((void)0)
#define TR_QCLK(ln, k) ((void)0)
#endif

// (the ANIM tile kernel calls xf_cache_fill(sc, cam_ray.time) right after this)
TR_DEV void lane_start_sample(Lane& ln, const Ray& cam_ray, uint32_t ks) {
    ln.flags = LF_ALIVE;
    ln.bounce = 0u;
    ln.ks = ks;
    LN_O(ln) = cam_ray.o; ln.d = cam_ray.d;
    ln.time = cam_ray.time; ln.col = 0u;
    ln.throughput = mk(1.0f, 1.0f, 1.0f);
    ln.illum = mk(0.0f, 0.0f, 0.0f);
}

// sample_02 / van_der_corput of one LD array at the current bounce (ld.rs:54-64, 91-93): the array's scramble word(s), its shuffle
// out of the scene's permutation pool (dev_math.h: TRAY-CBRNG v2), the (0,2)-sequence point of the shuffled index
// TRAY-CBRNG v3 (round 4): a path's shuffle of an array is the COMPOSITION of two pool permutations, idx = perm_q2[perm_q1[bounce]] -- q1 the
// low byte of the array's first scramble word, q2 the low byte of its second one (2-D arrays) or bits 8..15 of its only one (1-D arrays: the
// bits every point of the array shares) -- 65 536 shuffles per array instead of 256. With 256 the joint distribution of an array's points at
// two bounces was a sample of 256 draws over the 72 ordered pairs (n = 9): the covariance of those two values sat up to 4e-3 (12 sigma of
// 60 000 paths) off the value per-array Fisher-Yates shuffles (ld.rs:58,63) give; composed, it is inside the sampling error
// (tests/test_sampler_pool.py). Cost: a second byte load and a 4-bit reversal (the first entry's high nibble is the bit-reversed index).
TR_DEV uint32_t lane_perm_entry(const DevScene& sc, const Lane& ln, uint32_t s1, uint32_t s2) {
    const uint32_t off1 = ((s1 & (TR_PERM_POOL - 1u)) << 4) + ln.bounce;
    const uint32_t e1 = ln.perm_lds ? (uint32_t)ln.perm_lds[off1] : (uint32_t)sc.perm_pool[off1];
    const uint32_t off2 = ((s2 & (TR_PERM_POOL - 1u)) << 4) + (__brev(e1 >> 4) >> 28);
    return ln.perm_lds ? (uint32_t)ln.perm_lds[off2] : (uint32_t)sc.perm_pool[off2];
}
// Uniform (sampler/uniform.rs:36-46): every entry of every array is Range::new(0.0, 1.0).ind_sample(rng) -- next_f32 of a draw of its own,
// counter 64 + 32 * dim + 2 * bounce (+ 1 for the second coordinate); no scrambles, no shuffles.
TR_DEV float lane_uniform(const Lane& ln, uint32_t dim, uint32_t c) {
    return (float)(draw(ln.ks, 64u + 32u * dim + 2u * ln.bounce + c) >> 8) / 16777216.0f;
}
TR_DEV void lane_2d(const DevScene& sc, const Lane& ln, uint32_t dim, float& u0, float& u1) {
    if (ln.smp_kind == TRAY_SAMPLER_UNIFORM) { u0 = lane_uniform(ln, dim, 0u); u1 = lane_uniform(ln, dim, 1u); return; }
    const uint32_t sx = draw(ln.ks, dim), sy = draw(ln.ks, dim + 1u);
    const uint32_t e = lane_perm_entry(sc, ln, sx, sy);
    if (ln.smp_kind == TRAY_SAMPLER_ADAPTIVE) {   // ld::sample_2d(samples, scramble, self.samples_taken) (adaptive.rs:112-116): the points idx + offset
        const uint32_t idx = (__brev(e >> 4) >> 28) + ln.smp_offset;
        u0 = van_der_corput(idx, sx); u1 = sobol(idx, sy);
        return;
    }
    u0 = u24_to_unit(((e & 0xf0u) << 24) ^ sx);   // van_der_corput(idx, sx)
    u1 = u24_to_unit((e << 28) ^ sy);             // sobol(idx, sy)
}
TR_DEV float lane_1d(const DevScene& sc, const Lane& ln, uint32_t dim) {
    if (ln.smp_kind == TRAY_SAMPLER_UNIFORM) return lane_uniform(ln, dim, 0u);
    const uint32_t s = draw(ln.ks, dim);
    const uint32_t e = lane_perm_entry(sc, ln, s, s >> 8);
    if (ln.smp_kind == TRAY_SAMPLER_ADAPTIVE) return van_der_corput((__brev(e >> 4) >> 28) + ln.smp_offset, s);   // ld::sample_1d(.., self.samples_taken) (adaptive.rs:117-121)
    return u24_to_unit(((e & 0xf0u) << 24) ^ s);
}

TR_DEV Ray stage_a_ray(const Lane& ln) {
    Ray r;
    r.o = LN_O(ln); r.d = ln.d;
    r.min_t = ln.bounce == 0u ? 0.0f : 0.001f;   // camera ray (ray.rs:25-27) vs ray.min_t = 0.001 (path.rs:110)
    r.max_t = TR_INF;
    r.time = ln.time; r.col = ln.col;
    return r;
}
TR_DEV Ray stage_b_ray(const Lane& ln) {   // OcclusionTester::test_points (light/mod.rs:21-23): unnormalised segment (quirk Q4)
    Ray r;
    r.o = ln.bsdf.p; r.d = ln.aux_d; r.min_t = 0.001f; r.max_t = 0.999f;
    r.time = ln.time; r.col = ln.col;
    return r;
}
TR_DEV Ray stage_c_ray(const Lane& ln) {   // Ray::segment(p, w_i, 0.001, inf) (mod.rs:154)
    Ray r;
    r.o = ln.bsdf.p; r.d = ln.aux_d; r.min_t = 0.001f; r.max_t = TR_INF;
    r.time = ln.time; r.col = ln.col;
    return r;
}

// Stage A, after the ray hit: head of the loop body of Path::illumination (path.rs:69-79) and the light
// sample of estimate_direct (mod.rs:106-127). Sets LF_SHADOW when an occlusion ray has to be traced.
template <int ANIM>
TR_DEV void vertex_begin(const DevScene& sc, Lane& ln, const HitRec& rec, Counters& cnt) {
    cnt.vertices++;
    const Ray ray = stage_a_ray(ln);
    Hit hit = finish_hit<ANIM>(sc, ray, rec);
    if (ln.bounce == 0u) ln.first_ng = hit.ng;
    const TrayInstance* __restrict__ inst = sc.instances + hit.inst;
    if (ln.bounce == 0u || (ln.flags & LF_SPECULAR)) {
        if (inst->kind != TRAY_INST_RECEIVER) {
            f3 w = -ln.d;
            ln.illum = ln.illum + ln.throughput * emitter_radiance<ANIM>(sc, inst, w, ln.first_ng, ln.time);
        }
    }
    ln.bsdf = make_bsdf(sc, hit);
    if (sc.integrator == TRAY_INTEGRATOR_NORMALS_DEBUG) {   // NormalsDebug::illumination (integrator/normals_debug.rs:28-33): (bsdf.n + 1) / 2, the sample is done
        ln.illum = (ln.bsdf.n + mk(1.0f, 1.0f, 1.0f)) / 2.0f;
        ln.flags &= ~(LF_ALIVE | LF_SHADOW | LF_MIS | LF_MIS_MISS | LF_MIS_UNTESTED);
        return;
    }
    ln.direct = mk(0.0f, 0.0f, 0.0f);
    ln.t_vertex = ln.throughput;
    ln.flags &= ~(LF_SHADOW | LF_MIS | LF_LAST | LF_MIS_MISS | LF_MIS_UNTESTED);
    // sample_one_light (mod.rs:106-111), no 1/p_select (quirk Q6)
    float l1 = lane_1d(sc, ln, SD_L1);
    float fl = l1 * (float)sc.n_lights;
    uint32_t li_idx = fl > 0.0f ? (uint32_t)fl : 0u;
    if (li_idx > sc.n_lights - 1u) li_idx = sc.n_lights - 1u;
    ln.light_inst = sc.lights[li_idx];
    const TrayInstance* __restrict__ light = sc.instances + ln.light_inst;
    // Light::sample_incident (emitter.rs:165-186)
    f3 p_w;
    float x[TR_XF_WORDS];   // ANIM: self.transform.transform(time) (emitter.rs:168,175)
    if (ANIM) instance_xf_at<ANIM>(sc, light, ln.time, ln.col, x);
    if (light->kind == TRAY_INST_POINT_EMITTER) {
        f3 pos = ANIM ? xf_point_affine_w(x, x[25], mk(0.0f, 0.0f, 0.0f)) : xf_point(light->mat, mk(0.0f, 0.0f, 0.0f));
        ln.wi_l = normalized(pos - ln.bsdf.p);
        ln.li = inst_emission<ANIM>(sc, light, ln.time) / length_sqr(pos - ln.bsdf.p);
        ln.pdf_l = 1.0f;
        p_w = pos;
    } else {
        float l2x, l2y;
        lane_2d(sc, ln, SD_L2, l2x, l2y);
        f3 p_l = ANIM ? xf_point_affine_w(x + 12, x[24], ln.bsdf.p) : xf_point(light->inv, ln.bsdf.p);
        f3 p_sampled, normal;
        geom_sample(light, p_l, l2x, l2y, p_sampled, normal);
        f3 w_il = normalized(p_sampled - p_l);
        ln.pdf_l = geom_pdf(light, p_l, w_il);
        ln.li = emitter_radiance<ANIM>(sc, light, -w_il, normal, ln.time);
        p_w = ANIM ? xf_point_affine_w(x, x[25], p_sampled) : xf_point(light->mat, p_sampled);
        ln.wi_l = ANIM ? xf_vector(x, w_il) : xf_vector(light->mat, w_il);
    }
    if (ln.pdf_l > 0.0f && !is_black(ln.li)) {
        ln.aux_d = p_w - ln.bsdf.p;
        ln.flags |= LF_SHADOW;
    }
}

// BSDF query: the one place where BSDF::eval / BSDF::pdf run. Three kinds of query reach it:
//   WANT_LIGHT  light half of estimate_direct after an unoccluded shadow ray (mod.rs:127-139)
//   WANT_MIS    BSDF half of estimate_direct (mod.rs:141-153), may set LF_MIS (ray for stage C)
//   WANT_PATH   path continuation (path.rs:84-110): next stage A ray, or LF_LAST
// Returns the follow-up query.
template <int ANIM, int FEAT, uint32_t KM = KM_ALL>
TR_DEV uint32_t query_stage(const DevScene& sc, Lane& ln, uint32_t want, const f3 wo_sh) {
    const bool is_light = want == WANT_LIGHT, mis = want == WANT_MIS;
    const uint32_t flags = want == WANT_PATH ? BX_ALL : BX_NON_SPECULAR;
    const TrayInstance* __restrict__ light = sc.instances + ln.light_inst;
    const bool delta = light->kind == TRAY_INST_POINT_EMITTER;
    SampleHead h;
    TR_QCLK_START(ln);
    if (is_light) {
        h.wi_world = ln.wi_l; h.f = mk(0.0f, 0.0f, 0.0f); h.pdf = 0.0f; h.sampled_type = 0u;
        h.need_eval = true; h.need_pdf = !delta;
    } else {
        float u0, u1;
        lane_2d(sc, ln, mis ? SD_B2 : SD_P2, u0, u1);
        float one_d = lane_1d(sc, ln, mis ? SD_B1 : SD_P1);
        h = bsdf_sample_head_sh<FEAT, KM>(ln.bsdf, wo_sh, flags, u0, u1, one_d);
    }
    TR_QCLK(ln, 0);
    if (h.need_eval || h.need_pdf) {
        const f3 wi_sh = normalized(to_shading(ln.bsdf, h.wi_world));
        bsdf_eval_pdf_sh<FEAT, KM>(ln.bsdf, wo_sh, wi_sh, flags, h.need_eval, h.need_pdf, h.f, h.pdf);
    }
    TR_QCLK(ln, 1);
    const f3 f = h.f, w_i = h.wi_world;
    const float pdf = h.pdf;
    if (is_light) {
        if (!is_black(f)) {
            if (delta) {
                ln.direct = f * ln.li * fabsf(dot(ln.wi_l, ln.bsdf.n)) / ln.pdf_l;
            } else {
                float w = power_heuristic(1.0f, ln.pdf_l, 1.0f, pdf);
                ln.direct = f * ln.li * fabsf(dot(ln.wi_l, ln.bsdf.n)) * w / ln.pdf_l;
            }
        }
        return delta ? WANT_PATH : WANT_MIS;
    }
    if (mis) {
        if (pdf > 0.0f && !is_black(f)) {
            float w = 1.0f;
            if (!(h.sampled_type & BX_SPECULAR)) {
                // Light::pdf (emitter.rs:193-203)
                f3 p_l, wl;
                if (ANIM) {
                    float x[TR_XF_WORDS];
                    instance_inv_any<ANIM>(sc, light, ln.time, ln.col, x);   // (Light::pdf needs the inverse only)
                    p_l = xf_point_affine_w(x + 12, x[24], ln.bsdf.p);
                    wl = normalized(xf_vector(x + 12, w_i));
                } else {
                    p_l = xf_point(light->inv, ln.bsdf.p);
                    wl = normalized(xf_vector(light->inv, w_i));
                }
                float pl = geom_pdf(light, p_l, wl);
                if (pl == 0.0f) return WANT_PATH;   // `return direct_light` (mod.rs:146-148)
                w = power_heuristic(1.0f, pdf, 1.0f, pl);
            } else ln.flags |= LF_MIS_UNTESTED;
            // direct += f * li * |cos| * w / pdf_bsdf once li is known (mod.rs:163-165): keep the factors
            ln.mis_f = f;
            ln.li = mk(fabsf(dot(w_i, ln.bsdf.n)), w, pdf);
            ln.aux_d = w_i;
            ln.flags |= LF_MIS;
        }
        return WANT_PATH;
    }
    // path.rs:84-110 (the `illum += throughput * li` of path.rs:82 is applied in vertex_end)
    if (is_black(f) || pdf == 0.0f) { ln.flags |= LF_LAST; return WANT_NONE; }
    ln.flags = (h.sampled_type & BX_SPECULAR) ? (ln.flags | LF_SPECULAR) : (ln.flags & ~LF_SPECULAR);
    ln.throughput = ln.throughput * f * fabsf(dot(w_i, ln.bsdf.n)) / pdf;
    if (ln.bounce > sc.min_depth) {   // quirk Q2
        float cont_prob = fmaxf(0.5f, luminance(ln.throughput));
        if (rr_draw(ln.ks, ln.bounce) > cont_prob) { ln.flags |= LF_LAST; return WANT_NONE; }
        ln.throughput = ln.throughput / cont_prob;
    }
    if (ln.bounce == sc.max_depth) { ln.flags |= LF_LAST; return WANT_NONE; }
    ln.d = normalized(w_i);
    return WANT_NONE;
}

// Stage B after the occlusion ray: all BSDF queries of the vertex. Every lane of the wave enters (`active` = the lane has a vertex):
// the three kinds of query run as three wave-aligned passes -- LIGHT for the lanes whose shadow ray was not occluded, then MIS,
// then PATH -- so that the kind is wave-uniform inside query_stage (scalar branches; the sample head runs twice per vertex, not
// three times as it did when occluded lanes started with MIS while the others were still at LIGHT).
template <int ANIM, int FEAT, uint32_t KM = KM_ALL>
TR_DEV void vertex_queries(const DevScene& sc, Lane& ln, bool occluded, bool active = true) {
    uint32_t want = WANT_NONE;
    const DevMaterial* table_mat = nullptr;
    DevMaterial hit_mat;
    f3 wo_sh = mk(0.0f, 0.0f, 1.0f);
    if (active) {
        const bool delta = sc.instances[ln.light_inst].kind == TRAY_INST_POINT_EMITTER;
        want = ((ln.flags & LF_SHADOW) && !occluded) ? WANT_LIGHT : (delta ? WANT_PATH : WANT_MIS);
        // a textured material's lobes exist per hit only: lowered here from the parameters sampled at (u, v, time), as Material::bsdf
        // does (matte.rs:55-63 etc.), kept in private memory for the queries of this vertex
        table_mat = ln.bsdf.mat;
        if ((FEAT & FEAT_TEX) && table_mat->textured) {
            resolve_textured(sc, table_mat, ln.bsdf.u, ln.bsdf.v, ln.time, hit_mat);
            ln.bsdf.mat = &hit_mat;
        }
        // w_o in shading space, once per vertex: BSDF::eval, ::pdf and ::sample each start with the same to_shading + normalized of
        // the same vector (bsdf.rs:67-68,86,115-116). w_o is -d until the PATH query (the last one) writes the next ray's direction.
        wo_sh = normalized(to_shading(ln.bsdf, -ln.d));
    }
#pragma nounroll
    for (uint32_t kind = WANT_LIGHT; kind <= WANT_PATH; ++kind) {   // LIGHT -> MIS -> PATH
#if defined(TR_EMU_PROFILE)   // divergence-profile build of tests/emu: the passes numbered
        TR_EMU_PHASE((int)kind);
#endif
        const bool go = want == kind;
        if (!__any(go)) continue;
        if (go) want = query_stage<ANIM, FEAT, KM>(sc, ln, kind, wo_sh);
        TR_QCLK(ln, 2);
    }
#if defined(TR_EMU_PROFILE)
    TR_EMU_PHASE(0);
#endif
    if ((FEAT & FEAT_TEX) && active) ln.bsdf.mat = table_mat;
}

// Before stage C. The ray Ray::segment(p, w_i, 0.001, inf) only matters if its closest hit is this light (mod.rs:155-161). If the
// light's own primitive is missed by the ray as Instance::intersect will see it (object space, direction not renormalised,
// receiver.rs:29-35) with the ORIGINAL range, it is missed with every smaller max_t the traversal would bring (the range tests of
// sphere / rectangle / disk only reject more as max_t shrinks) and whatever its boxes say: the hit is not the light, li is black,
// nothing is added -- such a ray is counted and not traced (LF_MIS -> LF_MIS_MISS). A sphere light's pdf is the cone pdf for ANY
// direction (sphere.rs:126-140): without this every diffuse vertex of smallpt traced a third full ray (stage C was 15.6 % of its
// wave cycles; smallpt 576 -> 681 Msamples/s at 64 spp). Rectangle / disk lights answer the same question inside geom_pdf
// (normalised direction): testing again cost cornell_box 2 %, so the test runs only where no pdf test ran -- sphere lights, and
// specular samples of any light.
template <int ANIM>
TR_DEV void mis_ray_filter(const DevScene& sc, Lane& ln) {
    if (!(ln.flags & LF_MIS)) return;
    const TrayInstance* __restrict__ light = sc.instances + ln.light_inst;
    const uint32_t gt = light->geom_type;
    if (gt != TRAY_GEOM_SPHERE && !(ln.flags & LF_MIS_UNTESTED)) return;
    f3 p_l, d_l;
    if (ANIM) {
        float x[TR_XF_WORDS];
        instance_inv_any<ANIM>(sc, light, ln.time, ln.col, x);
        p_l = xf_point_affine_w(x + 12, x[24], ln.bsdf.p);
        d_l = xf_vector(x + 12, ln.aux_d);
    } else {
        p_l = xf_point(light->inv, ln.bsdf.p);
        d_l = xf_vector(light->inv, ln.aux_d);
    }
    float t_;
    const bool may_hit = gt == TRAY_GEOM_RECT ? rect_test(light->geom_params[0], light->geom_params[1], p_l, d_l, 0.001f, TR_INF, t_)
                       : (gt == TRAY_GEOM_SPHERE ? sphere_test(light->geom_params[0], p_l, d_l, 0.001f, TR_INF, t_)
                                                 : disk_test(light->geom_params[0], light->geom_params[1], p_l, d_l, 0.001f, TR_INF, t_));
    if (!may_hit) ln.flags = (ln.flags & ~LF_MIS) | LF_MIS_MISS;
}

// Stage C: tail of the BSDF half of estimate_direct (mod.rs:154-166), then path.rs:82 and the
// bookkeeping for the next vertex. Returns false when the camera sample is finished.
template <int ANIM>
TR_DEV bool vertex_end(const DevScene& sc, Lane& ln, bool mis_hit, const HitRec& rec) {
    if ((ln.flags & LF_MIS) && mis_hit && rec.inst == ln.light_inst) {   // same emitter object (mod.rs:157-160)
        const TrayInstance* __restrict__ light = sc.instances + ln.light_inst;
        const Ray r = stage_c_ray(ln);
        f3 ng = finish_hit_ng<ANIM>(sc, r, rec);
        f3 li2 = emitter_radiance<ANIM>(sc, light, -ln.aux_d, ng, ln.time);
        if (!is_black(li2)) ln.direct = ln.direct + ln.mis_f * li2 * ln.li.x * ln.li.y / ln.li.z;
    }
    ln.illum = ln.illum + ln.t_vertex * ln.direct;   // path.rs:82
    if (ln.flags & LF_LAST) return false;
    ln.bounce = ln.bounce + 1u;
    return true;
}

TR_DEV f3 lane_result(const Lane& ln) {   // per-sample clamp (multithreaded.rs:98-99, quirk Q3)
    return mk(clampf(ln.illum.x, 0.0f, 1.0f), clampf(ln.illum.y, 0.0f, 1.0f), clampf(ln.illum.z, 0.0f, 1.0f));
}

}  // namespace tr
