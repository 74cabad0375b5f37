// Device-side camera, per-sample LD sampler and the Path integrator as a per-lane phase machine.
//   Camera::generate_ray            film/camera.rs:150-157
//   LowDiscrepancy::get_samples*    sampler/ld.rs:33-64 (draws replaced by TRAY-CBRNG)
//   Path::illumination              integrator/path.rs:45-120
//   sample_one_light/estimate_direct integrator/mod.rs:106-169
//   Light for Emitter, OcclusionTester  geometry/emitter.rs:164-203, light/mod.rs:14-39
//
// The reference recurses Path::illumination -> estimate_direct -> Scene::intersect three times per
// path vertex. Here every lane is a small state machine whose step is "trace ONE ray, then process
// the answer": PH_EXTEND (camera / continuation ray), PH_SHADOW (occlusion ray of the light sample),
// PH_MIS (BSDF-sampled ray of estimate_direct). All lanes of a wave therefore meet in the same
// traversal code whatever stage of the reference's control flow they are in; the arithmetic and the
// order of every evaluation are the reference's.
#pragma once
#include "dev_bsdf.h"

namespace tr {

TR_DEV Ray camera_ray(const DevScene& sc, float px, float py, float time) {
    const TrayCamera& c = sc.camera;
    f3 q = xf_point(c.raster_to_cam, mk(px, py, 0.0f));
    f3 px_pos = mk(c.scaling[0], c.scaling[1], c.scaling[2]) * q;
    f3 d = normalized(px_pos);
    (void)time;   // frame_time only selects the (unanimated) camera / instance transforms
    Ray r;
    r.o = xf_point(c.cam_world, mk(0.0f, 0.0f, 0.0f));
    r.d = xf_vector(c.cam_world, d);
    r.min_t = 0.0f; r.max_t = TR_INF;
    return r;
}

// Per-pixel part of the sampler: sub-pixel position and shutter time of sample s
struct PixelSampler {
    uint32_t kp, scr_x, scr_y, key_xy, scr_t, key_t;
};
TR_DEV PixelSampler pixel_sampler(uint32_t kf, uint32_t pixel_index) {
    PixelSampler p;
    p.kp = key_pixel(kf, pixel_index);
    p.scr_x = draw(p.kp, PD_SCR_X); p.scr_y = draw(p.kp, PD_SCR_Y); p.key_xy = draw(p.kp, PD_PERM_XY);
    p.scr_t = draw(p.kp, PD_SCR_T); p.key_t = draw(p.kp, PD_PERM_T);
    return p;
}
TR_DEV void pixel_sample(const PixelSampler& p, uint32_t s, uint32_t spp, uint32_t px, uint32_t py, float& x, float& y, float& t) {
    uint32_t idx = permute(s, spp, p.key_xy);
    x = van_der_corput(idx, p.scr_x) + (float)px;   // ld.rs:43-46
    y = sobol(idx, p.scr_y) + (float)py;
    t = van_der_corput(permute(s, spp, p.key_t), p.scr_t);
}

TR_DEV float rr_draw(uint32_t ks, uint32_t bounce) { return (float)(draw(ks, SD_RR + bounce) >> 8) / 16777216.0f; }   // Rng::next_f32

TR_DEV f3 inst_emission(const TrayInstance* __restrict__ in) { return mk(in->emission[0], in->emission[1], in->emission[2]); }
// Emitter::radiance (emitter.rs:140-142)
TR_DEV f3 emitter_radiance(const TrayInstance* __restrict__ in, f3 w, f3 n) {
    return dot(w, n) > 0.0f ? inst_emission(in) : mk(0.0f, 0.0f, 0.0f);
}

enum : uint32_t { PH_NEW = 0, PH_EXTEND = 1, PH_SHADOW = 2, PH_MIS = 3, PH_DONE = 4 };
enum : uint32_t { WANT_NONE = 0, WANT_LIGHT = 1, WANT_MIS = 2, WANT_PATH = 3 };

// Everything a lane keeps between two traced rays
struct Lane {
    Ray ray;               // ray to trace next / traced last
    uint32_t phase;
    uint32_t bounce;       // index of the path vertex being shaded
    bool specular_bounce;
    f3 throughput, illum;
    f3 first_ng;           // hit.dg.ng of the camera ray's hit (quirk Q1)
    // shading context of the current vertex (BSDF::new)
    Bsdf bsdf;
    f3 w_o;
    // light sample waiting for its occlusion ray (PH_SHADOW)
    uint32_t light_inst;
    f3 li, wi_l;
    float pdf_l;
    f3 direct;             // direct_light of estimate_direct, accumulated over the light / BSDF halves
    f3 mis_weight;         // f * |cos| * w / pdf_bsdf of the BSDF-sampled half (PH_MIS)
    // per camera sample LD arrays (path.rs:48-60): only the sample key is kept; scrambles and
    // shuffle entries are re-derived from it when a bounce needs them
    uint32_t ks;
};

TR_DEV void lane_start_sample(const DevScene& sc, Lane& ln, const Ray& cam_ray, uint32_t ks) {
    ln.ray = cam_ray;
    ln.phase = PH_EXTEND;
    ln.bounce = 0u;
    ln.specular_bounce = false;
    ln.throughput = mk(1.0f, 1.0f, 1.0f);
    ln.illum = mk(0.0f, 0.0f, 0.0f);
    ln.ks = ks;
    (void)sc;
}

// sample_02 / van_der_corput of array `a` at the current bounce (ld.rs:54-64, 91-93)
TR_DEV void lane_2d(const DevScene& sc, const Lane& ln, uint32_t dim, float& u0, float& u1) {
    uint32_t idx = shuffle_entry(draw(ln.ks, dim + 2u), sc.max_depth + 1u, ln.bounce);
    u0 = van_der_corput(idx, draw(ln.ks, dim));
    u1 = sobol(idx, draw(ln.ks, dim + 1u));
}
TR_DEV float lane_1d(const DevScene& sc, const Lane& ln, uint32_t dim) {
    return van_der_corput(shuffle_entry(draw(ln.ks, dim + 1u), sc.max_depth + 1u, ln.bounce), draw(ln.ks, dim));
}

// After the PH_EXTEND ray hit something: head of the loop body of Path::illumination (path.rs:69-82)
// up to the light half of estimate_direct (mod.rs:124-127). Returns WANT_* for the sampling stage.
TR_DEV uint32_t shade_extend(const DevScene& sc, Lane& ln, const HitRec& rec, Counters& cnt) {
    cnt.vertices++;
    Hit hit = finish_hit(sc, ln.ray, rec);
    if (ln.bounce == 0u) ln.first_ng = hit.ng;
    const TrayInstance* __restrict__ inst = sc.instances + hit.inst;
    if (ln.bounce == 0u || ln.specular_bounce) {
        if (inst->kind != TRAY_INST_RECEIVER) {
            f3 w = -ln.ray.d;
            ln.illum = ln.illum + ln.throughput * emitter_radiance(inst, w, ln.first_ng);
        }
    }
    ln.bsdf = make_bsdf(sc, hit);
    ln.w_o = -ln.ray.d;
    ln.direct = mk(0.0f, 0.0f, 0.0f);
    // sample_one_light (mod.rs:106-111), no 1/p_select (quirk Q6)
    float l1 = lane_1d(sc, ln, SD_L1);
    float fl = l1 * (float)sc.n_lights;
    uint32_t li_idx = fl > 0.0f ? (uint32_t)fl : 0u;
    if (li_idx > sc.n_lights - 1u) li_idx = sc.n_lights - 1u;
    ln.light_inst = sc.lights[li_idx];
    const TrayInstance* __restrict__ light = sc.instances + ln.light_inst;
    // Light::sample_incident (emitter.rs:165-186)
    f3 p_w;
    if (light->kind == TRAY_INST_POINT_EMITTER) {
        f3 pos = xf_point(light->mat, mk(0.0f, 0.0f, 0.0f));
        ln.wi_l = normalized(pos - ln.bsdf.p);
        ln.li = inst_emission(light) / length_sqr(pos - ln.bsdf.p);
        ln.pdf_l = 1.0f;
        p_w = pos;
    } else {
        float l2x, l2y;
        lane_2d(sc, ln, SD_L2, l2x, l2y);
        f3 p_l = xf_point(light->inv, ln.bsdf.p);
        f3 p_sampled, normal;
        geom_sample(light, p_l, l2x, l2y, p_sampled, normal);
        f3 w_il = normalized(p_sampled - p_l);
        ln.pdf_l = geom_pdf(light, p_l, w_il);
        ln.li = emitter_radiance(light, -w_il, normal);
        p_w = xf_point(light->mat, p_sampled);
        ln.wi_l = xf_vector(light->mat, w_il);
    }
    if (ln.pdf_l > 0.0f && !is_black(ln.li)) {
        // OcclusionTester::test_points (light/mod.rs:21-23): unnormalised segment (quirk Q4)
        ln.ray.o = ln.bsdf.p; ln.ray.d = p_w - ln.bsdf.p; ln.ray.min_t = 0.001f; ln.ray.max_t = 0.999f;
        ln.phase = PH_SHADOW;
        return WANT_NONE;
    }
    return light->kind == TRAY_INST_POINT_EMITTER ? WANT_PATH : WANT_MIS;
}

// BSDF query stage: the one place where BSDF::eval / BSDF::pdf run. Three kinds of query reach it:
//   WANT_LIGHT  light half of estimate_direct after an unoccluded shadow ray (mod.rs:127-139)
//   WANT_MIS    BSDF half of estimate_direct (mod.rs:141-153), may queue the PH_MIS ray
//   WANT_PATH   path continuation (path.rs:82-115), queues the PH_EXTEND ray or ends the sample
// Returns the follow-up query, or WANT_NONE once a ray is queued / the sample is finished.
TR_DEV uint32_t query_stage(const DevScene& sc, Lane& ln, uint32_t want) {
    const bool is_light = want == WANT_LIGHT, mis = want == WANT_MIS;
    const uint32_t flags = want == WANT_PATH ? BX_ALL : BX_NON_SPECULAR;
    const TrayInstance* __restrict__ light = sc.instances + ln.light_inst;
    const bool delta = light->kind == TRAY_INST_POINT_EMITTER;
    SampleHead h;
    if (is_light) {
        h.wi_world = ln.wi_l; h.f = mk(0.0f, 0.0f, 0.0f); h.pdf = 0.0f; h.sampled_type = 0u;
        h.need_eval = true; h.need_pdf = !delta;
    } else {
        float u0, u1;
        lane_2d(sc, ln, mis ? SD_B2 : SD_P2, u0, u1);
        float one_d = lane_1d(sc, ln, mis ? SD_B1 : SD_P1);
        h = bsdf_sample_head(ln.bsdf, ln.w_o, flags, u0, u1, one_d);
    }
    if (h.need_eval) h.f = bsdf_eval(ln.bsdf, ln.w_o, h.wi_world, flags);
    if (h.need_pdf) h.pdf = bsdf_pdf(ln.bsdf, ln.w_o, h.wi_world, flags);
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
                f3 p_l = xf_point(light->inv, ln.bsdf.p);
                f3 wl = normalized(xf_vector(light->inv, w_i));
                float pl = geom_pdf(light, p_l, wl);
                if (pl == 0.0f) return WANT_PATH;   // `return direct_light` (mod.rs:146-148)
                w = power_heuristic(1.0f, pdf, 1.0f, pl);
            }
            // direct += f * li * |cos| * w / pdf_bsdf once li is known (mod.rs:163-165): keep the factors
            ln.mis_weight = f;
            ln.li = mk(fabsf(dot(w_i, ln.bsdf.n)), w, 0.0f);
            ln.pdf_l = pdf;
            ln.ray.o = ln.bsdf.p; ln.ray.d = w_i; ln.ray.min_t = 0.001f; ln.ray.max_t = TR_INF;
            ln.phase = PH_MIS;
            return WANT_NONE;
        }
        return WANT_PATH;
    }
    // path.rs:80-117
    ln.illum = ln.illum + ln.throughput * ln.direct;
    if (is_black(f) || pdf == 0.0f) { ln.phase = PH_NEW; return WANT_NONE; }
    ln.specular_bounce = (h.sampled_type & BX_SPECULAR) != 0u;
    ln.throughput = ln.throughput * f * fabsf(dot(w_i, ln.bsdf.n)) / pdf;
    if (ln.bounce > sc.min_depth) {   // quirk Q2
        float cont_prob = fmaxf(0.5f, luminance(ln.throughput));
        if (rr_draw(ln.ks, ln.bounce) > cont_prob) { ln.phase = PH_NEW; return WANT_NONE; }
        ln.throughput = ln.throughput / cont_prob;
    }
    if (ln.bounce == sc.max_depth) { ln.phase = PH_NEW; return WANT_NONE; }
    ln.ray.o = ln.bsdf.p;
    ln.ray.d = normalized(w_i);
    ln.ray.min_t = 0.001f; ln.ray.max_t = TR_INF;
    ln.bounce = ln.bounce + 1u;
    ln.phase = PH_EXTEND;
    return WANT_NONE;
}

// One step of the machine for a lane whose ray has just been traced. After the call either
// ln.phase is PH_NEW (sample finished: ln.illum is its radiance) or ln.ray holds the next ray.
TR_DEV void lane_step(const DevScene& sc, Lane& ln, bool hit, const HitRec& rec, Counters& cnt) {
    uint32_t want = WANT_NONE;
    if (ln.phase == PH_EXTEND) {
        if (!hit) { ln.phase = PH_NEW; return; }   // camera miss: black sample; continuation miss: path ends (path.rs:112-115)
        want = shade_extend(sc, ln, rec, cnt);
    } else if (ln.phase == PH_SHADOW) {
        const bool delta = sc.instances[ln.light_inst].kind == TRAY_INST_POINT_EMITTER;
        want = !hit ? WANT_LIGHT : (delta ? WANT_PATH : WANT_MIS);   // occluded: skip the light half
    } else {   // PH_MIS: direct += f * li * |cos| * w / pdf_bsdf, factors in the reference's order (mod.rs:154-165)
        if (hit && rec.inst == ln.light_inst) {   // same emitter object (mod.rs:157-160)
            const TrayInstance* __restrict__ light = sc.instances + ln.light_inst;
            f3 ng = finish_hit_ng(sc, ln.ray, rec);
            f3 li2 = emitter_radiance(light, -ln.ray.d, ng);
            if (!is_black(li2)) ln.direct = ln.direct + ln.mis_weight * li2 * ln.li.x * ln.li.y / ln.pdf_l;
        }
        want = WANT_PATH;
    }
    // at most three passes: WANT_LIGHT -> WANT_MIS -> WANT_PATH
    for (int pass = 0; pass < 3 && want != WANT_NONE; ++pass) want = query_stage(sc, ln, want);
}

TR_DEV f3 lane_result(const Lane& ln) {   // per-sample clamp (multithreaded.rs:98-99, quirk Q3)
    return mk(clampf(ln.illum.x, 0.0f, 1.0f), clampf(ln.illum.y, 0.0f, 1.0f), clampf(ln.illum.z, 0.0f, 1.0f));
}

}  // namespace tr
