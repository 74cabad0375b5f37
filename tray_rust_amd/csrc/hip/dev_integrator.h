// Device-side camera, per-sample LD sampler and the Path integrator, one path vertex per call.
//   Camera::generate_ray            film/camera.rs:150-157
//   LowDiscrepancy::get_samples*    sampler/ld.rs:33-64 (draws replaced by TRAY-CBRNG)
//   Path::illumination              integrator/path.rs:45-120
//   sample_one_light/estimate_direct integrator/mod.rs:106-169
//   Light for Emitter, OcclusionTester  geometry/emitter.rs:164-203, light/mod.rs:14-39
#pragma once
#include "dev_bsdf.h"

namespace tr {

TR_DEV Ray camera_ray(const DevScene& sc, float px, float py, float time) {
    const TrayCamera& c = sc.camera;
    f3 q = xf_point(c.raster_to_cam, mk(px, py, 0.0f));
    f3 px_pos = mk(c.scaling[0], c.scaling[1], c.scaling[2]) * q;
    f3 d = normalized(px_pos);
    float frame_time = (c.shutter_close - c.shutter_open) * time + c.shutter_open;
    Ray r;
    r.o = xf_point(c.cam_world, mk(0.0f, 0.0f, 0.0f));
    r.d = xf_vector(c.cam_world, d);
    r.min_t = 0.0f; r.max_t = TR_INF; r.time = frame_time;
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

// The six LD arrays of path.rs:48-60 for one camera sample, evaluated lazily per bounce
struct PathSampler {
    uint32_t ks;
    uint32_t scr[9];
    uint64_t perm[6];
};
TR_DEV void path_sampler_init(PathSampler& ps, uint32_t ks, uint32_t n) {
    ps.ks = ks;
    ps.scr[0] = draw(ks, SD_L2); ps.scr[1] = draw(ks, SD_L2 + 1); ps.perm[0] = shuffle_small(draw(ks, SD_L2 + 2), n);
    ps.scr[2] = draw(ks, SD_B2); ps.scr[3] = draw(ks, SD_B2 + 1); ps.perm[1] = shuffle_small(draw(ks, SD_B2 + 2), n);
    ps.scr[4] = draw(ks, SD_P2); ps.scr[5] = draw(ks, SD_P2 + 1); ps.perm[2] = shuffle_small(draw(ks, SD_P2 + 2), n);
    ps.scr[6] = draw(ks, SD_L1); ps.perm[3] = shuffle_small(draw(ks, SD_L1 + 1), n);
    ps.scr[7] = draw(ks, SD_B1); ps.perm[4] = shuffle_small(draw(ks, SD_B1 + 1), n);
    ps.scr[8] = draw(ks, SD_P1); ps.perm[5] = shuffle_small(draw(ks, SD_P1 + 1), n);
}
TR_DEV float rr_draw(uint32_t ks, uint32_t bounce) { return (float)(draw(ks, SD_RR + bounce) >> 8) / 16777216.0f; }   // Rng::next_f32

TR_DEV f3 inst_emission(const TrayInstance* __restrict__ in) { return mk(in->emission[0], in->emission[1], in->emission[2]); }
// Emitter::radiance (emitter.rs:140-142)
TR_DEV f3 emitter_radiance(const TrayInstance* __restrict__ in, f3 w, f3 n) {
    return dot(w, n) > 0.0f ? inst_emission(in) : mk(0.0f, 0.0f, 0.0f);
}

// Integrator::estimate_direct (integrator/mod.rs:122-169)
TR_DEV f3 estimate_direct(const DevScene& sc, f3 w_o, f3 p, const Bsdf& bsdf, float l2x, float l2y, float b2x, float b2y, float b1,
                          uint32_t light_inst, float time, Counters& cnt) {
    const uint32_t flags = BX_NON_SPECULAR;
    const TrayInstance* __restrict__ light = sc.instances + light_inst;
    const bool delta = light->kind == TRAY_INST_POINT_EMITTER;
    f3 direct = mk(0.0f, 0.0f, 0.0f);
    // Light::sample_incident (emitter.rs:165-186)
    f3 li, w_i, p_w;
    float pdf_light;
    if (delta) {
        f3 pos = xf_point(light->mat, mk(0.0f, 0.0f, 0.0f));
        w_i = normalized(pos - bsdf.p);
        li = inst_emission(light) / length_sqr(pos - bsdf.p);
        pdf_light = 1.0f;
        p_w = pos;
    } else {
        f3 p_l = xf_point(light->inv, bsdf.p);
        f3 p_sampled, normal;
        geom_sample(light, p_l, l2x, l2y, p_sampled, normal);
        f3 w_il = normalized(p_sampled - p_l);
        pdf_light = geom_pdf(light, p_l, w_il);
        li = emitter_radiance(light, -w_il, normal);
        p_w = xf_point(light->mat, p_sampled);
        w_i = xf_vector(light->mat, w_il);
    }
    bool unoccluded = false;
    if (pdf_light > 0.0f && !is_black(li)) {
        Ray sh;   // OcclusionTester::test_points (light/mod.rs:21-23): unnormalised segment (quirk Q4)
        sh.o = bsdf.p; sh.d = p_w - bsdf.p; sh.min_t = 0.001f; sh.max_t = 0.999f; sh.time = time;
        HitRec tmp;
        cnt.rays++;
        unoccluded = !scene_traverse<true>(sc, sh, tmp);
    }
    if (unoccluded) {
        f3 f = bsdf_eval(bsdf, w_o, w_i, flags);
        if (!is_black(f)) {
            if (delta) {
                direct = f * li * fabsf(dot(w_i, bsdf.n)) / pdf_light;
            } else {
                float pdf_bsdf = bsdf_pdf(bsdf, w_o, w_i, flags);
                float w = power_heuristic(1.0f, pdf_light, 1.0f, pdf_bsdf);
                direct = f * li * fabsf(dot(w_i, bsdf.n)) * w / pdf_light;
            }
        }
    }
    if (!delta) {
        f3 wi2;
        float pdf_bsdf;
        uint32_t sampled_type;
        f3 f = bsdf_sample(bsdf, w_o, flags, b2x, b2y, b1, wi2, pdf_bsdf, sampled_type);
        if (pdf_bsdf > 0.0f && !is_black(f)) {
            float w = 1.0f;
            if (!(sampled_type & BX_SPECULAR)) {
                // Light::pdf (emitter.rs:193-203)
                f3 p_l = xf_point(light->inv, p);
                f3 wl = normalized(xf_vector(light->inv, wi2));
                float pl = geom_pdf(light, p_l, wl);
                if (pl == 0.0f) return direct;
                w = power_heuristic(1.0f, pdf_bsdf, 1.0f, pl);
            }
            Ray r;
            r.o = p; r.d = wi2; r.min_t = 0.001f; r.max_t = TR_INF; r.time = time;
            f3 li2 = mk(0.0f, 0.0f, 0.0f);
            HitRec rec;
            cnt.rays++;
            if (scene_traverse<false>(sc, r, rec)) {
                if (rec.inst == light_inst) {   // same emitter object (mod.rs:157-160)
                    Hit h = finish_hit(sc, r, rec);
                    li2 = emitter_radiance(light, -wi2, h.ng);
                }
            }
            if (!is_black(li2)) direct = direct + f * li2 * fabsf(dot(wi2, bsdf.n)) * w / pdf_bsdf;
        }
    }
    return direct;
}

struct PathState {
    Ray ray;          // the ray that is (or was last) traced
    HitRec rec;       // closest hit of `ray`
    f3 throughput, illum;
    f3 first_ng;      // hit.dg.ng of the camera ray's hit (quirk Q1)
    uint32_t bounce;  // index of the vertex `rec` describes
    bool specular_bounce;
};

// One iteration of the loop in Path::illumination (path.rs:69-117) for the vertex in st.rec.
// Returns true when st.ray holds the continuation ray (to be traced by the caller, who then
// increments nothing: st.bounce already names the next vertex); false when the path ended.
TR_DEV bool path_vertex(const DevScene& sc, PathState& st, const PathSampler& ps, Counters& cnt) {
    cnt.vertices++;
    const uint32_t bounce = st.bounce;
    Hit hit = finish_hit(sc, st.ray, st.rec);
    if (bounce == 0u) st.first_ng = hit.ng;
    const TrayInstance* __restrict__ inst = sc.instances + hit.inst;
    if (bounce == 0u || st.specular_bounce) {
        if (inst->kind != TRAY_INST_RECEIVER) {
            f3 w = -st.ray.d;
            st.illum = st.illum + st.throughput * emitter_radiance(inst, w, st.first_ng);
        }
    }
    Bsdf bsdf = make_bsdf(sc, hit);
    f3 w_o = -st.ray.d;
    uint32_t iL2 = perm_at(ps.perm[0], bounce), iB2 = perm_at(ps.perm[1], bounce), iP2 = perm_at(ps.perm[2], bounce);
    float l2x = van_der_corput(iL2, ps.scr[0]), l2y = sobol(iL2, ps.scr[1]);
    float b2x = van_der_corput(iB2, ps.scr[2]), b2y = sobol(iB2, ps.scr[3]);
    float l1 = van_der_corput(perm_at(ps.perm[3], bounce), ps.scr[6]);
    float b1 = van_der_corput(perm_at(ps.perm[4], bounce), ps.scr[7]);
    // sample_one_light (mod.rs:106-111), no 1/p_select (quirk Q6)
    float fl = l1 * (float)sc.n_lights;
    uint32_t li_idx = fl > 0.0f ? (uint32_t)fl : 0u;
    if (li_idx > sc.n_lights - 1u) li_idx = sc.n_lights - 1u;
    f3 li = estimate_direct(sc, w_o, hit.p, bsdf, l2x, l2y, b2x, b2y, b1, sc.lights[li_idx], st.ray.time, cnt);
    st.illum = st.illum + st.throughput * li;

    float p2x = van_der_corput(iP2, ps.scr[4]), p2y = sobol(iP2, ps.scr[5]);
    float p1 = van_der_corput(perm_at(ps.perm[5], bounce), ps.scr[8]);
    f3 w_i;
    float pdf;
    uint32_t sampled_type;
    f3 f = bsdf_sample(bsdf, w_o, BX_ALL, p2x, p2y, p1, w_i, pdf, sampled_type);
    if (is_black(f) || pdf == 0.0f) return false;
    st.specular_bounce = (sampled_type & BX_SPECULAR) != 0u;
    st.throughput = st.throughput * f * fabsf(dot(w_i, bsdf.n)) / pdf;
    if (bounce > sc.min_depth) {   // quirk Q2
        float cont_prob = fmaxf(0.5f, luminance(st.throughput));
        if (rr_draw(ps.ks, bounce) > cont_prob) return false;
        st.throughput = st.throughput / cont_prob;
    }
    if (bounce == sc.max_depth) return false;
    st.ray.o = bsdf.p;
    st.ray.d = normalized(w_i);
    st.ray.min_t = 0.001f; st.ray.max_t = TR_INF;   // time is inherited (ray.rs:35-37)
    st.bounce = bounce + 1u;
    return true;
}

// Whole camera sample without regeneration (debug kernel): thread_work's inner loop body
// (multithreaded.rs:94-103). Returns the clamped radiance.
TR_DEV f3 trace_sample(const DevScene& sc, uint32_t kf, uint32_t px, uint32_t py, uint32_t s, uint32_t spp, float& sx, float& sy, Counters& cnt) {
    PixelSampler pix = pixel_sampler(kf, py * sc.width + px);
    float t;
    pixel_sample(pix, s, spp, px, py, sx, sy, t);
    PathState st;
    st.ray = camera_ray(sc, sx, sy, t);
    st.throughput = mk(1.0f, 1.0f, 1.0f);
    st.illum = mk(0.0f, 0.0f, 0.0f);
    st.first_ng = mk(0.0f, 0.0f, 0.0f);
    st.bounce = 0u;
    st.specular_bounce = false;
    PathSampler ps;
    path_sampler_init(ps, key_sample(pix.kp, s), sc.max_depth + 1u);
    for (;;) {
        cnt.rays++;
        if (!scene_traverse<false>(sc, st.ray, st.rec)) break;
        if (!path_vertex(sc, st, ps, cnt)) break;
    }
    return mk(clampf(st.illum.x, 0.0f, 1.0f), clampf(st.illum.y, 0.0f, 1.0f), clampf(st.illum.z, 0.0f, 1.0f));   // quirk Q3
}

}  // namespace tr
