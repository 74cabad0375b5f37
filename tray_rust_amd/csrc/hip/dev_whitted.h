// Whitted integrator (integrator/whitted.rs:41-68 with Integrator::specular_reflection / specular_transmission,
// integrator/mod.rs:49-97) for the lanes of one wave: the reference's recursion -- illumination() calls itself through the
// reflected and through the transmitted ray of every hit -- runs as an explicit frame stack per lane, and the wave steps
// through it together because the traversal (trace(): cooperative small-mesh test) wants the whole wave. One iteration of
// whitted_run = one closest-hit trace per lane that has a pending ray, the shadow rays of ALL lights for the lanes that
// hit (whitted.rs:56-62; the same 2-D sample for every light), the sampling of both specular children, then the unwinding
// of finished frames with the reference's own nesting of the arithmetic: parent.illum += f * li_child * |cos| / pdf.
//
// Random numbers (TRAY-CBRNG): every activation of illumination() is a NODE with a 32-bit key; the camera
// ray's node has the sample key (key_sample), the node behind the reflected / transmitted ray has draw(key, WD_CHILD + 0 / 1).
// LowDiscrepancy::get_samples_2d / _1d on a one-element slice (whitted.rs:46-47, mod.rs:57-60) is the scrambled (0,2)-sequence
// point of index 0 -- van_der_corput(0, s0), sobol(0, s1) with fresh scrambles, a shuffle of one element -- so a node's
// draws are: WD_L2 (+1) light sample, WD_R2 (+1), WD_R1 reflection, WD_T2 (+1), WD_T1 transmission.
// Not a hot path: frames live in private memory (scratch), one instantiation per ANIM with every lobe compiled in.
#pragma once
#include "dev_integrator.h"

namespace tr {

enum : uint32_t { WD_L2 = 0, WD_R2 = 2, WD_R1 = 4, WD_T2 = 5, WD_T1 = 7, WD_CHILD = 8 };
#define WH_MAX_DEPTH 16                 // deepest recursion the frame stack holds (tray_scene_create rejects more)
#define WH_MAX_FRAMES (WH_MAX_DEPTH + 1)

struct WhFrame {        // one activation of Whitted::illumination
    f3 illum;           // whitted.rs:48-67, accumulated in the reference's order
    f3 p;               // bsdf.p: origin of both children (ray.child(&bsdf.p, &w_i))
    f3 wr, fr;          // reflection child: direction, f
    f3 wt, ft;          // transmission child
    float cr, pr, ct, pt;   // |dot(w_i, bsdf.n)| and pdf of either child; pdf == 0 marks "no child"
    uint32_t key, depth, state;   // state 0: reflection child not yet run, 1: it has returned, 2: so has the transmission child
};

// Light::sample_incident for an instance of the light list (emitter.rs:165-186), as vertex_begin runs it
template <int ANIM>
TR_DEV void wh_light_sample(const DevScene& sc, const TrayInstance* __restrict__ light, f3 p, float u0, float u1, float time, uint32_t col,
                            f3& li, f3& w_i, float& pdf, f3& p_w) {
    float x[TR_XF_WORDS];
    if (ANIM) instance_xf_at<ANIM>(sc, light, time, col, x);
    if (light->kind == TRAY_INST_POINT_EMITTER) {
        const f3 pos = ANIM ? xf_point_affine_w(x, x[25], mk(0.0f, 0.0f, 0.0f)) : xf_point(light->mat, mk(0.0f, 0.0f, 0.0f));
        w_i = normalized(pos - p);
        li = inst_emission<ANIM>(sc, light, time) / length_sqr(pos - p);
        pdf = 1.0f;
        p_w = pos;
    } else {
        const f3 p_l = ANIM ? xf_point_affine_w(x + 12, x[24], p) : xf_point(light->inv, p);
        f3 p_sampled, normal;
        geom_sample(light, p_l, u0, u1, p_sampled, normal);
        const f3 w_il = normalized(p_sampled - p_l);
        pdf = geom_pdf(light, p_l, w_il);
        li = emitter_radiance<ANIM>(sc, light, -w_il, normal, time);
        p_w = ANIM ? xf_point_affine_w(x, x[25], p_sampled) : xf_point(light->mat, p_sampled);
        w_i = ANIM ? xf_vector(x, w_il) : xf_vector(light->mat, w_il);
    }
}

// Runs the camera sample of every lane of the wave to its end. Called by ALL lanes; `active` = the lane has a camera ray.
// Returns Whitted::illumination of the camera ray's hit (black on a miss, multithreaded.rs:102). cnt: this lane's activations of
// illumination() and rays; n_vertices / n_rays: the same as wave totals (the tile kernel's statistics).
template <int ANIM>
TR_DEV f3 whitted_run(const DevScene& sc, const DevScene* __restrict__ scp, uint32_t* __restrict__ stack, const Ray& cam, uint32_t ks, bool active,
                      Counters& cnt, uint32_t& n_vertices, uint32_t& n_rays, uint32_t smp_kind = TRAY_SAMPLER_LOW_DISCREPANCY, uint32_t smp_offset = 0u) {
    // the one-element arrays of an activation under the scene's Sampler (whitted.rs:46-47, mod.rs:59-60,83-84): LowDiscrepancy -- the scrambled
    // (0,2) point of index 0; Adaptive -- of index samples_taken (adaptive.rs:112-121); Uniform -- plain uniform draws (uniform.rs:36-46).
    // (constants at the tile kernel's call site: only the first form is compiled there)
#define WH_VDC(key_, d_) (smp_kind == TRAY_SAMPLER_UNIFORM ? (float)(draw(key_, d_) >> 8) / 16777216.0f : van_der_corput(smp_kind == TRAY_SAMPLER_ADAPTIVE ? smp_offset : 0u, draw(key_, d_)))
#define WH_SOB(key_, d_) (smp_kind == TRAY_SAMPLER_UNIFORM ? (float)(draw(key_, d_) >> 8) / 16777216.0f : sobol(smp_kind == TRAY_SAMPLER_ADAPTIVE ? smp_offset : 0u, draw(key_, d_)))
    constexpr int FEAT = FEAT_ALL | FEAT_TEX;
    WhFrame frames[WH_MAX_FRAMES];
    int sp = 0;
    bool pending = active, done = !active;
    f3 ro = cam.o, rd = cam.d;
    float rmin = 0.0f;
    uint32_t rdepth = 0u, rkey = ks;
    f3 result = mk(0.0f, 0.0f, 0.0f);
    while (__any(!done)) {
        // ---- scene.intersect of the pending rays
        Ray r;
        r.o = ro; r.d = rd; r.min_t = rmin; r.max_t = TR_INF; r.time = cam.time; r.col = cam.col;
        n_rays += (uint32_t)__popcll(__ballot(pending));
        if (pending) cnt.rays++;
        const TraceResult tr_ = trace<ANIM>(scp, stack, r, false, pending);
        const bool shade = pending && tr_.hit;   // a new activation of illumination()
        n_vertices += (uint32_t)__popcll(__ballot(shade));
        if (shade) cnt.vertices++;
        pending = false;
        Bsdf bsdf;
        bsdf.p = bsdf.n = bsdf.tan = mk(0.0f, 0.0f, 0.0f); bsdf.u = bsdf.v = 0.0f; bsdf.mat = nullptr; bsdf.merl_data = nullptr;
        DevMaterial hit_mat;
        f3 illum = mk(0.0f, 0.0f, 0.0f), w_o = -rd;
        float l2x = 0.0f, l2y = 0.0f;
        if (shade) {
            const Hit hit = finish_hit<ANIM>(sc, r, tr_.rec);
            bsdf = make_bsdf(sc, hit);
            if (bsdf.mat->textured) {   // Material::bsdf of a textured material: per hit (dev_tex.h)
                resolve_textured(sc, bsdf.mat, bsdf.u, bsdf.v, r.time, hit_mat);
                bsdf.mat = &hit_mat;
            }
            l2x = WH_VDC(rkey, WD_L2); l2y = WH_SOB(rkey, WD_L2 + 1u);
            const TrayInstance* __restrict__ inst = sc.instances + hit.inst;
            if (rdepth == 0u && inst->kind != TRAY_INST_RECEIVER)   // whitted.rs:49-54
                illum = illum + emitter_radiance<ANIM>(sc, inst, w_o, hit.ng, r.time);
        }
        // ---- every light of the scene (whitted.rs:56-62)
        for (uint32_t k = 0; k < sc.n_lights; ++k) {
            f3 li = mk(0.0f, 0.0f, 0.0f), w_i = li, f = li, p_w = li;
            float pdf = 0.0f;
            bool want = false;
            if (shade) {
                const TrayInstance* __restrict__ light = sc.instances + sc.lights[k];
                wh_light_sample<ANIM>(sc, light, bsdf.p, l2x, l2y, r.time, r.col, li, w_i, pdf, p_w);
                f = bsdf_eval<FEAT>(bsdf, w_o, w_i, BX_ALL);
                want = !is_black(li) && !is_black(f);
            }
            Ray sr;   // OcclusionTester::test_points (light/mod.rs:21-23)
            sr.o = bsdf.p; sr.d = p_w - bsdf.p; sr.min_t = 0.001f; sr.max_t = 0.999f; sr.time = r.time; sr.col = r.col;
            const unsigned long long wm = __ballot(want);
            if (wm != 0ull) {
                n_rays += (uint32_t)__popcll(wm);
                if (want) cnt.rays++;
                const TraceResult occ = trace<ANIM>(scp, stack, sr, true, want);
                if (want && !occ.hit) illum = illum + f * li * fabsf(dot(w_i, bsdf.n)) / pdf;
            }
        }
        // ---- the two specular children (mod.rs:49-97), sampled now: the draws of a node do not depend on the order of use
        if (shade) {
            WhFrame fr;
            fr.illum = illum; fr.p = bsdf.p; fr.key = rkey; fr.depth = rdepth; fr.state = 0u;
            fr.pr = 0.0f; fr.pt = 0.0f; fr.cr = 0.0f; fr.ct = 0.0f;
            fr.wr = fr.fr = fr.wt = fr.ft = mk(0.0f, 0.0f, 0.0f);
            if (rdepth < sc.max_depth) {   // whitted.rs:63
                f3 w_i;
                float pdf;
                uint32_t ty;
                f3 f = bsdf_sample(bsdf, w_o, BX_SPECULAR | BX_REFLECTION, WH_VDC(rkey, WD_R2), WH_SOB(rkey, WD_R2 + 1u), WH_VDC(rkey, WD_R1), w_i, pdf, ty);
                float c = fabsf(dot(w_i, bsdf.n));
                if (pdf > 0.0f && !is_black(f) && c != 0.0f) { fr.wr = w_i; fr.fr = f; fr.cr = c; fr.pr = pdf; }
                f = bsdf_sample(bsdf, w_o, BX_SPECULAR | BX_TRANSMISSION, WH_VDC(rkey, WD_T2), WH_SOB(rkey, WD_T2 + 1u), WH_VDC(rkey, WD_T1), w_i, pdf, ty);
                c = fabsf(dot(w_i, bsdf.n));
                if (pdf > 0.0f && !is_black(f) && c != 0.0f) { fr.wt = w_i; fr.ft = f; fr.ct = c; fr.pt = pdf; }
            }
            frames[sp] = fr;
            ++sp;
        }
        // ---- unwind: run the next child of the top frame, or return its value to the frame below
        if (!done) {
            bool has_carry = false;
            f3 carry = mk(0.0f, 0.0f, 0.0f);
            for (;;) {
                if (sp == 0) { done = true; break; }   // the camera ray missed: black (multithreaded.rs:102)
                WhFrame& f = frames[sp - 1];
                if (has_carry) {   // refl / transmit = f * li * |cos| / pdf (mod.rs:71, 94), illum = illum + that (whitted.rs:64-65)
                    if (f.state == 1u) f.illum = f.illum + f.fr * carry * f.cr / f.pr;
                    else f.illum = f.illum + f.ft * carry * f.ct / f.pt;
                    has_carry = false;
                }
                if (f.state == 0u) {
                    f.state = 1u;
                    if (f.pr > 0.0f) { ro = f.p; rd = f.wr; rmin = 0.001f; rdepth = f.depth + 1u; rkey = draw(f.key, WD_CHILD); pending = true; break; }
                    continue;
                }
                if (f.state == 1u) {
                    f.state = 2u;
                    if (f.pt > 0.0f) { ro = f.p; rd = f.wt; rmin = 0.001f; rdepth = f.depth + 1u; rkey = draw(f.key, WD_CHILD + 1u); pending = true; break; }
                    continue;
                }
                carry = f.illum; has_carry = true;
                --sp;
                if (sp == 0) { result = carry; done = true; break; }
            }
        }
    }
    return result;
#undef WH_VDC
#undef WH_SOB
}

}  // namespace tr
