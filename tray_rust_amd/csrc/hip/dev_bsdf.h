// Device-side shading: the reference's open Material / BxDF trait objects lowered to a closed set
// of lobes held in registers (no arena, no dynamic dispatch).
//   Material::bsdf      material/{matte,plastic,metal,glass,rough_glass,specular_metal,merl}.rs
//   BSDF                bxdf/bsdf.rs:38-134         BxDF defaults  bxdf/mod.rs:92-166
//   lobes               bxdf/{lambertian,oren_nayar,specular_reflection,specular_transmission,
//                             torrance_sparrow,microfacet_transmission,merl}.rs, microfacet/beckmann.rs,
//                             fresnel.rs
// Colours are RGB only: the reference's alpha channel never reaches the film (render_target.rs:142-145).
#pragma once
#include <cstring>

#include "dev_geom.h"

namespace tr {

enum { BX_REFLECTION = 1, BX_TRANSMISSION = 2, BX_DIFFUSE = 4, BX_GLOSSY = 8, BX_SPECULAR = 16 };
enum { BX_ALL = 31, BX_NON_SPECULAR = 15 };   // BxDFType::all / non_specular (bxdf/mod.rs:64-82)

enum { LB_LAMBERTIAN = 0, LB_OREN_NAYAR, LB_SPEC_REFL_DIEL, LB_SPEC_REFL_COND, LB_SPEC_TRANS, LB_TS_DIEL, LB_TS_COND, LB_MF_TRANS, LB_MERL };

// Register copy of one precomputed DevLobe
struct Lobe {
    uint32_t kind, type;
    f3 color;
    float eta_t;   // dielectric: Dielectric::new(1.0, eta_t)
    float width;   // Beckmann width | Oren-Nayar: a
    float ob;      // Oren-Nayar b
};

// BSDF: shading frame in registers, lobes fetched from the material table when needed
struct Bsdf {
    f3 p, n, tan;
    float u, v;   // hit.dg.u / v for textured materials (dead in kernels built without FEAT_TEX)
    const DevMaterial* __restrict__ mat;
    const float* __restrict__ merl_data;
};

TR_DEV Lobe load_lobe(const DevMaterial* __restrict__ m, int i) {
    const DevLobe* __restrict__ d = m->lobe + i;
    Lobe l;
    l.kind = d->kind; l.type = d->type;
    l.color = mk(d->color[0], d->color[1], d->color[2]);
    l.eta_t = d->eta_t; l.width = d->width; l.ob = d->ob;
    return l;
}

TR_DEV float cos_theta(f3 v) { return v.z; }
TR_DEV float cos_theta_sqr(f3 v) { return v.z * v.z; }
TR_DEV float sin_theta_sqr(f3 v) { return fmaxf(0.0f, 1.0f - v.z * v.z); }
TR_DEV float sin_theta(f3 v) { return sqrtf(sin_theta_sqr(v)); }
TR_DEV float tan_theta(f3 v) { float s2 = sin_theta_sqr(v); return s2 <= 0.0f ? 0.0f : sqrtf(s2) / cos_theta(v); }
TR_DEV float tan_theta_sqr(f3 v) { return sin_theta_sqr(v) / cos_theta_sqr(v); }
TR_DEV float cos_phi(f3 v) { float s = sin_theta(v); return s == 0.0f ? 1.0f : clampf(v.x / s, -1.0f, 1.0f); }
TR_DEV float sin_phi(f3 v) { float s = sin_theta(v); return s == 0.0f ? 0.0f : clampf(v.y / s, -1.0f, 1.0f); }
TR_DEV bool same_hemisphere(f3 a, f3 b) { return a.z * b.z > 0.0f; }
TR_DEV bool lobe_matches(uint32_t type, uint32_t flags) { return (type & ~flags) == 0u; }   // is_subset

// fresnel.rs:10-14,50-68 with eta_i = 1
TR_DEV float fresnel_dielectric(float eta_t, float cos_i) {
    float ci = clampf(cos_i, -1.0f, 1.0f);
    float ei = ci > 0.0f ? 1.0f : eta_t, et = ci > 0.0f ? eta_t : 1.0f;
    float sin_t = ei / et * sqrtf(fmaxf(0.0f, 1.0f - ci * ci));
    if (sin_t >= 1.0f) return 1.0f;
    float ct = sqrtf(fmaxf(0.0f, 1.0f - sin_t * sin_t));
    float c = fabsf(ci);
    float r_par = (et * c - ei * ct) / (et * c + ei * ct);
    float r_perp = (ei * c - et * ct) / (ei * c + et * ct);
    return 0.5f * (r_par * r_par + r_perp * r_perp);
}
// fresnel.rs:16-25,86-88
TR_DEV f3 fresnel_conductor(f3 eta, f3 k, float cos_i) {
    float ci = fabsf(cos_i);
    f3 one = mk(1.0f, 1.0f, 1.0f);
    f3 a = (eta * eta + k * k) * ci * ci;
    f3 r_par = (a - eta * ci * 2.0f + one) / (a + eta * ci * 2.0f + one);
    f3 b = eta * eta + k * k;
    f3 c2 = mk(ci * ci, ci * ci, ci * ci);
    f3 r_perp = (b - eta * ci * 2.0f + c2) / (b + eta * ci * 2.0f + c2);
    return (r_par + r_perp) * 0.5f;
}

// microfacet/beckmann.rs:26-64
TR_DEV float beckmann_d(float width, f3 w_h) {
    float tan_sqr = tan_theta_sqr(w_h);
    if (isinf(tan_sqr)) return 0.0f;
    float c2 = cos_theta_sqr(w_h);
    float cos_theta_4 = c2 * c2, width_sqr = width * width;
    return lm_exp(-tan_sqr / width_sqr) / (kPi * width_sqr * cos_theta_4);
}
TR_DEV f3 beckmann_sample(float width, float u0, float u1) {
    float log_sample = lm_log(1.0f - u0);
    if (isinf(log_sample)) log_sample = 0.0f;
    float tan_theta_sqr_v = -(width * width) * log_sample;
    float phi = 2.0f * kPi * u1;
    float cos_t = 1.0f / sqrtf(1.0f + tan_theta_sqr_v);
    float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
    float sn, cs;
    lm_sincos(phi, sn, cs);
    return mk(sin_t * cs, sin_t * sn, cos_t);   // linalg::spherical_dir
}
TR_DEV float beckmann_pdf(float width, f3 w_h) { return fabsf(w_h.z) * beckmann_d(width, w_h); }
TR_DEV float beckmann_g1(float width, f3 v) {
    float a = 1.0f / (width * fabsf(tan_theta(v)));
    if (a < 1.6f) {
        float a_sqr = a * a;
        return (3.535f * a + 2.181f * a_sqr) / (1.0f + 2.276f * a + 2.577f * a_sqr);
    }
    return 1.0f;
}

// microfacet/ggx.rs:27-57 (powf(x, 2.0) as a product: exact; powf(x, 4.0) is libm's powf as in ggx.rs:30 -- one rounding of x^4, not the two of (x * x) * (x * x))
TR_DEV float ggx_d(float width, f3 w_h) {
    if (cos_theta(w_h) > 0.0f) {
        float width_sqr = width * width;
        float c = cos_theta(w_h);
        float t = tan_theta(w_h);
        float s = width_sqr + t * t;
        float denom = kPi * powf(c, 4.0f) * (s * s);
        return width_sqr / denom;
    }
    return 0.0f;
}
TR_DEV f3 ggx_sample(float width, float u0, float u1) {
    float t = width * sqrtf(u0) / sqrtf(1.0f - u0);
    float tan_theta_sqr_v = t * t;
    float cos_t = 1.0f / sqrtf(1.0f + tan_theta_sqr_v);
    float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
    float phi = 2.0f * kPi * u1;
    float sn, cs;
    lm_sincos(phi, sn, cs);
    return mk(sin_t * cs, sin_t * sn, cos_t);
}
TR_DEV float ggx_g1(float width, f3 v) {
    float t = width * fabsf(tan_theta(v));
    return 2.0f / (1.0f + sqrtf(1.0f + t * t));
}
// the lobe's MicrofacetDistribution: Beckmann (what every material of the reference builds) or GGX (DevLobe::ob != 0 on the
// microfacet lobes; Oren-Nayar's B lives in the same field of its own lobe kind)
// (compiled into the kernels built with FEAT_TEX only -- the instantiation for scenes that use the rarely needed extras: image
// textures and GGX -- so that the common kernels carry neither the test nor the code)
#define TR_GGX(FEAT, l) (((FEAT) & 8) != 0 && (l).ob != 0.0f)
template <int FEAT> TR_DEV float mf_d(const Lobe& l, f3 w_h) { return TR_GGX(FEAT, l) ? ggx_d(l.width, w_h) : beckmann_d(l.width, w_h); }
template <int FEAT> TR_DEV f3 mf_sample(const Lobe& l, float u0, float u1) { return TR_GGX(FEAT, l) ? ggx_sample(l.width, u0, u1) : beckmann_sample(l.width, u0, u1); }
template <int FEAT> TR_DEV float mf_pdf(const Lobe& l, f3 w_h) { return fabsf(w_h.z) * mf_d<FEAT>(l, w_h); }
template <int FEAT> TR_DEV float mf_g1(const Lobe& l, f3 v) { return TR_GGX(FEAT, l) ? ggx_g1(l.width, v) : beckmann_g1(l.width, v); }

// microfacet_transmission.rs:34-60 with Dielectric(1, eta_t)
TR_DEV void mt_eta(float eta_t, f3 w_o, float& e0, float& e1) {
    if (cos_theta(w_o) > 0.0f) { e0 = 1.0f; e1 = eta_t; } else { e0 = eta_t; e1 = 1.0f; }
}
TR_DEV float mt_jacobian(f3 w_o, f3 w_i, f3 w_h, float e0, float e1) {
    float wi_dot_h = dot(w_i, w_h), wo_dot_h = dot(w_o, w_h);
    float s = e1 * wi_dot_h + e0 * wo_dot_h;
    float denom = s * s;
    if (denom != 0.0f) return fabsf(e0 * e0 * fabsf(wo_dot_h) / denom);
    return 0.0f;
}
TR_DEV f3 mt_half_vector(f3 w_o, f3 w_i, float e0, float e1) { return normalized(-e1 * w_i - e0 * w_o); }

// bxdf/merl.rs:41-83
TR_DEV uint32_t merl_index(float val, float max, uint32_t n_vals) {
    float f = val / max * (float)n_vals;
    uint32_t idx = f > 0.0f ? (f >= 4294967040.0f ? 0xffffffffu : (uint32_t)f) : 0u;   // saturating `as usize`
    return idx > n_vals - 1u ? n_vals - 1u : idx;
}
TR_DEV f3 merl_eval(const float* __restrict__ brdf, f3 w_oi, f3 w_ii) {
    f3 w_i = w_ii;
    f3 w_h = w_oi + w_i;
    if (w_h.z < 0.0f) { w_i = -w_i; w_h = -w_h; }
    if (length_sqr(w_h) == 0.0f) return mk(0.0f, 0.0f, 0.0f);
    w_h = normalized(w_h);
    float theta_h = lm_acos(clampf(w_h.z, -1.0f, 1.0f));
    float cos_phi_h = cos_phi(w_h), sin_phi_h = sin_phi(w_h);
    float cos_theta_h = cos_theta(w_h), sin_theta_h = sin_theta(w_h);
    f3 w_hx = mk(cos_phi_h * cos_theta_h, sin_phi_h * cos_theta_h, -sin_theta_h);
    f3 w_hy = mk(-sin_phi_h, cos_phi_h, 0.0f);
    f3 w_d = mk(dot(w_i, w_hx), dot(w_i, w_hy), dot(w_i, w_h));
    float theta_d = lm_acos(clampf(w_d.z, -1.0f, 1.0f));
    float phi_d = lm_atan2(w_d.y, w_d.x);
    if (phi_d < 0.0f) phi_d = phi_d + kPi * 2.0f;
    if (phi_d > kPi) phi_d = phi_d - kPi;   // quirk Q10
    uint32_t th = merl_index(sqrtf(fmaxf(0.0f, 2.0f * theta_h / kPi)), 1.0f, 90u);
    uint32_t td = merl_index(theta_d, kPi / 2.0f, 90u);
    uint32_t pd = merl_index(phi_d, kPi, 180u);
    uint32_t i = pd + 180u * (td + th * 90u);
    return mk(brdf[3u * i], brdf[3u * i + 1u], brdf[3u * i + 2u]);
}

// ---- per-lobe BxDF::{eval,pdf,sample} (shading space) --------------------------------------
// FEAT: which of the two register-hungry lobe kinds the scene's materials contain. The kernels that hold BSDF code are
// instantiated per feature set and tray_scene_create picks the smallest one that covers the scene: leaving the MERL table
// lookup and the microfacet-transmission code out of the single eval / pdf site cuts the tile kernel's scratch from 740 to
// 592 B per lane (cornell_box 454 -> 484, smallpt 385 -> 421 Msamples/s at 64 spp), leaving out the specular lobes and the
// conductor Fresnel term as well to 508 B (cornell_box 545).
enum : int { FEAT_NONE = 0, FEAT_MERL = 1, FEAT_MF_TRANS = 2, FEAT_SPEC = 4, FEAT_ALL = 7,   // FEAT_SPEC: specular lobes and conductor Fresnel
             FEAT_TEX = 8 };   // image textures: materials whose lobes are lowered per hit (dev_tex.h); always together with FEAT_ALL
// KM: bit per lobe kind (1 << LB_*) that can occur at all. The kind-pure shading kernels of the wavefront schedule (wavefront.h:
// k_wf_query_kind, fed by the material sort of k_wf_begin) are instantiated with the lobes of ONE material kind, so every other
// case of the switches below is dead code there; everywhere else KM_ALL keeps them all.
enum : uint32_t { KM_ALL = 0x1ffu };
#define LOBE_ON(k) ((KM >> (k)) & 1u)
// lobes a material kind (TRAY_MAT_*) lowers to (lower_material below), and the FEAT set its code needs
TR_DEV constexpr uint32_t km_of_material(int mk) {
    return mk == TRAY_MAT_MATTE ? (1u << LB_LAMBERTIAN) | (1u << LB_OREN_NAYAR)
         : mk == TRAY_MAT_PLASTIC ? (1u << LB_LAMBERTIAN) | (1u << LB_TS_DIEL)
         : mk == TRAY_MAT_METAL ? (1u << LB_TS_COND)
         : mk == TRAY_MAT_GLASS ? (1u << LB_SPEC_REFL_DIEL) | (1u << LB_SPEC_TRANS)
         : mk == TRAY_MAT_ROUGH_GLASS ? (1u << LB_TS_DIEL) | (1u << LB_MF_TRANS)
         : mk == TRAY_MAT_SPECULAR_METAL ? (1u << LB_SPEC_REFL_COND)
         : (1u << LB_MERL);
}
TR_DEV constexpr int feat_of_material(int mk) {
    return mk == TRAY_MAT_METAL || mk == TRAY_MAT_GLASS || mk == TRAY_MAT_SPECULAR_METAL ? FEAT_SPEC
         : mk == TRAY_MAT_ROUGH_GLASS ? FEAT_MF_TRANS : mk == TRAY_MAT_MERL ? FEAT_MERL : FEAT_NONE;
}
template <int FEAT, uint32_t KM = KM_ALL>
TR_DEV f3 lobe_eval(const Bsdf& b, const Lobe& l, f3 w_o, f3 w_i) {
    switch (l.kind) {
        case LB_LAMBERTIAN: return LOBE_ON(LB_LAMBERTIAN) ? l.color * kInvPi : mk(0.0f, 0.0f, 0.0f);
        case LB_OREN_NAYAR: {
            if (!LOBE_ON(LB_OREN_NAYAR)) return mk(0.0f, 0.0f, 0.0f);
            float sin_o = sin_theta(w_o), sin_i = sin_theta(w_i);
            float max_cos = 0.0f;
            if (sin_i > 1e-4f && sin_o > 1e-4f)
                max_cos = fmaxf(0.0f, cos_phi(w_i) * cos_phi(w_o) + sin_phi(w_i) * sin_phi(w_o));
            float sin_alpha, tan_beta;
            if (fabsf(cos_theta(w_i)) > fabsf(cos_theta(w_o))) { sin_alpha = sin_o; tan_beta = sin_i / fabsf(cos_theta(w_i)); }
            else { sin_alpha = sin_i; tan_beta = sin_o / fabsf(cos_theta(w_o)); }
            return l.color * kInvPi * (l.width + l.ob * max_cos * sin_alpha * tan_beta);
        }
        case LB_TS_DIEL:
        case LB_TS_COND: {
            if (!LOBE_ON(LB_TS_DIEL) && !LOBE_ON(LB_TS_COND)) return mk(0.0f, 0.0f, 0.0f);
            float cos_to = fabsf(cos_theta(w_o)), cos_ti = fabsf(cos_theta(w_i));
            if (cos_to == 0.0f || cos_ti == 0.0f) return mk(0.0f, 0.0f, 0.0f);
            f3 w_h = w_i + w_o;
            if (w_h.x == 0.0f && w_h.y == 0.0f && w_h.z == 0.0f) return mk(0.0f, 0.0f, 0.0f);
            w_h = normalized(w_h);
            float d = mf_d<FEAT>(l, w_h);
            f3 f;
            if (!(FEAT & FEAT_SPEC) || !LOBE_ON(LB_TS_COND) || (LOBE_ON(LB_TS_DIEL) && l.kind == LB_TS_DIEL)) { float fr = fresnel_dielectric(l.eta_t, dot(w_i, w_h)); f = mk(fr, fr, fr); }
            else f = fresnel_conductor(mk(b.mat->eta[0], b.mat->eta[1], b.mat->eta[2]), mk(b.mat->k[0], b.mat->k[1], b.mat->k[2]), dot(w_i, w_h));
            float g = mf_g1<FEAT>(l, w_i) * mf_g1<FEAT>(l, w_o);
            return l.color * f * d * g / (4.0f * cos_ti * cos_to);
        }
        case LB_MF_TRANS: {
            if (!(FEAT & FEAT_MF_TRANS) || !LOBE_ON(LB_MF_TRANS)) return mk(0.0f, 0.0f, 0.0f);   // no such lobe in this scene (checked by the host)
            if (same_hemisphere(w_o, w_i)) return mk(0.0f, 0.0f, 0.0f);
            float cos_to = cos_theta(w_o), cos_ti = cos_theta(w_i);
            if (cos_to == 0.0f || cos_ti == 0.0f) return mk(0.0f, 0.0f, 0.0f);
            float e0, e1;
            mt_eta(l.eta_t, w_o, e0, e1);
            f3 w_h = mt_half_vector(w_o, w_i, e0, e1);
            float d = mf_d<FEAT>(l, w_h);
            float fr = 1.0f - fresnel_dielectric(l.eta_t, dot(w_i, w_h));
            float g = mf_g1<FEAT>(l, w_i) * mf_g1<FEAT>(l, w_o);
            float wi_dot_h = dot(w_i, w_h);
            float jac = mt_jacobian(w_o, w_i, w_h, e0, e1);
            f3 f = mk(fr, fr, fr);
            return l.color * (fabsf(wi_dot_h) / (fabsf(w_i.z) * fabsf(w_o.z))) * (f * g * d) * jac;
        }
        case LB_MERL: return ((FEAT & FEAT_MERL) && LOBE_ON(LB_MERL)) ? merl_eval(b.merl_data + b.mat->merl_offset, w_o, w_i) : mk(0.0f, 0.0f, 0.0f);
        default: return mk(0.0f, 0.0f, 0.0f);   // specular lobes evaluate to black
    }
}
template <int FEAT, uint32_t KM = KM_ALL>
TR_DEV float lobe_pdf(const Lobe& l, f3 w_o, f3 w_i) {
    if ((LOBE_ON(LB_TS_DIEL) || LOBE_ON(LB_TS_COND)) && (l.kind == LB_TS_DIEL || l.kind == LB_TS_COND)) {
        if (!same_hemisphere(w_o, w_i)) return 0.0f;
        f3 w_h = normalized(w_o + w_i);
        float jac = 1.0f / (4.0f * fabsf(dot(w_o, w_h)));
        return mf_pdf<FEAT>(l, w_h) * jac;
    }
    if ((FEAT & FEAT_MF_TRANS) && LOBE_ON(LB_MF_TRANS) && l.kind == LB_MF_TRANS) {
        if (same_hemisphere(w_o, w_i)) return 0.0f;
        float e0, e1;
        mt_eta(l.eta_t, w_o, e0, e1);
        f3 w_h = mt_half_vector(w_o, w_i, e0, e1);
        return mf_pdf<FEAT>(l, w_h) * mt_jacobian(w_o, w_i, w_h, e0, e1);
    }
    return same_hemisphere(w_o, w_i) ? fabsf(cos_theta(w_i)) * kInvPi : 0.0f;   // bxdf/mod.rs:112-121
}
// BxDF::sample. For specular lobes returns f; for the others only the direction and the lobe's own pdf
// are produced: BSDF::sample (bsdf.rs:103-109) replaces f by BSDF::eval for every non-specular lobe.
// want_pdf = false: the caller replaces the lobe's own pdf by BSDF::pdf anyway (bsdf.rs:103-105: a non-specular lobe of a BSDF with more
// than one matching lobe), so it is not computed (pdf = 0); the direction, and whether one exists at all, do not depend on it.
template <int FEAT, uint32_t KM = KM_ALL>
TR_DEV f3 lobe_sample(const Bsdf& b, const Lobe& l, f3 w_o, float u0, float u1, f3& w_i, float& pdf, bool want_pdf = true) {
    const f3 zero = mk(0.0f, 0.0f, 0.0f);
    switch (l.kind) {
        case LB_SPEC_REFL_DIEL:
        case LB_SPEC_REFL_COND: {
            if (!(FEAT & FEAT_SPEC) || (!LOBE_ON(LB_SPEC_REFL_DIEL) && !LOBE_ON(LB_SPEC_REFL_COND))) { w_i = zero; pdf = 0.0f; return zero; }   // no such lobe in this scene (checked by the host)
            w_i = mk(-w_o.x, -w_o.y, w_o.z);
            if (w_i.z != 0.0f) {
                f3 f;
                if (!LOBE_ON(LB_SPEC_REFL_COND) || (LOBE_ON(LB_SPEC_REFL_DIEL) && l.kind == LB_SPEC_REFL_DIEL)) { float fr = fresnel_dielectric(l.eta_t, cos_theta(w_o)); f = mk(fr, fr, fr); }
                else f = fresnel_conductor(mk(b.mat->eta[0], b.mat->eta[1], b.mat->eta[2]), mk(b.mat->k[0], b.mat->k[1], b.mat->k[2]), cos_theta(w_o));
                pdf = 1.0f;
                return f * l.color / fabsf(cos_theta(w_i));
            }
            pdf = 0.0f;
            return zero;
        }
        case LB_SPEC_TRANS: {
            if (!(FEAT & FEAT_SPEC) || !LOBE_ON(LB_SPEC_TRANS)) { w_i = zero; pdf = 0.0f; return zero; }
            bool entering = cos_theta(w_o) > 0.0f;
            float ei = entering ? 1.0f : l.eta_t, et = entering ? l.eta_t : 1.0f;
            f3 n = entering ? mk(0.0f, 0.0f, 1.0f) : mk(0.0f, 0.0f, -1.0f);
            if (refract(w_o, n, ei / et, w_i)) {
                float fr = 1.0f - fresnel_dielectric(l.eta_t, cos_theta(w_i));
                pdf = 1.0f;
                return mk(fr, fr, fr) * l.color / fabsf(cos_theta(w_i));
            }
            w_i = zero; pdf = 0.0f;
            return zero;
        }
        case LB_TS_DIEL:
        case LB_TS_COND: {
            if (!LOBE_ON(LB_TS_DIEL) && !LOBE_ON(LB_TS_COND)) { w_i = zero; pdf = 0.0f; return zero; }
            if (w_o.z == 0.0f) { w_i = zero; pdf = 0.0f; return zero; }
            f3 w_h = mf_sample<FEAT>(l, u0, u1);
            if (!same_hemisphere(w_o, w_h)) w_h = -w_h;
            w_i = reflect(w_o, w_h);
            if (!same_hemisphere(w_o, w_i)) { w_i = zero; pdf = 0.0f; return zero; }
            pdf = want_pdf ? lobe_pdf<FEAT, KM>(l, w_o, w_i) : 0.0f;
            return zero;
        }
        case LB_MF_TRANS: {
            if (!(FEAT & FEAT_MF_TRANS) || !LOBE_ON(LB_MF_TRANS)) { w_i = zero; pdf = 0.0f; return zero; }
            f3 w_h = mf_sample<FEAT>(l, u0, u1);
            if (!same_hemisphere(w_o, w_h)) w_h = -w_h;
            float e0, e1;
            mt_eta(l.eta_t, w_o, e0, e1);
            f3 wi;
            if (refract(w_o, w_h, e0 / e1, wi)) {
                if (same_hemisphere(w_o, wi)) { w_i = zero; pdf = 0.0f; return zero; }
                w_i = wi;
                pdf = want_pdf ? lobe_pdf<FEAT, KM>(l, w_o, w_i) : 0.0f;
                return zero;
            }
            w_i = zero; pdf = 0.0f;
            return zero;
        }
        default: {   // cosine hemisphere (bxdf/mod.rs:102-109)
            if (!LOBE_ON(LB_LAMBERTIAN) && !LOBE_ON(LB_OREN_NAYAR) && !LOBE_ON(LB_MERL)) { w_i = zero; pdf = 0.0f; return zero; }
            w_i = cos_sample_hemisphere(u0, u1);
            if (w_o.z < 0.0f) w_i.z *= -1.0f;
            pdf = want_pdf ? lobe_pdf<FEAT, KM>(l, w_o, w_i) : 0.0f;
            return zero;
        }
    }
}

// ---- BSDF (world space) ------------------------------------------------------------------------
#define TR_BITAN(b) cross((b).tan, (b).n)
TR_DEV f3 to_shading(const Bsdf& b, f3 v) { return mk(dot(v, TR_BITAN(b)), dot(v, b.tan), dot(v, b.n)); }   // bsdf.rs:52-55
TR_DEV f3 from_shading(const Bsdf& b, f3 v) {   // bsdf.rs:57-61
    const f3 bt = TR_BITAN(b);
    return mk(bt.x * v.x + b.tan.x * v.y + b.n.x * v.z, bt.y * v.x + b.tan.y * v.y + b.n.y * v.z,
              bt.z * v.x + b.tan.z * v.y + b.n.z * v.z);
}
// The *_sh variants take w_o / w_i already in (normalised) shading space: BSDF::eval, ::pdf and ::sample each start with the
// same to_shading + normalized of the same vectors (bsdf.rs:67-68,86,115-116); the vertex step computes them once.
template <int FEAT, uint32_t KM = KM_ALL>
TR_DEV f3 bsdf_eval_sh(const Bsdf& b, f3 w_o, f3 w_i, uint32_t flags) {   // bsdf.rs:66-79
    if (w_o.z * w_i.z > 0.0f) flags &= ~(uint32_t)BX_TRANSMISSION; else flags &= ~(uint32_t)BX_REFLECTION;
    f3 sum = mk(0.0f, 0.0f, 0.0f);
    const int n = (int)b.mat->n_lobes;
#pragma nounroll
    for (int i = 0; i < n; ++i) {
        Lobe l = load_lobe(b.mat, i);
        if (lobe_matches(l.type, flags)) sum = sum + lobe_eval<FEAT, KM>(b, l, w_o, w_i);
    }
    return sum;
}
template <int FEAT>
TR_DEV f3 bsdf_eval(const Bsdf& b, f3 wo_world, f3 wi_world, uint32_t flags) {
    return bsdf_eval_sh<FEAT>(b, normalized(to_shading(b, wo_world)), normalized(to_shading(b, wi_world)), flags);
}
template <int FEAT, uint32_t KM = KM_ALL>
TR_DEV float bsdf_pdf_sh(const Bsdf& b, f3 w_o, f3 w_i, uint32_t flags) {   // bsdf.rs:114-125
    float pdf_val = 0.0f;
    int n_comps = 0;
    const int n = (int)b.mat->n_lobes;
#pragma nounroll
    for (int i = 0; i < n; ++i) {
        Lobe l = load_lobe(b.mat, i);
        if (lobe_matches(l.type, flags)) { pdf_val = pdf_val + lobe_pdf<FEAT, KM>(l, w_o, w_i); ++n_comps; }
    }
    return n_comps > 0 ? pdf_val / (float)n_comps : 0.0f;
}
template <int FEAT>
TR_DEV float bsdf_pdf(const Bsdf& b, f3 wo_world, f3 wi_world, uint32_t flags) {
    return bsdf_pdf_sh<FEAT>(b, normalized(to_shading(b, wo_world)), normalized(to_shading(b, wi_world)), flags);
}
// BSDF::eval and BSDF::pdf of the same (w_o, w_i) in one pass over the lobes (query_stage needs both for the light half and for every
// two-lobe material). A Torrance-Sparrow lobe's eval (torrance_sparrow.rs:40-57) and pdf (:72-81) both start from the half vector
// normalized(w_i + w_o) -- w_o + w_i in pdf(): the same three sums -- and the distribution's D(w_h): computed once here, every other
// operation as in lobe_eval / lobe_pdf, so f and pdf carry the bits of bsdf_eval_sh / bsdf_pdf_sh.
template <int FEAT, uint32_t KM = KM_ALL>
TR_DEV void bsdf_eval_pdf_sh(const Bsdf& b, f3 w_o, f3 w_i, uint32_t flags, bool need_eval, bool need_pdf, f3& f_out, float& pdf_out) {
    uint32_t fl_eval = flags;
    if (w_o.z * w_i.z > 0.0f) fl_eval &= ~(uint32_t)BX_TRANSMISSION; else fl_eval &= ~(uint32_t)BX_REFLECTION;
    f3 sum = mk(0.0f, 0.0f, 0.0f);
    float pdf_val = 0.0f;
    int n_comps = 0;
    const int n = (int)b.mat->n_lobes;
#pragma nounroll
    for (int i = 0; i < n; ++i) {
        const Lobe l = load_lobe(b.mat, i);
        const bool em = need_eval && lobe_matches(l.type, fl_eval), pm = need_pdf && lobe_matches(l.type, flags);
        if (!(em || pm)) continue;
        const bool ts = (LOBE_ON(LB_TS_DIEL) || LOBE_ON(LB_TS_COND)) && (l.kind == LB_TS_DIEL || l.kind == LB_TS_COND);
        if (ts && em && pm) {
            // lobe_eval's and lobe_pdf's own early outs first, then the shared half vector and D
            const float cos_to = fabsf(cos_theta(w_o)), cos_ti = fabsf(cos_theta(w_i));
            f3 w_h = w_i + w_o;
            const bool e_zero = cos_to == 0.0f || cos_ti == 0.0f || (w_h.x == 0.0f && w_h.y == 0.0f && w_h.z == 0.0f);
            const bool p_zero = !same_hemisphere(w_o, w_i);
            f3 e = mk(0.0f, 0.0f, 0.0f);
            float p = 0.0f;
            if (!(e_zero && p_zero)) {
                w_h = normalized(w_h);
                const float d = mf_d<FEAT>(l, w_h);
                if (!e_zero) {
                    f3 fr3;
                    if (!(FEAT & FEAT_SPEC) || !LOBE_ON(LB_TS_COND) || (LOBE_ON(LB_TS_DIEL) && l.kind == LB_TS_DIEL)) { float fr = fresnel_dielectric(l.eta_t, dot(w_i, w_h)); fr3 = mk(fr, fr, fr); }
                    else fr3 = fresnel_conductor(mk(b.mat->eta[0], b.mat->eta[1], b.mat->eta[2]), mk(b.mat->k[0], b.mat->k[1], b.mat->k[2]), dot(w_i, w_h));
                    const float g = mf_g1<FEAT>(l, w_i) * mf_g1<FEAT>(l, w_o);
                    e = l.color * fr3 * d * g / (4.0f * cos_ti * cos_to);
                }
                if (!p_zero) {
                    const float jac = 1.0f / (4.0f * fabsf(dot(w_o, w_h)));
                    p = fabsf(w_h.z) * d * jac;   // mf_pdf(l, w_h) * jac
                }
            }
            sum = sum + e;
            pdf_val = pdf_val + p; ++n_comps;
        } else {
            if (em) sum = sum + lobe_eval<FEAT, KM>(b, l, w_o, w_i);
            if (pm) { pdf_val = pdf_val + lobe_pdf<FEAT, KM>(l, w_o, w_i); ++n_comps; }
        }
    }
    if (need_eval) f_out = sum;
    if (need_pdf) pdf_out = n_comps > 0 ? pdf_val / (float)n_comps : 0.0f;
}
// Head of BSDF::sample (bsdf.rs:85-102): choose the lobe, sample its direction. Outputs the world
// direction, the lobe's own pdf, f for specular lobes, the sampled type bits (0 = nothing sampled)
// and which of BSDF::pdf / BSDF::eval the tail (bsdf.rs:103-109) still has to evaluate.
struct SampleHead {
    f3 wi_world, f;
    float pdf;
    uint32_t sampled_type;
    bool need_eval, need_pdf;
};
template <int FEAT, uint32_t KM = KM_ALL>
TR_DEV SampleHead bsdf_sample_head_sh(const Bsdf& b, f3 w_o, uint32_t flags, float u0, float u1, float one_d) {
    const f3 zero = mk(0.0f, 0.0f, 0.0f);
    SampleHead h;
    h.wi_world = zero; h.f = zero; h.pdf = 0.0f; h.sampled_type = 0u; h.need_eval = false; h.need_pdf = false;
    const uint32_t n_lobes = b.mat->n_lobes;
    bool m0 = n_lobes > 0u && lobe_matches(b.mat->lobe[0].type, flags);
    bool m1 = n_lobes > 1u && lobe_matches(b.mat->lobe[1].type, flags);
    int n_matching = (int)m0 + (int)m1;
    if (n_matching == 0) return h;
    float fc = one_d * (float)n_matching;
    int comp = fc > 0.0f ? (int)fc : 0;
    if (comp > n_matching - 1) comp = n_matching - 1;
    int li = (m0 && comp == 0) ? 0 : 1;   // matching_at(comp): the comp-th matching lobe
    const Lobe l = load_lobe(b.mat, li);
    f3 w_i;
    float pdf_v;
    f3 f = lobe_sample<FEAT, KM>(b, l, w_o, u0, u1, w_i, pdf_v, n_matching <= 1);   // (n_matching > 1: BSDF::pdf below replaces a non-specular lobe's own pdf)
    if (length_sqr(w_i) == 0.0f) return h;
    h.wi_world = normalized(from_shading(b, w_i));
    bool specular = (l.type & BX_SPECULAR) != 0u;
    h.f = f; h.pdf = pdf_v; h.sampled_type = l.type;
    h.need_pdf = !specular && n_matching > 1;
    h.need_eval = !specular;
    return h;
}
template <int FEAT>
TR_DEV SampleHead bsdf_sample_head(const Bsdf& b, f3 wo_world, uint32_t flags, float u0, float u1, float one_d) {
    return bsdf_sample_head_sh<FEAT>(b, normalized(to_shading(b, wo_world)), flags, u0, u1, one_d);
}
// Whole BSDF::sample (used by the BSDF debug kernel; the tile kernel shares one eval / pdf site
// between the light and the BSDF halves of estimate_direct and the path continuation, dev_integrator.h)
TR_DEV f3 bsdf_sample(const Bsdf& b, f3 wo_world, uint32_t flags, float u0, float u1, float one_d, f3& wi_world, float& pdf_out, uint32_t& sampled_type) {
    SampleHead h = bsdf_sample_head<FEAT_ALL | FEAT_TEX>(b, wo_world, flags, u0, u1, one_d);
    if (h.need_pdf) h.pdf = bsdf_pdf<FEAT_ALL | FEAT_TEX>(b, wo_world, h.wi_world, flags);
    if (h.need_eval) h.f = bsdf_eval<FEAT_ALL | FEAT_TEX>(b, wo_world, h.wi_world, flags);
    wi_world = h.wi_world; pdf_out = h.pdf; sampled_type = h.sampled_type;
    return h.f;
}

// BSDF::new (bsdf.rs:38-44; quirk Q8: tan is not renormalised); lobes come from the material table
TR_DEV Bsdf make_bsdf(const DevScene& sc, const Hit& hit) {
    Bsdf b;
    b.n = normalized(hit.n);
    f3 bt = normalized(hit.dp_du);
    b.tan = cross(b.n, bt);
    b.p = hit.p;
    b.u = hit.u; b.v = hit.v;
    b.mat = sc.materials + sc.instances[hit.inst].material_id;
    b.merl_data = sc.merl_data;
    return b;
}

// Material::bsdf for the seven materials (material/{matte,plastic,metal,glass,rough_glass,specular_metal,merl}.rs) from parameter
// VALUES: on the host once per material whose parameters are constants, on the device per hit for textured materials (dev_tex.h).
// f32 arithmetic in the reference's order; this translation unit is built with -ffp-contract=off for host and device.
#ifdef TR_HOST_EMU
#define TR_HD inline
#else
#define TR_HD __host__ __device__ inline
#endif
TR_HD void lower_values(DevMaterial& d, uint32_t kind, const float* c0, const float* c1, float f0, float f1, uint32_t microfacet = 0u) {
    const float mfd = microfacet == TRAY_MF_GGX ? 1.0f : 0.0f;   // DevLobe::ob of a microfacet lobe: 0 = Beckmann, 1 = GGX
    d.n_lobes = 0u;
    const float white[3] = {1.0f, 1.0f, 1.0f};
#define TR_ADD_LOBE(K, T, C, ETA, W, OB) do { DevLobe& l_ = d.lobe[d.n_lobes++]; l_.kind = (K); l_.type = (T); l_.color[0] = (C)[0]; l_.color[1] = (C)[1]; \
                                              l_.color[2] = (C)[2]; l_.eta_t = (ETA); l_.width = (W); l_.ob = (OB); } while (0)
#define TR_BLACK(C) ((C)[0] == 0.0f && (C)[1] == 0.0f && (C)[2] == 0.0f)
#define TR_BECKMANN(W) ((W) > 0.000001f ? (W) : 0.000001f)   /* Beckmann::new (f32::max) */
    for (int i = 0; i < 3; ++i) { d.eta[i] = c0[i]; d.k[i] = c1[i]; }
    d.eta[3] = 0.0f; d.k[3] = 0.0f;
    d.mat_kind = kind <= TRAY_MAT_MERL ? kind : (uint32_t)TRAY_MAT_MERL;
    switch (kind) {
        case TRAY_MAT_MATTE:   // matte.rs:52-65, oren_nayar.rs:26-34
            if (f0 == 0.0f) TR_ADD_LOBE(LB_LAMBERTIAN, BX_DIFFUSE | BX_REFLECTION, c0, 1.0f, 0.0f, 0.0f);
            else {
                float sigma = 3.14159265358979323846f / 180.0f * f0;
                sigma *= sigma;
                float a = 1.0f - 0.5f * sigma / (sigma + 0.33f);
                float bb = 0.45f * sigma / (sigma + 0.09f);
                TR_ADD_LOBE(LB_OREN_NAYAR, BX_DIFFUSE | BX_REFLECTION, c0, 1.0f, a, bb);
            }
            break;
        case TRAY_MAT_PLASTIC:   // plastic.rs:59-88
            if (!TR_BLACK(c0)) TR_ADD_LOBE(LB_LAMBERTIAN, BX_DIFFUSE | BX_REFLECTION, c0, 1.0f, 0.0f, 0.0f);
            if (!TR_BLACK(c1)) TR_ADD_LOBE(LB_TS_DIEL, BX_GLOSSY | BX_REFLECTION, c1, 1.5f, TR_BECKMANN(f0), mfd);
            break;
        case TRAY_MAT_METAL:   // metal.rs:56-67
            TR_ADD_LOBE(LB_TS_COND, BX_GLOSSY | BX_REFLECTION, white, 1.0f, TR_BECKMANN(f0), mfd);
            break;
        case TRAY_MAT_GLASS:   // glass.rs:51-78
            if (!TR_BLACK(c0)) TR_ADD_LOBE(LB_SPEC_REFL_DIEL, BX_SPECULAR | BX_REFLECTION, c0, f0, 0.0f, 0.0f);
            if (!TR_BLACK(c1)) TR_ADD_LOBE(LB_SPEC_TRANS, BX_SPECULAR | BX_TRANSMISSION, c1, f0, 0.0f, 0.0f);
            break;
        case TRAY_MAT_ROUGH_GLASS:   // rough_glass.rs:57-85
            if (!TR_BLACK(c0)) TR_ADD_LOBE(LB_TS_DIEL, BX_GLOSSY | BX_REFLECTION, c0, f0, TR_BECKMANN(f1), mfd);
            if (!TR_BLACK(c1)) TR_ADD_LOBE(LB_MF_TRANS, BX_GLOSSY | BX_TRANSMISSION, c1, f0, TR_BECKMANN(f1), mfd);
            break;
        case TRAY_MAT_SPECULAR_METAL:   // specular_metal.rs:49-58
            TR_ADD_LOBE(LB_SPEC_REFL_COND, BX_SPECULAR | BX_REFLECTION, white, 1.0f, 0.0f, 0.0f);
            break;
        default:   // TRAY_MAT_MERL, material/merl.rs:88-92
            TR_ADD_LOBE(LB_MERL, BX_GLOSSY | BX_REFLECTION, white, 1.0f, 0.0f, 0.0f);
            break;
    }
#undef TR_ADD_LOBE
#undef TR_BLACK
#undef TR_BECKMANN
}
// host: one DevMaterial per TrayMaterial (constants lowered here; textured ones keep their parameters for the per-hit lowering)
inline DevMaterial lower_material(const TrayMaterial& m, const TrayMerlTable* tables) {
    DevMaterial d;
    std::memset(&d, 0, sizeof d);
    lower_values(d, m.kind, m.c0, m.c1, m.f0, m.f1, m.microfacet);
    d.microfacet = m.microfacet;
    if (m.kind == TRAY_MAT_MERL) d.merl_offset = tables ? tables[m.table].offset : 0;
    d.tex_c0 = m.tex_c0; d.tex_c1 = m.tex_c1; d.tex_f0 = m.tex_f0; d.tex_f1 = m.tex_f1;
    d.textured = (m.tex_c0 != TRAY_NO_TEXTURE || m.tex_c1 != TRAY_NO_TEXTURE || m.tex_f0 != TRAY_NO_TEXTURE || m.tex_f1 != TRAY_NO_TEXTURE) ? 1u : 0u;
    for (int i = 0; i < 3; ++i) { d.c0[i] = m.c0[i]; d.c1[i] = m.c1[i]; }
    d.f0 = m.f0; d.f1 = m.f1;
    return d;
}

}  // namespace tr
