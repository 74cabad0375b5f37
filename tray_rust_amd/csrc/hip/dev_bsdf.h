// Device-side shading: the reference's open Material / BxDF trait objects lowered to a closed set
// of lobes held in registers (no arena, no dynamic dispatch).
//   Material::bsdf      material/{matte,plastic,metal,glass,rough_glass,specular_metal,merl}.rs
//   BSDF                bxdf/bsdf.rs:38-134         BxDF defaults  bxdf/mod.rs:92-166
//   lobes               bxdf/{lambertian,oren_nayar,specular_reflection,specular_transmission,
//                             torrance_sparrow,microfacet_transmission,merl}.rs, microfacet/beckmann.rs,
//                             fresnel.rs
// Colours are RGB only: the reference's alpha channel never reaches the film (render_target.rs:142-145).
#pragma once
#include "dev_geom.h"

namespace tr {

enum { BX_REFLECTION = 1, BX_TRANSMISSION = 2, BX_DIFFUSE = 4, BX_GLOSSY = 8, BX_SPECULAR = 16 };
enum { BX_ALL = 31, BX_NON_SPECULAR = 15 };   // BxDFType::all / non_specular (bxdf/mod.rs:64-82)

enum { LB_LAMBERTIAN = 0, LB_OREN_NAYAR, LB_SPEC_REFL_DIEL, LB_SPEC_REFL_COND, LB_SPEC_TRANS, LB_TS_DIEL, LB_TS_COND, LB_MF_TRANS, LB_MERL };

struct Lobe {
    uint32_t kind, type;
    f3 color;
    float eta_t;   // dielectric: Dielectric::new(1.0, eta_t)
    float width;   // Beckmann width | Oren-Nayar: a
    float ob;      // Oren-Nayar b
};

struct Bsdf {
    f3 p, n, ng, tan, bitan;
    const TrayMaterial* __restrict__ mat;
    const float* __restrict__ merl;
    int n_lobes;
    Lobe lobe[2];
};

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
    return expf(-tan_sqr / width_sqr) / (kPi * width_sqr * cos_theta_4);
}
TR_DEV f3 beckmann_sample(float width, float u0, float u1) {
    float log_sample = logf(1.0f - u0);
    if (isinf(log_sample)) log_sample = 0.0f;
    float tan_theta_sqr_v = -(width * width) * log_sample;
    float phi = 2.0f * kPi * u1;
    float cos_t = 1.0f / sqrtf(1.0f + tan_theta_sqr_v);
    float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
    return mk(sin_t * cosf(phi), sin_t * sinf(phi), cos_t);   // linalg::spherical_dir
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
    float theta_h = acosf(clampf(w_h.z, -1.0f, 1.0f));
    float cos_phi_h = cos_phi(w_h), sin_phi_h = sin_phi(w_h);
    float cos_theta_h = cos_theta(w_h), sin_theta_h = sin_theta(w_h);
    f3 w_hx = mk(cos_phi_h * cos_theta_h, sin_phi_h * cos_theta_h, -sin_theta_h);
    f3 w_hy = mk(-sin_phi_h, cos_phi_h, 0.0f);
    f3 w_d = mk(dot(w_i, w_hx), dot(w_i, w_hy), dot(w_i, w_h));
    float theta_d = acosf(clampf(w_d.z, -1.0f, 1.0f));
    float phi_d = atan2f(w_d.y, w_d.x);
    if (phi_d < 0.0f) phi_d = phi_d + kPi * 2.0f;
    if (phi_d > kPi) phi_d = phi_d - kPi;   // quirk Q10
    uint32_t th = merl_index(sqrtf(fmaxf(0.0f, 2.0f * theta_h / kPi)), 1.0f, 90u);
    uint32_t td = merl_index(theta_d, kPi / 2.0f, 90u);
    uint32_t pd = merl_index(phi_d, kPi, 180u);
    uint32_t i = pd + 180u * (td + th * 90u);
    return mk(brdf[3u * i], brdf[3u * i + 1u], brdf[3u * i + 2u]);
}

// ---- per-lobe BxDF::{eval,pdf,sample} (shading space) --------------------------------------
TR_DEV f3 lobe_eval(const Bsdf& b, const Lobe& l, f3 w_o, f3 w_i) {
    switch (l.kind) {
        case LB_LAMBERTIAN: return l.color * kInvPi;
        case LB_OREN_NAYAR: {
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
            float cos_to = fabsf(cos_theta(w_o)), cos_ti = fabsf(cos_theta(w_i));
            if (cos_to == 0.0f || cos_ti == 0.0f) return mk(0.0f, 0.0f, 0.0f);
            f3 w_h = w_i + w_o;
            if (w_h.x == 0.0f && w_h.y == 0.0f && w_h.z == 0.0f) return mk(0.0f, 0.0f, 0.0f);
            w_h = normalized(w_h);
            float d = beckmann_d(l.width, w_h);
            f3 f;
            if (l.kind == LB_TS_DIEL) { float fr = fresnel_dielectric(l.eta_t, dot(w_i, w_h)); f = mk(fr, fr, fr); }
            else f = fresnel_conductor(mk(b.mat->c0[0], b.mat->c0[1], b.mat->c0[2]), mk(b.mat->c1[0], b.mat->c1[1], b.mat->c1[2]), dot(w_i, w_h));
            float g = beckmann_g1(l.width, w_i) * beckmann_g1(l.width, w_o);
            return l.color * f * d * g / (4.0f * cos_ti * cos_to);
        }
        case LB_MF_TRANS: {
            if (same_hemisphere(w_o, w_i)) return mk(0.0f, 0.0f, 0.0f);
            float cos_to = cos_theta(w_o), cos_ti = cos_theta(w_i);
            if (cos_to == 0.0f || cos_ti == 0.0f) return mk(0.0f, 0.0f, 0.0f);
            float e0, e1;
            mt_eta(l.eta_t, w_o, e0, e1);
            f3 w_h = mt_half_vector(w_o, w_i, e0, e1);
            float d = beckmann_d(l.width, w_h);
            float fr = 1.0f - fresnel_dielectric(l.eta_t, dot(w_i, w_h));
            float g = beckmann_g1(l.width, w_i) * beckmann_g1(l.width, w_o);
            float wi_dot_h = dot(w_i, w_h);
            float jac = mt_jacobian(w_o, w_i, w_h, e0, e1);
            f3 f = mk(fr, fr, fr);
            return l.color * (fabsf(wi_dot_h) / (fabsf(w_i.z) * fabsf(w_o.z))) * (f * g * d) * jac;
        }
        case LB_MERL: return merl_eval(b.merl, w_o, w_i);
        default: return mk(0.0f, 0.0f, 0.0f);   // specular lobes evaluate to black
    }
}
TR_DEV float lobe_pdf(const Lobe& l, f3 w_o, f3 w_i) {
    if (l.kind == LB_TS_DIEL || l.kind == LB_TS_COND) {
        if (!same_hemisphere(w_o, w_i)) return 0.0f;
        f3 w_h = normalized(w_o + w_i);
        float jac = 1.0f / (4.0f * fabsf(dot(w_o, w_h)));
        return beckmann_pdf(l.width, w_h) * jac;
    }
    if (l.kind == LB_MF_TRANS) {
        if (same_hemisphere(w_o, w_i)) return 0.0f;
        float e0, e1;
        mt_eta(l.eta_t, w_o, e0, e1);
        f3 w_h = mt_half_vector(w_o, w_i, e0, e1);
        return beckmann_pdf(l.width, w_h) * mt_jacobian(w_o, w_i, w_h, e0, e1);
    }
    return same_hemisphere(w_o, w_i) ? fabsf(cos_theta(w_i)) * kInvPi : 0.0f;   // bxdf/mod.rs:112-121
}
TR_DEV f3 lobe_sample(const Bsdf& b, const Lobe& l, f3 w_o, float u0, float u1, f3& w_i, float& pdf) {
    const f3 zero = mk(0.0f, 0.0f, 0.0f);
    switch (l.kind) {
        case LB_SPEC_REFL_DIEL:
        case LB_SPEC_REFL_COND: {
            w_i = mk(-w_o.x, -w_o.y, w_o.z);
            if (w_i.z != 0.0f) {
                f3 f;
                if (l.kind == LB_SPEC_REFL_DIEL) { float fr = fresnel_dielectric(l.eta_t, cos_theta(w_o)); f = mk(fr, fr, fr); }
                else f = fresnel_conductor(mk(b.mat->c0[0], b.mat->c0[1], b.mat->c0[2]), mk(b.mat->c1[0], b.mat->c1[1], b.mat->c1[2]), cos_theta(w_o));
                pdf = 1.0f;
                return f * l.color / fabsf(cos_theta(w_i));
            }
            pdf = 0.0f;
            return zero;
        }
        case LB_SPEC_TRANS: {
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
            if (w_o.z == 0.0f) { w_i = zero; pdf = 0.0f; return zero; }
            f3 w_h = beckmann_sample(l.width, u0, u1);
            if (!same_hemisphere(w_o, w_h)) w_h = -w_h;
            w_i = reflect(w_o, w_h);
            if (!same_hemisphere(w_o, w_i)) { w_i = zero; pdf = 0.0f; return zero; }
            pdf = lobe_pdf(l, w_o, w_i);
            return lobe_eval(b, l, w_o, w_i);
        }
        case LB_MF_TRANS: {
            f3 w_h = beckmann_sample(l.width, u0, u1);
            if (!same_hemisphere(w_o, w_h)) w_h = -w_h;
            float e0, e1;
            mt_eta(l.eta_t, w_o, e0, e1);
            f3 wi;
            if (refract(w_o, w_h, e0 / e1, wi)) {
                if (same_hemisphere(w_o, wi)) { w_i = zero; pdf = 0.0f; return zero; }
                w_i = wi;
                pdf = lobe_pdf(l, w_o, w_i);
                return lobe_eval(b, l, w_o, w_i);
            }
            w_i = zero; pdf = 0.0f;
            return zero;
        }
        default: {   // cosine hemisphere (bxdf/mod.rs:102-109)
            w_i = cos_sample_hemisphere(u0, u1);
            if (w_o.z < 0.0f) w_i.z *= -1.0f;
            pdf = lobe_pdf(l, w_o, w_i);
            return lobe_eval(b, l, w_o, w_i);
        }
    }
}

// ---- BSDF (world space) ------------------------------------------------------------------------
TR_DEV f3 to_shading(const Bsdf& b, f3 v) { return mk(dot(v, b.bitan), dot(v, b.tan), dot(v, b.n)); }   // bsdf.rs:52-55
TR_DEV f3 from_shading(const Bsdf& b, f3 v) {   // bsdf.rs:57-61
    return mk(b.bitan.x * v.x + b.tan.x * v.y + b.n.x * v.z, b.bitan.y * v.x + b.tan.y * v.y + b.n.y * v.z,
              b.bitan.z * v.x + b.tan.z * v.y + b.n.z * v.z);
}
TR_DEV f3 bsdf_eval(const Bsdf& b, f3 wo_world, f3 wi_world, uint32_t flags) {   // bsdf.rs:66-79
    f3 w_o = normalized(to_shading(b, wo_world)), w_i = normalized(to_shading(b, wi_world));
    if (w_o.z * w_i.z > 0.0f) flags &= ~(uint32_t)BX_TRANSMISSION; else flags &= ~(uint32_t)BX_REFLECTION;
    f3 sum = mk(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < 2; ++i)
        if (i < b.n_lobes && lobe_matches(b.lobe[i].type, flags)) sum = sum + lobe_eval(b, b.lobe[i], w_o, w_i);
    return sum;
}
TR_DEV float bsdf_pdf(const Bsdf& b, f3 wo_world, f3 wi_world, uint32_t flags) {   // bsdf.rs:114-125
    f3 w_o = normalized(to_shading(b, wo_world)), w_i = normalized(to_shading(b, wi_world));
    float pdf_val = 0.0f;
    int n_comps = 0;
    for (int i = 0; i < 2; ++i)
        if (i < b.n_lobes && lobe_matches(b.lobe[i].type, flags)) { pdf_val = pdf_val + lobe_pdf(b.lobe[i], w_o, w_i); ++n_comps; }
    return n_comps > 0 ? pdf_val / (float)n_comps : 0.0f;
}
// bsdf.rs:85-111; returns f, writes wi_world, pdf, sampled lobe type bits (0 = nothing sampled)
TR_DEV f3 bsdf_sample(const Bsdf& b, f3 wo_world, uint32_t flags, float u0, float u1, float one_d, f3& wi_world, float& pdf_out, uint32_t& sampled_type) {
    const f3 zero = mk(0.0f, 0.0f, 0.0f);
    bool m0 = b.n_lobes > 0 && lobe_matches(b.lobe[0].type, flags);
    bool m1 = b.n_lobes > 1 && lobe_matches(b.lobe[1].type, flags);
    int n_matching = (int)m0 + (int)m1;
    if (n_matching == 0) { wi_world = zero; pdf_out = 0.0f; sampled_type = 0u; return zero; }
    float fc = one_d * (float)n_matching;
    int comp = fc > 0.0f ? (int)fc : 0;
    if (comp > n_matching - 1) comp = n_matching - 1;
    // matching_at(comp): the comp-th matching lobe
    int li = (m0 && comp == 0) ? 0 : 1;
    const Lobe l = b.lobe[li];
    f3 w_o = normalized(to_shading(b, wo_world));
    f3 w_i;
    float pdf_v;
    f3 f = lobe_sample(b, l, w_o, u0, u1, w_i, pdf_v);
    if (length_sqr(w_i) == 0.0f) { wi_world = zero; pdf_out = 0.0f; sampled_type = 0u; return zero; }
    wi_world = normalized(from_shading(b, w_i));
    bool specular = (l.type & BX_SPECULAR) != 0u;
    if (!specular && n_matching > 1) pdf_v = bsdf_pdf(b, wo_world, wi_world, flags);
    if (!specular) f = bsdf_eval(b, wo_world, wi_world, flags);
    pdf_out = pdf_v;
    sampled_type = l.type;
    return f;
}

// Material::bsdf + BSDF::new (bsdf.rs:38-44; quirk Q8: tan is not renormalised)
TR_DEV Bsdf make_bsdf(const DevScene& sc, const Hit& hit) {
    Bsdf b;
    b.n = normalized(hit.n);
    f3 bt = normalized(hit.dp_du);
    b.tan = cross(b.n, bt);
    b.bitan = cross(b.tan, b.n);
    b.p = hit.p;
    b.ng = hit.ng;
    const TrayMaterial* __restrict__ m = sc.materials + sc.instances[hit.inst].material_id;
    b.mat = m;
    b.merl = nullptr;
    b.n_lobes = 0;
    f3 c0 = mk(m->c0[0], m->c0[1], m->c0[2]), c1 = mk(m->c1[0], m->c1[1], m->c1[2]);
    const f3 white = mk(1.0f, 1.0f, 1.0f);
    float f0 = m->f0, f1 = m->f1;
    Lobe l;
    l.eta_t = 1.0f; l.width = 0.0f; l.ob = 0.0f;
    switch (m->kind) {
        case TRAY_MAT_MATTE: {   // matte.rs:52-65, oren_nayar.rs:26-34
            l.color = c0; l.type = BX_DIFFUSE | BX_REFLECTION;
            if (f0 == 0.0f) { l.kind = LB_LAMBERTIAN; }
            else {
                l.kind = LB_OREN_NAYAR;
                float sigma = to_radians(f0);
                sigma *= sigma;
                l.width = 1.0f - 0.5f * sigma / (sigma + 0.33f);
                l.ob = 0.45f * sigma / (sigma + 0.09f);
            }
            b.lobe[b.n_lobes++] = l;
            break;
        }
        case TRAY_MAT_PLASTIC: {   // plastic.rs:59-88
            if (!is_black(c0)) { l.kind = LB_LAMBERTIAN; l.type = BX_DIFFUSE | BX_REFLECTION; l.color = c0; b.lobe[b.n_lobes++] = l; }
            if (!is_black(c1)) {
                l.kind = LB_TS_DIEL; l.type = BX_GLOSSY | BX_REFLECTION; l.color = c1; l.eta_t = 1.5f; l.width = fmaxf(f0, 0.000001f);
                b.lobe[b.n_lobes++] = l;
            }
            break;
        }
        case TRAY_MAT_METAL: {   // metal.rs:56-67
            l.kind = LB_TS_COND; l.type = BX_GLOSSY | BX_REFLECTION; l.color = white; l.width = fmaxf(f0, 0.000001f);
            b.lobe[b.n_lobes++] = l;
            break;
        }
        case TRAY_MAT_GLASS: {   // glass.rs:51-78
            l.eta_t = f0;
            if (!is_black(c0)) { l.kind = LB_SPEC_REFL_DIEL; l.type = BX_SPECULAR | BX_REFLECTION; l.color = c0; b.lobe[b.n_lobes++] = l; }
            if (!is_black(c1)) { l.kind = LB_SPEC_TRANS; l.type = BX_SPECULAR | BX_TRANSMISSION; l.color = c1; b.lobe[b.n_lobes++] = l; }
            break;
        }
        case TRAY_MAT_ROUGH_GLASS: {   // rough_glass.rs:57-85
            l.eta_t = f0; l.width = fmaxf(f1, 0.000001f);
            if (!is_black(c0)) { l.kind = LB_TS_DIEL; l.type = BX_GLOSSY | BX_REFLECTION; l.color = c0; b.lobe[b.n_lobes++] = l; }
            if (!is_black(c1)) { l.kind = LB_MF_TRANS; l.type = BX_GLOSSY | BX_TRANSMISSION; l.color = c1; b.lobe[b.n_lobes++] = l; }
            break;
        }
        case TRAY_MAT_SPECULAR_METAL: {   // specular_metal.rs:49-58
            l.kind = LB_SPEC_REFL_COND; l.type = BX_SPECULAR | BX_REFLECTION; l.color = white;
            b.lobe[b.n_lobes++] = l;
            break;
        }
        default: {   // TRAY_MAT_MERL, material/merl.rs:88-92
            l.kind = LB_MERL; l.type = BX_GLOSSY | BX_REFLECTION; l.color = white;
            b.merl = sc.merl_data + sc.merl_tables[m->table].offset;
            b.lobe[b.n_lobes++] = l;
            break;
        }
    }
    return b;
}

}  // namespace tr
