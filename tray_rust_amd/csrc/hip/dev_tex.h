// Device-side image textures: the reference's Texture trait objects (src/texture/mod.rs:15-40, image.rs:9-47,
// animated_image.rs:7-58) over RGBA8 frames in HBM, and the per-hit lowering of a textured material.
// Material::bsdf samples every parameter at (hit.dg.u, hit.dg.v, hit.dg.time) and builds its BxDFs from the sampled values
// (e.g. matte.rs:55-63); materials whose parameters are all constants are lowered once on the host instead (dev_bsdf.h).
#pragma once
#include "dev_bsdf.h"

namespace tr {

TR_DEV uint32_t f32_as_u32(float x) { return x > 0.0f ? (x >= 4294967296.0f ? 0xffffffffu : (uint32_t)x) : 0u; }   // Rust's saturating `as u32`

struct Rgba { float r, g, b, a; };
TR_DEV Rgba operator*(Rgba c, float s) { Rgba o; o.r = c.r * s; o.g = c.g * s; o.b = c.b * s; o.a = c.a * s; return o; }
TR_DEV Rgba operator+(Rgba x, Rgba y) { Rgba o; o.r = x.r + y.r; o.g = x.g + y.g; o.b = x.b + y.b; o.a = x.a + y.a; return o; }

// Image::get_color (image.rs:24-33): clamped texel, channels / 255
TR_DEV Rgba tex_texel(const DevScene& sc, const TrayTexFrame* __restrict__ fr, uint32_t x, uint32_t y) {
    x = x > fr->width - 1u ? fr->width - 1u : x;
    y = y > fr->height - 1u ? fr->height - 1u : y;
    const uint32_t px = *reinterpret_cast<const uint32_t*>(sc.tex_data + fr->offset + ((size_t)y * fr->width + x) * 4u);   // offsets are multiples of 4
    Rgba c;
    c.r = (float)(px & 0xffu) / 255.0f; c.g = (float)((px >> 8) & 0xffu) / 255.0f; c.b = (float)((px >> 16) & 0xffu) / 255.0f; c.a = (float)(px >> 24) / 255.0f;
    return c;
}
// Image::sample_color: bilinear_interpolate over the texels (x as u32, y as u32) .. +1 (texture/mod.rs:22-40)
TR_DEV Rgba image_sample(const DevScene& sc, const TrayTexFrame* __restrict__ fr, float u, float v) {
    const float x = u * (float)fr->width, y = v * (float)fr->height;
    const uint32_t x0 = f32_as_u32(x), y0 = f32_as_u32(y);
    const Rgba s00 = tex_texel(sc, fr, x0, y0), s10 = tex_texel(sc, fr, x0 + 1u, y0), s01 = tex_texel(sc, fr, x0, y0 + 1u), s11 = tex_texel(sc, fr, x0 + 1u, y0 + 1u);
    const float sx = x - (float)x0, sy = y - (float)y0;
    return s00 * (1.0f - sx) * (1.0f - sy) + s10 * sx * (1.0f - sy) + s01 * (1.0f - sx) * sy + s11 * sx * sy;
}
// Texture::sample_color of an Image or an AnimatedImage (active_keyframes: binary search of the frame times, lerp between the two
// frames around `time`). sample_f32 is the same arithmetic on the red channel alone (get_float), so it is this function's .r
TR_DEV Rgba texture_sample(const DevScene& sc, uint32_t tex, float u, float v, float time) {
    const TrayTexture t = sc.textures[tex];
    const TrayTexFrame* __restrict__ fr = sc.tex_frames + t.first_frame;
    if (t.n_frames < 2u) return image_sample(sc, fr, u, v);
    uint32_t a = 0u, b = t.n_frames, lo = 0u;
    bool two = false, exact = false;
    while (a < b) {
        const uint32_t mid = a + (b - a) / 2u;
        const float tm = fr[mid].time;
        if (tm == time) { lo = mid; exact = true; break; }
        if (tm < time) a = mid + 1u; else b = mid;
    }
    if (!exact) {
        if (a == t.n_frames) lo = t.n_frames - 1u;
        else if (a == 0u) lo = 0u;
        else { lo = a - 1u; two = true; }
    }
    if (!two) return image_sample(sc, fr + lo, u, v);
    const float x = (time - fr[lo].time) / (fr[lo + 1u].time - fr[lo].time);
    return image_sample(sc, fr + lo, u, v) * (1.0f - x) + image_sample(sc, fr + lo + 1u, u, v) * x;   // linalg::lerp
}

// Material::bsdf of a textured material at one hit: sample the textured parameters, lower the values (dev_bsdf.h: lower_values)
TR_DEV void resolve_textured(const DevScene& sc, const DevMaterial* __restrict__ m, float u, float v, float time, DevMaterial& out) {
    float c0[3] = {m->c0[0], m->c0[1], m->c0[2]}, c1[3] = {m->c1[0], m->c1[1], m->c1[2]};
    float f0 = m->f0, f1 = m->f1;
    if (m->tex_c0 != TRAY_NO_TEXTURE) { const Rgba c = texture_sample(sc, m->tex_c0, u, v, time); c0[0] = c.r; c0[1] = c.g; c0[2] = c.b; }
    if (m->tex_c1 != TRAY_NO_TEXTURE) { const Rgba c = texture_sample(sc, m->tex_c1, u, v, time); c1[0] = c.r; c1[1] = c.g; c1[2] = c.b; }
    if (m->tex_f0 != TRAY_NO_TEXTURE) f0 = texture_sample(sc, m->tex_f0, u, v, time).r;
    if (m->tex_f1 != TRAY_NO_TEXTURE) f1 = texture_sample(sc, m->tex_f1, u, v, time).r;
    lower_values(out, m->mat_kind, c0, c1, f0, f1, m->microfacet);
    out.merl_offset = m->merl_offset;
    out.textured = 0u;
}

}  // namespace tr
