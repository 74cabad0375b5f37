// Device-side math for the gfx950 path tracer: float3 helpers, affine/projective point transforms
// with the reference's w test, the TRAY-CBRNG counter RNG and the (0,2)-sequence.
// Arithmetic order follows the reference expression by expression (file:line cited per function);
// the translation unit is compiled with -ffp-contract=off so nothing is fused that rustc would not fuse.
#pragma once
#ifndef TR_HOST_EMU   // (tests/emu/hip_emu.h stands in for the runtime header when the device code is compiled for the host)
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace tr {

#define TR_DEV __device__ __forceinline__
#ifndef TR_HOST_EMU
#define TR_DYN_LDS(T, name) extern __shared__ T name[]   // the workgroup's dynamic LDS (tests/emu supplies its own definition)
#define TR_EMU_PHASE(k) ((void)0)   // numbers the passes of a loop whose iterations the lanes of a wave execute together; read by the host emulation's divergence profile only
// LDS through its own address space: ds_read / ds_write instead of flat accesses (a `volatile` generic pointer into LDS compiles to
// flat_load / flat_store with a full s_waitcnt after every access). Lanes of one wave exchange data through it between
// TR_WAVE_SYNC()s: LDS operations of a wave complete in order, the fences only keep the compiler from moving accesses across.
typedef __attribute__((address_space(3))) float* LdsF;
typedef __attribute__((address_space(3))) uint32_t* LdsU;
typedef const __attribute__((address_space(3))) uint8_t* LdsB;
#define TR_LDS_F(generic_ptr) ((LdsF)(generic_ptr))
#define TR_LDS_U(generic_ptr) ((LdsU)(generic_ptr))
#define TR_LDS_B(generic_ptr) ((LdsB)(generic_ptr))
#define TR_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
typedef float* LdsF;
typedef uint32_t* LdsU;
typedef const uint8_t* LdsB;
#define TR_LDS_F(generic_ptr) ((float*)(generic_ptr))
#define TR_LDS_U(generic_ptr) ((uint32_t*)(generic_ptr))
#define TR_LDS_B(generic_ptr) ((const uint8_t*)(generic_ptr))
#define TR_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 0.318309886183790671538f;
static constexpr float kPiOver4 = 0.785398163397448309616f;
static constexpr float kEps = 1.1920929e-7f;   // f32::EPSILON
#define TR_INF __builtin_huge_valf()

}  // namespace tr
#include "dev_libm.h"
namespace tr {
// f32::{sin, cos, acos, atan2, exp, ln} of the reference are the host libm's functions -- none of them correctly rounded, so ocml's differ
// from them in the last bit for a few per cent of the arguments, and bxdf/merl.rs:63-79 turns such a bit into another table entry. The device
// calls glibc's algorithms restated (dev_libm.h; tools/libm_port_check.cpp: every bit pattern agrees with the system libm), so a camera sample's
// radiance is the oracle's bit for bit. -DTR_OCML_LIBM builds rounds 1-4's form (ocml) for the A/B of what that costs.
#ifdef TR_OCML_LIBM
TR_DEV void lm_sincos(float x, float& s, float& c) { c = cosf(x); s = sinf(x); }
TR_DEV float lm_sin(float x) { return sinf(x); }
TR_DEV float lm_acos(float x) { return acosf(x); }
TR_DEV float lm_atan2(float y, float x) { return atan2f(y, x); }
TR_DEV float lm_exp(float x) { return expf(x); }
TR_DEV float lm_log(float x) { return logf(x); }
#else
TR_DEV void lm_sincos(float x, float& s, float& c) { ref_sincosf2(x, s, c); }
TR_DEV float lm_sin(float x) { float s, c; ref_sincosf2(x, s, c); return s; }
TR_DEV float lm_acos(float x) { return ref_acosf(x); }
TR_DEV float lm_atan2(float y, float x) { return ref_atan2f(y, x); }
TR_DEV float lm_exp(float x) { return ref_expf(x); }
TR_DEV float lm_log(float x) { return ref_logf(x); }
#endif

struct f3 {
    float x, y, z;
};
TR_DEV f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
TR_DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
TR_DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
TR_DEV f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
TR_DEV f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
TR_DEV f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
TR_DEV f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
TR_DEV f3 operator/(f3 a, f3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
TR_DEV f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
TR_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // linalg/mod.rs:42-44
TR_DEV f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
TR_DEV float length_sqr(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
TR_DEV f3 normalized(f3 a) { float l = sqrtf(length_sqr(a)); return mk(a.x / l, a.y / l, a.z / l); }   // vector.rs:33-36
// 1.0f / x. The compiler builds the correctly rounded quotient from v_div_scale x 2, v_rcp_f32, five v_fma, v_div_fmas and v_div_fixup (~11 VALU
// instructions, 42 cycles: profiles/r04_ubench_valu.txt). gfx950's v_rcp_f32 followed by ONE Newton step in fma form gives the same bits for EVERY x
// with 2^-126 <= |x| < 2^126 -- tools/ubench_rcp.hip compares all 2^32 arguments on the MI355X (profiles/r05_rcp_exhaustive.txt: the only arguments
// that differ have a biased exponent of 0 or >= 253) -- so the wave takes the three-instruction form when all of its active lanes are inside that
// range and the compiler's division otherwise (a wave-uniform branch). -DTR_IEEE_RCP: the compiler's division everywhere, as before.
TR_DEV bool rcp_in_range(float x) { return ((__float_as_uint(x) & 0x7fffffffu) - 0x00800000u) < 0x7e000000u; }   // biased exponent 1 ... 252
// `relevant`: the lane's quotient will be read (lanes that only run along -- no ray, no path -- hold stale or zero arguments and must not send the wave
// down the slow branch; what they compute is never used)
#if defined(TR_HOST_EMU) || defined(TR_IEEE_RCP)
TR_DEV float rcp_rn(float x, bool relevant = true) { (void)relevant; return 1.0f / x; }
TR_DEV f3 rcp_rn3(f3 d, bool relevant = true) { (void)relevant; return mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
#else
TR_DEV float rcp_newton(float x) { const float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r); }
TR_DEV float rcp_rn(float x, bool relevant = true) {
    if (__all(!relevant || rcp_in_range(x))) return rcp_newton(x);
    return 1.0f / x;
}
TR_DEV f3 rcp_rn3(f3 d, bool relevant = true) {
    if (__all(!relevant || (rcp_in_range(d.x) && rcp_in_range(d.y) && rcp_in_range(d.z)))) return mk(rcp_newton(d.x), rcp_newton(d.y), rcp_newton(d.z));
    return mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}
#endif
TR_DEV bool is_black(f3 c) { return c.x == 0.0f && c.y == 0.0f && c.z == 0.0f; }   // color.rs:47-49
TR_DEV float luminance(f3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
TR_DEV float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }   // linalg/mod.rs:51-53
TR_DEV float lerpf(float t, float a, float b) { return a * (1.0f - t) + b * t; }
TR_DEV float to_radians(float d) { return kPi / 180.0f * d; }

// Two f32 side by side for the packed VALU operations of gfx950 (v_pk_add_f32 / v_pk_mul_f32: two lanes' worth of arithmetic per
// issue slot, each half rounded exactly like the plain instruction -- tools/pk_denorm_check.hip; no v_pk_fma is formed: the
// translation unit is built with -ffp-contract=off). profiles/r04_ubench_valu.txt: a packed instruction costs 4.5-4.9 cycles against
// 3.8-3.9 for the plain one at >= 4 waves per SIMD, i.e. 0.6 of the issue time per f32.
#ifndef TR_HOST_EMU
typedef float f2 __attribute__((ext_vector_type(2)));
#else
struct f2 { float x, y; };
inline f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
inline f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
inline f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
#endif
TR_DEV f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
TR_DEV f2 splat2(float a) { return mk2(a, a); }
// (a - s.lo, b - s.lo) etc.: a packed operation whose second operand is ONE half of a register pair for both results -- the op_sel
// operand modifiers of the VOP3P encoding (op_sel picks the half that feeds the low result, op_sel_hi the high one), so two scalars
// share a pair and nothing is moved. hipcc does not form these from shuffles (it copies the scalar into both halves of a fresh pair).
#ifndef TR_HOST_EMU
TR_DEV f2 pk_sub_lo(f2 a, f2 s) { f2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(s)); return r; }
TR_DEV f2 pk_sub_hi(f2 a, f2 s) { f2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(s)); return r; }
TR_DEV f2 pk_mul_lo(f2 a, f2 s) { f2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(s)); return r; }
TR_DEV f2 pk_mul_hi(f2 a, f2 s) { f2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(s)); return r; }
#else
inline f2 pk_sub_lo(f2 a, f2 s) { return f2{a.x - s.x, a.y - s.x}; }
inline f2 pk_sub_hi(f2 a, f2 s) { return f2{a.x - s.y, a.y - s.y}; }
inline f2 pk_mul_lo(f2 a, f2 s) { return f2{a.x * s.x, a.y * s.x}; }
inline f2 pk_mul_hi(f2 a, f2 s) { return f2{a.x * s.y, a.y * s.y}; }
#endif

// Transform * Point / inv_mul_point (transform.rs:152-163,199-216): m is a row-major 4x4
TR_DEV f3 xf_point(const float* __restrict__ m, f3 p) {
    f3 r;
    r.x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    r.y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    r.z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    // quirk Q5: divides only when w is (almost) one. x / 1.0f == x for every x, so the (three IEEE) divides are skipped when w is
    // exactly one -- the case of every affine instance transform (row 3 = 0 0 0 1): same bits, ~45 FMA-equivalents less per call
    if (w != 1.0f && fabsf(w - 1.0f) < kEps) r = r / w;
    return r;
}
TR_DEV f3 xf_vector(const float* __restrict__ m, f3 v) {   // transform.rs:165-172,218-229
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
TR_DEV f3 xf_normal_t(const float* __restrict__ inv, f3 n) {   // Transform * Normal uses inv^T (transform.rs:231-243)
    return mk(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z, inv[2] * n.x + inv[6] * n.y + inv[10] * n.z);
}

TR_DEV void coordinate_system(f3 e1, f3& e2, f3& e3) {   // linalg/mod.rs:96-108
    if (fabsf(e1.x) > fabsf(e1.y)) {
        float inv_len = 1.0f / sqrtf(e1.x * e1.x + e1.z * e1.z);
        e2 = mk(-e1.z * inv_len, 0.0f, e1.x * inv_len);
    } else {
        float inv_len = 1.0f / sqrtf(e1.y * e1.y + e1.z * e1.z);
        e2 = mk(0.0f, e1.z * inv_len, -e1.y * inv_len);
    }
    e3 = cross(e1, e2);
}
TR_DEV f3 reflect(f3 w, f3 v) { return 2.0f * dot(w, v) * v - w; }   // linalg/mod.rs:110-112
TR_DEV bool refract(f3 w, f3 n, float eta, f3& out) {   // linalg/mod.rs:117-127
    float cos_t1 = dot(n, w);
    float sin_t1_sqr = fmaxf(0.0f, 1.0f - cos_t1 * cos_t1);
    float sin_t2_sqr = eta * eta * sin_t1_sqr;
    if (sin_t2_sqr >= 1.0f) return false;
    float cos_t2 = sqrtf(1.0f - sin_t2_sqr);
    out = eta * -w + (eta * cos_t1 - cos_t2) * n;
    return true;
}
TR_DEV bool solve_quadratic(float a, float b, float c, float& t0, float& t1) {   // linalg/mod.rs:78-94
    float discrim_sqr = b * b - 4.0f * a * c;
    if (discrim_sqr < 0.0f) return false;
    float discrim = sqrtf(discrim_sqr);
    float q = b < 0.0f ? -0.5f * (b - discrim) : -0.5f * (b + discrim);
    float x = q / a, y = c / q;
    if (x > y) { t0 = y; t1 = x; } else { t0 = x; t1 = y; }
    return true;
}

// ---- mc.rs
TR_DEV void concentric_sample_disk(float u0, float u1, float& dx, float& dy) {   // mc.rs:23-51
    float sx = 2.0f * u0 - 1.0f, sy = 2.0f * u1 - 1.0f;
    if (sx == 0.0f && sy == 0.0f) { dx = sx; dy = sy; return; }
    float radius, theta;
    if (sx >= -sy) {
        if (sx > sy) { radius = sx; theta = sy > 0.0f ? sy / sx : 8.0f + sy / sx; }
        else { radius = sy; theta = 2.0f - sx / sy; }
    } else if (sx <= sy) { radius = -sx; theta = 4.0f + sy / sx; }
    else { radius = -sy; theta = 6.0f - sx / sy; }
    theta = theta * kPiOver4;
    float sn, cs;
    lm_sincos(theta, sn, cs);
    dx = radius * cs;
    dy = radius * sn;
}
TR_DEV f3 cos_sample_hemisphere(float u0, float u1) {   // mc.rs:11-16
    float dx, dy;
    concentric_sample_disk(u0, u1, dx, dy);
    return mk(dx, dy, sqrtf(fmaxf(0.0f, 1.0f - dx * dx - dy * dy)));
}
TR_DEV float power_heuristic(float n_f, float pdf_f, float n_g, float pdf_g) {   // mc.rs:56-60
    float f = n_f * pdf_f, g = n_g * pdf_g;
    return (f * f) / (f * f + g * g);
}

// ---- TRAY-CBRNG: stateless counter-based draws (definition: DESIGN.md; oracle has its own copy)
TR_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
TR_DEV uint32_t key_frame(uint64_t seed, uint32_t frame) {
    uint32_t h = mix32((uint32_t)seed + 0x9E3779B9u);
    h = mix32(h ^ (uint32_t)(seed >> 32));
    return mix32(h + frame);
}
TR_DEV uint32_t key_pixel(uint32_t kf, uint32_t pixel_index) { return mix32(kf + 0x9E3779B1u * (pixel_index + 1u)); }
TR_DEV uint32_t key_sample(uint32_t kp, uint32_t s) { return mix32((kp ^ 0xA511E9B3u) + 0x9E3779B1u * (s + 1u)); }
TR_DEV uint32_t draw(uint32_t key, uint32_t dim) { return mix32(key + 0x9E3779B9u * (dim + 1u)); }
// pass j of a pixel under the Adaptive sampler (adaptive.rs:92-110: every get_samples call draws fresh scrambles and shuffles)
TR_DEV uint32_t key_pass(uint32_t kp, uint32_t j) { return mix32((kp ^ 0x41445054u) + 0x9E3779B1u * (j + 1u)); }

enum { PD_SCR_X = 0, PD_SCR_Y = 1, PD_PERM_XY = 2, PD_SCR_T = 3, PD_PERM_T = 4 };
enum { SD_L2 = 0, SD_B2 = 3, SD_P2 = 6, SD_L1 = 9, SD_B1 = 11, SD_P1 = 13, SD_RR = 16 };

// Kensler's hashed permutation of [0, l): random access replacement of rng.shuffle (ld.rs:58,63)
TR_DEV uint32_t permute(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893du;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8; i *= 0x0929eb3fu;
        i ^= p >> 23;
        i ^= (i & w) >> 1; i *= 1u | p >> 27;
        i *= 0x6935fa69u;
        i ^= (i & w) >> 11; i *= 0x74dcb303u;
        i ^= (i & w) >> 2; i *= 0x9e501cc3u;
        i ^= (i & w) >> 2; i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}
// ---- per-path shuffles of the six LD arrays (ld.rs:58,63 called from path.rs:55-60), TRAY-CBRNG v2.
// The reference shuffles every (max_depth + 1)-long array of every camera sample with its thread's unseeded RNG. Here a path's
// shuffle of an array is ONE of TR_PERM_POOL permutations of [0, n) that exist once per scene: permutation q of the pool is
// the Fisher-Yates shuffle (loop shape of Rng::shuffle: i from n-1 down to 1, j = (r16 * (i + 1)) >> 16, r16 the 16-bit
// fields of draw(key_q, k >> 1), k = n-1-i) under key_q = mix32(0x50455250 + q), and an array picks q = low byte of its (first)
// scramble word -- bits the scrambled 24-bit fractions never use. Round 2 ran that Fisher-Yates per array and per path vertex
// (six times ~100 instructions at every vertex: 9 % of the tile kernel); drawing from a pool costs one byte load.
// Pool entry (q, b) holds what the (0,2)-sequence point of index idx = perm_q[b] needs: high nibble = bit-reversed idx (the top
// four bits of van_der_corput's __brev(idx), idx < 16), low nibble = the top four bits of sobol()'s xor of direction numbers
// (0x8, 0xC, 0xA, 0xF for idx bits 0..3) -- so both coordinates are one xor with the scramble word, the same bits the loops of
// van_der_corput() / sobol() below produce for idx < 16.
#define TR_PERM_POOL 256
#define TR_PERM_BYTES (TR_PERM_POOL * 16)
#ifdef TR_HOST_EMU
#define TR_HDI inline
#else
#define TR_HDI __host__ __device__ inline
#endif
TR_HDI uint32_t mix32_hd(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// host: the scene's pool for arrays of n = max_depth + 1 <= 16 entries (tray_scene_create uploads it; the tile kernel keeps it in LDS)
inline void perm_pool_build(uint32_t n, uint8_t* out /* TR_PERM_BYTES */) {
    if (n > 16u) n = 16u;
    for (uint32_t q = 0; q < TR_PERM_POOL; ++q) {
        const uint32_t key = mix32_hd(0x50455250u + q);
        uint8_t perm[16];
        for (uint32_t i = 0; i < 16u; ++i) perm[i] = (uint8_t)i;
        for (uint32_t i = n > 0u ? n - 1u : 0u; i >= 1u; --i) {
            const uint32_t k = n - 1u - i;
            const uint32_t word = mix32_hd(key + 0x9E3779B9u * ((k >> 1) + 1u));   // draw(key, k >> 1)
            const uint32_t r16 = (k & 1u) ? (word >> 16) : (word & 0xffffu);
            const uint32_t j = (r16 * (i + 1u)) >> 16;
            const uint8_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
        }
        for (uint32_t b = 0; b < 16u; ++b) {
            const uint32_t idx = b < n ? perm[b] : 0u;
            const uint32_t rev4 = ((idx & 1u) << 3) | ((idx & 2u) << 1) | ((idx & 4u) >> 1) | ((idx & 8u) >> 3);
            const uint32_t sob4 = ((idx & 1u) ? 0x8u : 0u) ^ ((idx & 2u) ? 0xCu : 0u) ^ ((idx & 4u) ? 0xAu : 0u) ^ ((idx & 8u) ? 0xFu : 0u);
            out[q * 16u + b] = (uint8_t)((rev4 << 4) | sob4);
        }
    }
}

// ---- sampler/ld.rs:91-119
TR_DEV float u24_to_unit(uint32_t v) { return fminf((float)((v >> 8) & 0xffffffu) / 16777216.0f, 1.0f - kEps); }
TR_DEV float van_der_corput(uint32_t n, uint32_t scramble) { return u24_to_unit(__brev(n) ^ scramble); }
TR_DEV float sobol(uint32_t n, uint32_t scramble) {
    uint32_t i = 1u << 31;
    while (n != 0) {
        if (n & 1u) scramble ^= i;
        n >>= 1;
        i ^= i >> 1;
    }
    return u24_to_unit(scramble);
}

}  // namespace tr
