// glibc's acosf / sinf / cosf / atanf / atan2f / expf / logf restated for the device (and, through tools/libm_port_check.cpp, compiled for the
// host to be compared with the system libm bit for bit). Plain C++ apart from TR_DEV / __float_as_uint / __uint_as_float, which dev_math.h,
// the host emulation or the checker provide.
// Provenance: the algorithms, polynomial coefficients and the two small tables are those of the GNU C Library 2.35 (sysdeps/ieee754/flt-32:
// e_acosf.c, s_atanf.c, e_atan2f.c -- derived from Sun's fdlibm, "Copyright (C) 1993 by Sun Microsystems, Inc. ... Permission to use, copy,
// modify, and distribute this software is freely granted, provided that this notice is preserved" --; s_sincosf.h, e_expf.c, e_exp2f_data.c,
// e_logf.c, e_logf_data.c -- contributed by Arm's optimized-routines, Copyright (C) the Free Software Foundation, distributed under the GNU
// Lesser General Public License 2.1 or later). They are restated here (not copied: one function per algorithm, branch structure re-cut for
// SIMT) for ONE purpose: the reference's f32 methods resolve to exactly these functions on Linux, and parity with it is checked bit for bit.
#pragma once
#include <stdint.h>
#include <string.h>

namespace tr {

#if defined(__HIP_DEVICE_COMPILE__) || (defined(__HIPCC__) && !defined(TR_HOST_EMU))
#define TR_DEV_TABLE static __device__ const
TR_DEV uint64_t ref_d2u(double d) { return (uint64_t)__double_as_longlong(d); }
TR_DEV double ref_u2d(uint64_t u) { return __longlong_as_double((long long)u); }
#else   // host emulation of the device source (tests/emu), tools/libm_port_check.cpp
#define TR_DEV_TABLE static const
TR_DEV uint64_t ref_d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
TR_DEV double ref_u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
#endif

// The three libm calls of Quaternion::slerp (quaternion.rs:101-113): f32::acos / cos / sin are the host libm's acosf / cosf / sinf, on
// Linux glibc's -- neither correctly rounded (acosf differs from the rounded f64 value for 7.8 % of the arguments in (-1, 0.9995), sinf /
// cosf for 1.5 % / 1.1 % in (0, pi): tools/libm_port_check.c), and every differing ulp moves a whole instance for that path. So the device
// runs glibc 2.35's own algorithms (the image's libm, which the oracle calls): acosf is fdlibm's e_acosf.c in f32, sinf / cosf the
// double-precision polynomials of s_sincosf.h after the fast pi/2 reduction. tools/libm_port_check.c compares these restatements with
// the system libm bit for bit: acosf on all 2 130 706 434 arguments in [-1, 1], sinf / cosf on all 1 078 774 990 floats in [0, 3.2]
// (slerp's angles lie in [0, pi]) -- zero differences, with and without fused multiply-adds in the f64 polynomial.
// (one rational p(z) / q(z) and one square root for all three ranges -- the ranges differ in z and in how the pieces are put together --, so that
// the lanes of a wave whose arguments fall into different ranges run ONE division and ONE square root, not three of each)
TR_DEV float ref_acosf(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
                pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
                pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix >= 0x3f800000) {
        if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
        return (x - x) / (x - x);
    }
    const bool small = ix < 0x3f000000;   // |x| < 0.5
    if (small && ix <= 0x23000000) return pio2_hi + pio2_lo;
    const float z = small ? x * x : (hx < 0 ? (one + x) * 0.5f : (one - x) * 0.5f);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    if (small) return pio2_hi - (x - (pio2_lo - x * r));
    const float s = sqrtf(z);
    if (hx < 0) {   // x < -0.5
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);   // x > 0.5
    const float c = (z - df * df) / (s + df);
    const float w = r * s + c;
    return 2.0f * (df + w);
}
// sinf (want_cos = 0) / cosf (1) for |y| < 120; the sign table and the negated cosine coefficients of __sincosf_table[1] are `neg`
TR_DEV float ref_sincosf(float y, int want_cos) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;   // abstop12
    double x = (double)y;
    int n = 0;
    double sgn = 1.0, neg = 1.0;
    if (top < ((0x3f490fdbu >> 20) & 0x7ffu)) {   // |y| < pi / 4: no reduction, quadrant 0
        if (top < ((0x39800000u >> 20) & 0x7ffu)) return want_cos ? 1.0f : y;   // |y| < 2^-12
    } else {   // reduce_fast: n = round(y * 2 / pi) through a 2^24-scaled truncation, x = y - n * pi / 2
        const double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * hpi;
        const int k = n & 3;
        sgn = (k == 1 || k == 2) ? -1.0 : 1.0;   // __sincosf_table[0].sign[n & 3]
        if (n & 2) neg = -1.0;                   // __sincosf_table[1]: the cosine coefficients negated, the sine ones as they are
    }
    const double x2 = x * x;
    x = x * sgn;
    if (((n ^ want_cos) & 1) == 0) {
        const double x3 = x * x2, sa = s2 + x2 * s3, x7 = x3 * x2, sb = x + x3 * s1;
        return (float)(sb + x7 * sa);
    }
    const double x4 = x2 * x2, cb = neg * c3 + x2 * (neg * c4), ca = neg * c0 + x2 * (neg * c1), x6 = x4 * x2, cc = ca + x4 * (neg * c2);
    return (float)(cc + x6 * cb);
}

// cosf and sinf of ONE argument (mc.rs:49-50, linalg::spherical_dir: the reference calls f32::cos and f32::sin side by side), |y| < 119, with
// the reduction and the squares shared. On an x86-64 host with FMA (every one this runs beside) glibc dispatches to its `-fma` builds of
// s_sinf.c / s_cosf.c / e_expf.c, the same C compiled with -mfma, in which gcc contracts a * b + c: the fma() calls below are where the
// system libm's machine code has vfmadd (read off its disassembly); unfused, 28 of the 2.2e9 arguments in (-100, 100) differ, fused none
TR_DEV void ref_sincosf2(float y, float& sin_out, float& cos_out) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
    double x = (double)y;
    int n = 0;
    double sgn = 1.0, neg = 1.0;
    if (top < ((0x3f490fdbu >> 20) & 0x7ffu)) {
        if (top < ((0x39800000u >> 20) & 0x7ffu)) { sin_out = y; cos_out = 1.0f; return; }
    } else {
        const double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = fma(-(double)n, hpi, x);
        const int k = n & 3;
        sgn = (k == 1 || k == 2) ? -1.0 : 1.0;
        if (n & 2) neg = -1.0;
    }
    const double x2 = x * x;
    x = x * sgn;
    const double x3 = x * x2, sa = fma(x2, s3, s2), x7 = x3 * x2, sb = fma(x3, s1, x);
    const float sv = (float)fma(x7, sa, sb);
    const double x4 = x2 * x2, cb = fma(x2, neg * c4, neg * c3), ca = fma(x2, neg * c1, neg * c0), x6 = x4 * x2, cc = fma(x4, neg * c2, ca);
    const float cv = (float)fma(x6, cb, cc);
    // quadrant n: sin(y) is the sine polynomial for even n, the cosine one for odd n; cos(y) the other way round
    sin_out = (n & 1) ? cv : sv;
    cos_out = (n & 1) ? sv : cv;
}

// glibc 2.35's atanf (sysdeps/ieee754/flt-32/s_atanf.c: fdlibm in f32) and atan2f (e_atan2f.c) for finite arguments -- what f32::atan2
// resolves to in bxdf/merl.rs:72 (linalg::spherical_phi), sphere.rs:71 and the sphere's sampling code
TR_DEV float ref_atanf(float x) {
    const float atanhi0 = 4.6364760399e-01f, atanhi1 = 7.8539812565e-01f, atanhi2 = 9.8279368877e-01f, atanhi3 = 1.5707962513e+00f;
    const float atanlo0 = 5.0121582440e-09f, atanlo1 = 3.7748947079e-08f, atanlo2 = 3.4473217170e-08f, atanlo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f,
                aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
                aT10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {   // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi3 + atanlo3 : -atanhi3 - atanlo3;
    }
    const bool direct = ix < 0x3ee00000;   // |x| < 0.4375: no reduction
    if (direct && ix < 0x31000000) return x;   // |x| < 2^-29
    // the four reductions of s_atanf.c are quotients num / den of sums with one rounding each, as written there; ONE division serves whichever
    // range the lane's argument is in (the lanes of a wave rarely agree)
    float hi = 0.0f, lo = 0.0f;
    if (!direct) {
        const float ax = fabsf(x);
        float num, den;
        if (ix < 0x3f300000) { hi = atanhi0; lo = atanlo0; num = 2.0f * ax - 1.0f; den = 2.0f + ax; }          // 7/16 <= |x| < 11/16
        else if (ix < 0x3f980000) { hi = atanhi1; lo = atanlo1; num = ax - 1.0f; den = ax + 1.0f; }              // < 19/16
        else if (ix < 0x401c0000) { hi = atanhi2; lo = atanlo2; num = ax - 1.5f; den = 1.0f + 1.5f * ax; }        // < 39/16
        else { hi = atanhi3; lo = atanlo3; num = -1.0f; den = ax; }
        x = num / den;
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (direct) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}
TR_DEV float ref_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff, hy = (int32_t)__float_as_uint(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return ref_atanf(y);
    const int m = (int)(((uint32_t)hy >> 31) & 1u) | (int)(((uint32_t)hx >> 30) & 2u);   // 2 * sign(x) + sign(y)
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = ref_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// glibc 2.35's expf (sysdeps/ieee754/flt-32/e_expf.c + e_exp2f_data.c, N = 32) and logf (e_logf.c + e_logf_data.c, N = 16): f64 arithmetic
// around a table -- Beckmann's D and its sampling (beckmann.rs:33-48) call f32::exp / f32::ln. The tables are glibc's
TR_DEV_TABLE uint64_t kRefExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
TR_DEV float ref_expf(float x) {
    const double inv_ln2_n = 0x1.71547652b82fep+5, shift = 0x1.8p+52;
    const double p0 = 0x1.c6af84b912394p-20, p1 = 0x1.ebfce50fac4f3p-13, p2 = 0x1.62e42ff0c52d6p-6;   // poly_scaled
    const uint32_t abstop = (__float_as_uint(x) >> 20) & 0x7ffu;
    if (abstop >= ((0x42b00000u >> 20) & 0x7ffu)) {   // |x| >= 88 or NaN
        if (__float_as_uint(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);   // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                          // underflow
    }
    const double xd = (double)x;
    double kd = fma(inv_ln2_n, xd, shift);   // (the -fma build never rounds z = InvLn2N * xd by itself: kd and r both come from the exact product)
    const uint64_t ki = ref_d2u(kd);
    kd = kd - shift;
    const double r = fma(inv_ln2_n, xd, -kd);
    const uint64_t t = kRefExp2fTab[ki & 31u] + (ki << (52 - 5));
    const double s = ref_u2d(t);
    const double zz = fma(p0, r, p1);
    const double r2 = r * r;
    double yy = fma(p2, r, 1.0);
    yy = fma(zz, r2, yy);
    yy = yy * s;
    return (float)yy;
}
TR_DEV_TABLE double kRefLogfTab[32] = {   // {invc, logc} x 16
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2, 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2,
    0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3, 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4, 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5,
    0x1p+0, 0x0p+0, 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3, 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,
    0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};
TR_DEV float ref_logf(float x) {
    const double ln2 = 0x1.62e42fefa39efp-1, a0 = -0x1.00ea348b88334p-2, a1 = 0x1.5575b0be00b6ap-2, a2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {   // x < 0x1p-126, inf or NaN
        if (ix * 2u == 0u) return __uint_as_float(0xff800000u);   // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __uint_as_float(0x7fc00000u);
        ix = __float_as_uint(x * 0x1p23f);   // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> (23 - 4)) & 15u;
    const int32_t k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = kRefLogfTab[2u * i], logc = kRefLogfTab[2u * i + 1u];
    const double z = (double)__uint_as_float(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * ln2;
    const double r2 = r * r;
    double y = a1 * r + a2;
    y = a0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

}  // namespace tr
