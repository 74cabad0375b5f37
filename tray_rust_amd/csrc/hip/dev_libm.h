// glibc's acosf / sinf / cosf restated for the device (and, through tools/libm_port_check.cpp, compiled for the host to be compared
// with the system libm bit for bit). Plain C++ apart from TR_DEV / __float_as_uint / __uint_as_float, which dev_math.h, the host
// emulation or the checker provide.
#pragma once
#include <stdint.h>

namespace tr {

// The three libm calls of Quaternion::slerp (quaternion.rs:101-113): f32::acos / cos / sin are the host libm's acosf / cosf / sinf, on
// Linux glibc's -- neither correctly rounded (acosf differs from the rounded f64 value for 7.8 % of the arguments in (-1, 0.9995), sinf /
// cosf for 1.5 % / 1.1 % in (0, pi): tools/libm_port_check.c), and every differing ulp moves a whole instance for that path. So the device
// runs glibc 2.35's own algorithms (the image's libm, which the oracle calls): acosf is fdlibm's e_acosf.c in f32, sinf / cosf the
// double-precision polynomials of s_sincosf.h after the fast pi/2 reduction. tools/libm_port_check.c compares these restatements with
// the system libm bit for bit: acosf on all 2 130 706 434 arguments in [-1, 1], sinf / cosf on all 1 078 774 990 floats in [0, 3.2]
// (slerp's angles lie in [0, pi]) -- zero differences, with and without fused multiply-adds in the f64 polynomial.
TR_DEV float ref_acosf(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
                pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
                pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {   // |x| < 0.5
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {   // x < -0.5
        const float z = (one + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrtf(z), r = p / q, w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (one - x) * 0.5f, s = sqrtf(z);   // x > 0.5
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q, w = r * s + c;
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

}  // namespace tr
