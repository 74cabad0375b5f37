// Are the device's restatements of glibc's acosf / sinf / cosf (tray_rust_amd/csrc/hip/dev_libm.h) the SYSTEM libm's functions, bit for bit?
// (The reference's f32::acos / sin / cos in Quaternion::slerp, quaternion.rs:101-113, resolve to them on Linux; the oracle calls them.)
//   g++ -O2 -fno-builtin -ffp-contract=off tools/libm_port_check.cpp -o /tmp/libm_port_check -lm && /tmp/libm_port_check [stride]
// stride 1 (default): acosf on all 2 130 706 434 arguments in [-1, 1], sinf / cosf on all 1 078 774 990 floats in [0, 3.2]  (~1.5 min);
// also how often the functions differ from the correctly rounded value (f64 result rounded once) on slerp's argument ranges.
// Round 4, glibc 2.35 (Ubuntu 22.04 image): 0 differences; acosf != rounded f64 for 7.76 % of the arguments in (-1, 0.9995), sinf 1.50 %, cosf 1.08 % in (0, pi).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define TR_DEV static inline
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
#include "../tray_rust_amd/csrc/hip/dev_libm.h"

int main(int argc, char** argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 1u;
    long da = 0, ds = 0, dc = 0, na = 0, ns = 0;
    for (uint64_t b = 0; b <= 0x3f800000u; b += stride)
        for (int sgn = 0; sgn < 2; ++sgn) {
            const float x = __uint_as_float((uint32_t)b | (sgn ? 0x80000000u : 0u));
            const float a = acosf(x), p = tr::ref_acosf(x);
            if (std::memcmp(&a, &p, 4)) { if (da < 5) std::printf("acosf(%a): libm %a, device source %a\n", x, a, p); ++da; }
            ++na;
        }
    const uint32_t hi = __float_as_uint(3.2f);
    for (uint64_t b = 0; b <= hi; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float s = sinf(x), ps = tr::ref_sincosf(x, 0), c = cosf(x), pc = tr::ref_sincosf(x, 1);
        if (std::memcmp(&s, &ps, 4)) { if (ds < 5) std::printf("sinf(%a): libm %a, device source %a\n", x, s, ps); ++ds; }
        if (std::memcmp(&c, &pc, 4)) { if (dc < 5) std::printf("cosf(%a): libm %a, device source %a\n", x, c, pc); ++dc; }
        ++ns;
    }
    std::printf("acosf: %ld of %ld arguments differ; sinf: %ld, cosf: %ld of %ld\n", da, na, ds, dc, ns);
    // how far the libm functions are from "correctly rounded" on slerp's ranges (why rounding an f64 result was not enough)
    long ra = 0, rs = 0, rc = 0;
    const long n = 4000000;
    std::srand(1);
    for (long i = 0; i < n; ++i) {
        float x = (float)std::rand() / (float)RAND_MAX * 1.9995f - 1.0f;
        if (x > 0.9995f) x = 0.9995f;
        const float a = acosf(x);
        if (a != (float)std::acos((double)x)) ++ra;
        const float th = a * ((float)std::rand() / (float)RAND_MAX);
        if (sinf(th) != (float)std::sin((double)th)) ++rs;
        if (cosf(th) != (float)std::cos((double)th)) ++rc;
    }
    std::printf("libm vs f64 rounded once: acosf differs for %.2f %% of arguments in (-1, 0.9995), sinf %.2f %%, cosf %.2f %% in (0, pi)\n", 100.0 * ra / n, 100.0 * rs / n, 100.0 * rc / n);
    return (da || ds || dc) ? 1 : 0;
}
