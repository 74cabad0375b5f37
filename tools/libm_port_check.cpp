// Are the device's restatements of glibc's acosf / sinf / cosf (tray_rust_amd/csrc/hip/dev_libm.h) the SYSTEM libm's functions, bit for bit?
// (The reference's f32::acos / sin / cos in Quaternion::slerp, quaternion.rs:101-113, resolve to them on Linux; the oracle calls them.)
//   g++ -O2 -fno-builtin -ffp-contract=off tools/libm_port_check.cpp -o /tmp/libm_port_check -lm && /tmp/libm_port_check [stride]
// stride 1 (default): acosf on all 2 130 706 434 arguments in [-1, 1], sinf / cosf on all 1 078 774 990 floats in [0, 3.2]  (~1.5 min);
// also how often the functions differ from the correctly rounded value (f64 result rounded once) on slerp's argument ranges.
// Round 5: + atanf / expf / logf on ALL 2^32 bit patterns, sinf / cosf through the shared form (ref_sincosf2) on every |x| < 119, atan2f on 1.6e9
// pseudo-random pairs (full bit patterns and the unit square) and the special-case grid -- what the BSDFs, the sphere and the samplers call
// (bxdf/merl.rs:63-75, microfacet/beckmann.rs:33-48, mc.rs:49-50, sphere.rs:71). Eight threads, ~1 min at stride 1.
// Round 4, glibc 2.35 (Ubuntu 22.04 image): 0 differences; acosf != rounded f64 for 7.76 % of the arguments in (-1, 0.9995), sinf 1.50 %, cosf 1.08 % in (0, pi).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
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
    {   // round 5: the BSDFs' libm
        auto same = [](float a, float b) { return std::memcmp(&a, &b, 4) == 0 || (a != a && b != b); };
        const int NT = 8;
        std::atomic<long> d_atan{0}, d_exp{0}, d_log{0}, d_sc{0}, d_at2{0}, n_all{0}, n_sc{0}, n_at2{0};
        std::vector<std::thread> th;
        for (int t = 0; t < NT; ++t) th.emplace_back([&, t] {
            long la = 0, ls = 0, l2 = 0;
            for (uint64_t b = (uint64_t)t * stride; b <= 0xffffffffull; b += (uint64_t)NT * stride) {
                const float x = __uint_as_float((uint32_t)b);
                ++la;
                if (!same(atanf(x), tr::ref_atanf(x)) && d_atan++ < 3) std::printf("atanf(%a): libm %a, device source %a\n", x, atanf(x), tr::ref_atanf(x));
                if (!same(expf(x), tr::ref_expf(x)) && d_exp++ < 3) std::printf("expf(%a): libm %a, device source %a\n", x, expf(x), tr::ref_expf(x));
                if (!same(logf(x), tr::ref_logf(x)) && d_log++ < 3) std::printf("logf(%a): libm %a, device source %a\n", x, logf(x), tr::ref_logf(x));
                if (std::fabs(x) < 119.0f) {
                    float sv, cv;
                    tr::ref_sincosf2(x, sv, cv);
                    ++ls;
                    if ((!same(sv, sinf(x)) || !same(cv, cosf(x))) && d_sc++ < 3) std::printf("sinf / cosf(%a): libm %a %a, device source %a %a\n", x, sinf(x), cosf(x), sv, cv);
                }
            }
            uint64_t st = 88172645463325252ull + (uint64_t)t;
            for (long i = 0; i < 200000000L / (long)stride; ++i) {
                st ^= st << 13; st ^= st >> 7; st ^= st << 17;
                float y = __uint_as_float((uint32_t)st), x = __uint_as_float((uint32_t)(st >> 32));
                if (i & 1) { y = (float)((int32_t)(uint32_t)st) * (1.0f / 2147483648.0f); x = (float)((int32_t)(uint32_t)(st >> 32)) * (1.0f / 2147483648.0f); }
                ++l2;
                if (!same(atan2f(y, x), tr::ref_atan2f(y, x)) && d_at2++ < 3) std::printf("atan2f(%a, %a): libm %a, device source %a\n", y, x, atan2f(y, x), tr::ref_atan2f(y, x));
            }
            n_all += la; n_sc += ls; n_at2 += l2;
        });
        for (auto& t : th) t.join();
        const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, 1e-30f, -1e-30f, 1e30f, -1e30f, __builtin_huge_valf(), -__builtin_huge_valf(), __builtin_nanf(""), 0.5f, -2.0f, 1e-45f};
        for (float y : sp) for (float x : sp) { ++n_at2; if (!same(atan2f(y, x), tr::ref_atan2f(y, x)) && d_at2++ < 3) std::printf("atan2f(%a, %a) special\n", y, x); }
        std::printf("atanf: %ld, expf: %ld, logf: %ld of %ld bit patterns differ; sinf / cosf (shared form, |x| < 119): %ld of %ld; atan2f: %ld of %ld pairs\n",
                    d_atan.load(), d_exp.load(), d_log.load(), n_all.load(), d_sc.load(), n_sc.load(), d_at2.load(), n_at2.load());
        if (d_atan || d_exp || d_log || d_sc || d_at2) return 1;
    }
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
