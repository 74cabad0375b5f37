// Is 1.0f / x -- the correctly rounded quotient the compiler builds from v_div_scale / v_rcp / five v_fma / v_div_fmas / v_div_fixup (~11 VALU instructions,
// 42 cycles: profiles/r04_ubench_valu.txt) -- reproduced by a shorter sequence on EVERY one of the 2^32 arguments?  Candidates:
//   A  v_rcp_f32 + one Newton step in fma form   (3 instructions)
//   B  v_rcp_f32 + two Newton steps              (5)
//   C  B + v_div_fixup_f32                       (6)
// For each: the number of arguments whose result differs in any bit from 1.0f / x, and the exponent range of |x| outside of which all mismatches lie.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_rcp.hip -o /tmp/ubench_rcp && /tmp/ubench_rcp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

struct Tally { unsigned long long bad[3]; unsigned int lo_exp[3], hi_exp[3]; unsigned long long bad_mid[3]; };

__device__ inline float newton(float x, float r) { const float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ inline bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

__global__ void k(unsigned int hi, Tally* out) {
    const uint32_t bits = hi << 24 | (blockIdx.x * blockDim.x + threadIdx.x);
    const float x = __uint_as_float(bits);
    const float ref = 1.0f / x;
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float a = newton(x, r0);
    const float b = newton(x, a);
    const float c = __builtin_amdgcn_div_fixupf(b, x, 1.0f);
    const float cand[3] = {a, b, c};
    const uint32_t ex = (bits >> 23) & 255u;
    for (int i = 0; i < 3; ++i)
        if (!same(cand[i], ref)) {
            atomicAdd(&out->bad[i], 1ull);
            if (ex >= 127u - 64u && ex <= 127u + 64u) atomicAdd(&out->bad_mid[i], 1ull);   // 2^-64 <= |x| < 2^65
            if (ex < 127u) atomicMax(&out->lo_exp[i], ex); else atomicMin(&out->hi_exp[i], ex);
        }
}

int main() {
    Tally* d; Tally h;
    std::memset(&h, 0, sizeof h);
    for (int i = 0; i < 3; ++i) { h.lo_exp[i] = 0u; h.hi_exp[i] = 255u; }
    hipMalloc(&d, sizeof h); hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
    for (unsigned int hi = 0; hi < 256; ++hi) hipLaunchKernelGGL(k, dim3(1u << 16), dim3(256), 0, 0, hi, d);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* name[3] = {"A rcp + 1 Newton step (3 instr)", "B rcp + 2 Newton steps (5 instr)", "C B + v_div_fixup (6 instr)"};
    for (int i = 0; i < 3; ++i)
        std::printf("%-34s differs from 1.0f / x on %llu of 2^32 arguments; %llu of them with 2^-64 <= |x| < 2^65; mismatches below 1: biased exponent <= %u, at or above 1: >= %u\n",
                    name[i], h.bad[i], h.bad_mid[i], h.lo_exp[i], h.hi_exp[i]);
    return 0;
}
