// Instruction-throughput micro-benchmark for gfx950 (torch-free, seconds on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_valu.hip -o tools/ubench_valu && gpurun -- tools/ubench_valu
// Each kernel runs 8 independent dependency chains of one operation per lane so that the SIMD's issue rate,
// not the latency, sets the time; results are printed as issue cycles per wave-instruction relative to
// v_fma_f32 (= 2 cycles on a SIMD-32, MI355X_MICROARCH.md) and as the measured time.
// Why: the PMC profile of k_path_tiles shows ~4 cycles per VALU instruction (nominal 2); this prices the
// operations the path tracer is made of (u32 multiplies of the counter hash, IEEE divide / sqrt, ocml libm).
// Round 4: the shader clock under the load is MEASURED (s_memtime ticks against the 100 MHz s_memrealtime, per kernel), every price
// is given in measured cycles, the plain and the packed f32 operations (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two f32 per lane
// and instruction) are timed at 1, 2, 4 and 8 waves per SIMD -- what the VALU peak of bench.py's roofline.valu follows from.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAINS 8
#define ITERS 2048

template <class Op>
__global__ __launch_bounds__(256) void k_bench(float* out, float seed, Op op, unsigned long long* clocks) {
    float v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = seed + (float)(threadIdx.x * CHAINS + c) * 1.0009765625f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
#pragma nounroll
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) v[c] = op(v[c]);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (clocks && threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = w1 - w0; }
    float s = 0.0f;
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    if (s == 12345.678f) out[threadIdx.x] = s;   // keep the chains alive
}
// the same for operations on a pair of f32 per lane (a 64-bit register pair)
typedef float f2v __attribute__((ext_vector_type(2)));
template <class Op>
__global__ __launch_bounds__(256) void k_bench2(float* out, float seed, Op op, unsigned long long* clocks) {
    f2v v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { v[c].x = seed + (float)(threadIdx.x * CHAINS + c) * 1.0009765625f; v[c].y = v[c].x * 0.75f; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
#pragma nounroll
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) v[c] = op(v[c]);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (clocks && threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = w1 - w0; }
    float s = 0.0f;
    for (int c = 0; c < CHAINS; ++c) s += v[c].x + v[c].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}
#define ASM2(name, text) struct name { __device__ f2v operator()(f2v x) const { f2v r; asm volatile(text : "=v"(r) : "v"(x)); return r; } }
ASM2(OpPkFma, "v_pk_fma_f32 %0, %1, %1, %1");
ASM2(OpPkMul, "v_pk_mul_f32 %0, %1, %1");
ASM2(OpPkAdd, "v_pk_add_f32 %0, %1, %1");
ASM2(OpPkMov, "v_pk_mov_b32 %0, %1, %1");
// the plain instruction twice on the two halves (what the packed form replaces)
struct OpMulTwice { __device__ f2v operator()(f2v x) const { f2v r; asm volatile("v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3" : "=&v"(r.x), "=&v"(r.y) : "v"(x.x), "v"(x.y)); return r; } };
struct OpPkMulC { __device__ f2v operator()(f2v x) const { return x * x; } };   // what the compiler makes of a 2-vector multiply

#define ASM1(name, text) struct name { __device__ float operator()(float x) const { float r; asm volatile(text : "=v"(r) : "v"(x)); return r; } }
ASM1(OpFma, "v_fma_f32 %0, %1, %1, %1");
ASM1(OpMulLo, "v_mul_lo_u32 %0, %1, %1");
ASM1(OpMulHi, "v_mul_hi_u32 %0, %1, %1");
ASM1(OpMul24, "v_mul_u32_u24 %0, %1, %1");
ASM1(OpMad24, "v_mad_u32_u24 %0, %1, %1, %1");
ASM1(OpRcp, "v_rcp_f32 %0, %1");
ASM1(OpSqrt, "v_sqrt_f32 %0, %1");
ASM1(OpRsq, "v_rsq_f32 %0, %1");
ASM1(OpExp, "v_exp_f32 %0, %1");
ASM1(OpLog, "v_log_f32 %0, %1");
ASM1(OpSin, "v_sin_f32 %0, %1");
ASM1(OpXor, "v_xor_b32 %0, %1, %1");
ASM1(OpAdd, "v_add_u32 %0, %1, %1");
ASM1(OpLshr, "v_lshrrev_b32 %0, 15, %1");
ASM1(OpBfrev, "v_bfrev_b32 %0, %1");
ASM1(OpCvtU, "v_cvt_f32_u32 %0, %1");
ASM1(OpMulF, "v_mul_f32 %0, %1, %1");
ASM1(OpAddF, "v_add_f32 %0, %1, %1");
ASM1(OpCnd, "v_cndmask_b32 %0, %1, %1, vcc");
ASM1(OpMax, "v_max_f32 %0, %1, %1");
ASM1(OpMov, "v_mov_b32 %0, %1");
struct OpDiv { __device__ float operator()(float x) const { return 1.0009f / x; } };           // IEEE f32 divide (correctly rounded, hipcc default)
struct OpDiv2 { __device__ float operator()(float x) const { return x / 1.0009f; } };          // divide by a constant (still a divide: no reciprocal substitution)
struct OpSqrtF { __device__ float operator()(float x) const { return sqrtf(x); } };            // IEEE sqrt
struct OpSinF { __device__ float operator()(float x) const { return sinf(x); } };
struct OpCosF { __device__ float operator()(float x) const { return cosf(x); } };
struct OpAcosF { __device__ float operator()(float x) const { return acosf(x * 1e-3f); } };
struct OpAtan2F { __device__ float operator()(float x) const { return atan2f(x, 1.5f); } };
struct OpExpF { __device__ float operator()(float x) const { return expf(-x * 1e-3f); } };
struct OpLogF { __device__ float operator()(float x) const { return logf(x); } };
struct OpPowF { __device__ float operator()(float x) const { return powf(x, 0.4166667f); } };
struct OpFloor { __device__ float operator()(float x) const { return floorf(x * 0.99f); } };
struct OpMix32 {   // one mix32 of TRAY-CBRNG (2 u32 multiplies, 3 shift-xor)
    __device__ float operator()(float xf) const {
        uint32_t x = __float_as_uint(xf);
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        return __uint_as_float(x);
    }
};
struct OpF64Fma { __device__ float operator()(float x) const { double d = (double)x; d = d * d + d; return (float)d; } };
struct OpNorm {   // normalized(): 3 mul/add, sqrt, 3 divides
    __device__ float operator()(float x) const {
        float y = x * 0.5f, z = x * 0.25f;
        float l = sqrtf(x * x + y * y + z * z);
        return x / l + y / l + z / l;
    }
};

struct Measured { double ms, ghz; };
// waves_per_simd: 1, 2, 4, 8 (workgroups of 4 waves per CU)
template <class Op, bool PAIR>
static Measured measure(Op op, int waves_per_simd) {
    float* d = nullptr;
    hipMalloc(&d, 1024);
    const int blocks = 256 * waves_per_simd;
    unsigned long long* dc = nullptr;
    hipMalloc(&dc, (size_t)blocks * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&] {
        if constexpr (PAIR) hipLaunchKernelGGL(k_bench2<Op>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, op, dc);
        else hipLaunchKernelGGL(k_bench<Op>, dim3(blocks), dim3(256), 0, 0, d, 1.5f, op, dc);
    };
    go();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        go();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> hc((size_t)blocks * 2);
    hipMemcpy(hc.data(), dc, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double ticks = 0, wall = 0;
    for (int b = 0; b < blocks; ++b) { ticks += (double)hc[2 * b]; wall += (double)hc[2 * b + 1]; }
    hipFree(d); hipFree(dc);
    return Measured{(double)best, wall > 0 ? ticks / wall * 0.1 : 0.0};   // s_memrealtime counts at 100 MHz
}
static double g_ref_ms = 0.0, g_ghz = 2.4;
template <class Op, bool PAIR = false>
static double run(const char* name, Op op, int waves_per_simd = 8) {
    const Measured m = measure<Op, PAIR>(op, waves_per_simd);
    // wave-instructions per SIMD: waves_per_simd waves, each ITERS * CHAINS calls
    const double calls_per_simd = (double)waves_per_simd * ITERS * CHAINS;
    const double ns_per_call = m.ms * 1e6 / calls_per_simd;
    printf("%-26s %d waves/SIMD %8.3f ms  %6.3f ns/call/SIMD  shader clock %.3f GHz (s_memtime / s_memrealtime)  = %5.2f cycles per wave-instruction",
           name, waves_per_simd, m.ms, ns_per_call, m.ghz, ns_per_call * m.ghz);
    if (g_ref_ms > 0.0 && waves_per_simd == 8) printf("  = %5.2f x v_fma_f32", m.ms / g_ref_ms);
    printf("\n");
    return m.ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s, %d CUs, clock %d MHz (hipDeviceProp)\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    printf("-- plain and packed f32, by occupancy (8 independent chains per lane; a packed call processes TWO f32 per lane)\n");
    for (int w : {1, 2, 4, 8}) {
        run("v_fma_f32", OpFma(), w); run("v_mul_f32", OpMulF(), w); run("v_add_f32", OpAddF(), w);
        run<OpPkFma, true>("v_pk_fma_f32", OpPkFma(), w); run<OpPkMul, true>("v_pk_mul_f32", OpPkMul(), w); run<OpPkAdd, true>("v_pk_add_f32", OpPkAdd(), w);
        run<OpMulTwice, true>("2 x v_mul_f32 (pair)", OpMulTwice(), w); run<OpPkMulC, true>("float2 * float2 (compiler)", OpPkMulC(), w);
        run<OpPkMov, true>("v_pk_mov_b32", OpPkMov(), w);
    }
    printf("-- instruction prices at 8 waves per SIMD\n");
    g_ref_ms = run("v_fma_f32", OpFma());
#define R(name, Op) run(name, Op())
    R("v_mul_f32", OpMulF); R("v_add_f32", OpAddF); R("v_max_f32", OpMax); R("v_mov_b32", OpMov);
    R("v_xor_b32", OpXor); R("v_add_u32", OpAdd); R("v_lshrrev_b32", OpLshr); R("v_cndmask_b32", OpCnd); R("v_bfrev_b32", OpBfrev); R("v_cvt_f32_u32", OpCvtU);
    R("v_mul_lo_u32", OpMulLo); R("v_mul_hi_u32", OpMulHi); R("v_mul_u32_u24", OpMul24); R("v_mad_u32_u24", OpMad24);
    R("v_rcp_f32", OpRcp); R("v_sqrt_f32", OpSqrt); R("v_rsq_f32", OpRsq); R("v_exp_f32", OpExp); R("v_log_f32", OpLog); R("v_sin_f32", OpSin);
    R("f32 divide (IEEE)", OpDiv); R("f32 divide by const", OpDiv2); R("sqrtf (IEEE)", OpSqrtF); R("floorf", OpFloor);
    R("sinf (ocml)", OpSinF); R("cosf (ocml)", OpCosF); R("acosf (ocml)", OpAcosF); R("atan2f (ocml)", OpAtan2F);
    R("expf (ocml)", OpExpF); R("logf (ocml)", OpLogF); R("powf (ocml)", OpPowF);
    R("mix32 (CBRNG)", OpMix32); R("f64 fma + cvt", OpF64Fma); R("normalized(f3)", OpNorm);
    return 0;
}
