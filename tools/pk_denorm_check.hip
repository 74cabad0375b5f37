// Do the packed f32 VALU ops (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32), which hipcc's SLP vectoriser emits for adjacent f32
// arithmetic, round exactly like their scalar forms on gfx950 -- subnormal operands / results included? (parity with an IEEE CPU
// restatement depends on it; also times them against the scalar forms)
//   hipcc -O3 --offload-arch=gfx950 tools/pk_denorm_check.hip -o tools/pk_denorm_check && gpurun -- tools/pk_denorm_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

__global__ void k(const float* a, const float* b, const float* c, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], z = c[i];
    float m1, s1, f1;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(x), "v"(y));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(x), "v"(z));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(x), "v"(y), "v"(z));
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 px = {x, x}, py = {y, y}, pz = {z, z}, m2, s2, f2;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m2) : "v"(px), "v"(py));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s2) : "v"(px), "v"(pz));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(f2) : "v"(px), "v"(py), "v"(pz));
    out[6 * i + 0] = __float_as_uint(m1); out[6 * i + 1] = __float_as_uint(m2.x);
    out[6 * i + 2] = __float_as_uint(s1); out[6 * i + 3] = __float_as_uint(s2.y);
    out[6 * i + 4] = __float_as_uint(f1); out[6 * i + 5] = __float_as_uint(f2.x);
}

int main() {
    const int n = 1 << 22;
    std::vector<float> a(n), b(n), c(n);
    std::mt19937 rng(5);
    auto rnd_bits = [&](int cls) {   // cls 0: any finite, 1: subnormal, 2: tiny normal, 3: ordinary
        uint32_t u = rng();
        if (cls == 1) u &= 0x807fffffu;
        else if (cls == 2) u = (u & 0x807fffffu) | ((1u + (rng() % 30u)) << 23);
        else if (cls == 3) u = (u & 0x807fffffu) | ((100u + (rng() % 56u)) << 23);
        else if ((u & 0x7f800000u) == 0x7f800000u) u &= 0xbfffffffu;
        float f; std::memcpy(&f, &u, 4); return f;
    };
    for (int i = 0; i < n; ++i) { int cls = i & 3; a[i] = rnd_bits(cls); b[i] = rnd_bits((i >> 2) & 3); c[i] = rnd_bits((i >> 4) & 3); }
    float *da, *db, *dc; uint32_t* dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, (size_t)n * 24);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
    std::vector<uint32_t> out((size_t)n * 6);
    hipMemcpy(out.data(), dout, (size_t)n * 24, hipMemcpyDeviceToHost);
    long bad[3] = {0, 0, 0}, cpu_bad[3] = {0, 0, 0}, sub[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        float r[3] = {a[i] * b[i], a[i] + c[i], __builtin_fmaf(a[i], b[i], c[i])};
        for (int j = 0; j < 3; ++j) {
            uint32_t s = out[6 * (size_t)i + 2 * j], p = out[6 * (size_t)i + 2 * j + 1], h; std::memcpy(&h, &r[j], 4);
            bool nan = (h & 0x7fffffffu) > 0x7f800000u;
            if (s != p && !nan) { if (bad[j] < 3) printf("op %d: a=%a b=%a c=%a scalar %08x packed %08x cpu %08x\n", j, a[i], b[i], c[i], s, p, h); bad[j]++; }
            if (s != h && !nan) cpu_bad[j]++;
            if ((h & 0x7f800000u) == 0 && (h & 0x7fffffu)) sub[j]++;
        }
    }
    printf("packed vs scalar mismatches (mul, add, fma): %ld %ld %ld of %d; scalar vs host IEEE: %ld %ld %ld; subnormal results: %ld %ld %ld\n",
           bad[0], bad[1], bad[2], n, cpu_bad[0], cpu_bad[1], cpu_bad[2], sub[0], sub[1], sub[2]);
    return 0;
}
