// What does a traversal step's fetch cost on gfx950?  (torch-free, seconds on the GPU box)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o tools/ubench_gather && gpurun -- tools/ubench_gather
// Every lane chases pointers through a table of 64-byte records (the size of the two 32-byte boxes a step of k_wf_trace_dyn
// reads): the next record's index comes out of the record just read, as a child index comes out of a node. Three ways to fetch:
//   lane   each lane reads its own record with four 16-byte loads (what the traversal does: 4 instructions x <= 64 lines)
//   quad   the four lanes of a quad read one record together, 16 bytes each (one instruction covers 16 records), and the
//          records are handed to their owners through LDS -- 4x fewer (instruction, line) pairs for the L1's tagger,
//          the same bytes
//   half   `lane` with 32-byte records (one box per step)
//   wide   `lane` with 128-byte records (eight 16-byte loads: what a step over a 4-wide node would fetch)
// over tables of 2 MB (L2 resident) .. 512 MB (HBM), with all 64 lanes of a wave active or only every third one (the
// traversal kernel runs at 37 % of its lanes), at 4 waves per SIMD. Printed: G records/s and ns per dependent step.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256, 4) void k_chase(const uint4* __restrict__ table, uint32_t mask, int steps, int sparse, uint32_t* out) {
    __shared__ uint4 s_rec[4][64][4];   // [wave][ray][part]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
    const bool active = !sparse || (lane % 3u) == 0u;
    uint32_t acc = 0u;
    for (int s = 0; s < steps; ++s) {
        uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a, d = a;
        if (MODE == 0) {
            if (active) { const uint4* p = table + (size_t)idx * 4u; a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
        } else if (MODE == 2) {
            if (active) { const uint4* p = table + (size_t)idx * 2u; a = p[0]; b = p[1]; }
        } else if (MODE == 3) {
            if (active) {
                const uint4* p = table + (size_t)idx * 8u;
                const uint4 e = p[4], f = p[5], g = p[6], h = p[7];
                a = p[0]; b = p[1]; c = p[2]; d = p[3];
                a.x ^= e.x; b.x ^= f.x; c.x ^= g.x; d.x ^= h.x;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int src = 16 * i + (int)(lane >> 2);
                const uint32_t ridx = __shfl(idx, src);
                const bool ract = !sparse || (src % 3) == 0;
                if (ract) s_rec[wave][src][lane & 3u] = table[(size_t)ridx * 4u + (lane & 3u)];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (active) { a = s_rec[wave][lane][0]; b = s_rec[wave][lane][1]; c = s_rec[wave][lane][2]; d = s_rec[wave][lane][3]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        acc += a.x ^ b.y ^ c.z ^ d.w;
        if (active) idx = (a.x + b.x + c.x + d.x + (uint32_t)s) & mask;
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main() {
    const int cus = 256, grid = cus * 4, steps = 512;
    uint32_t* d_out; CHECK(hipMalloc(&d_out, 4));
    const char* names[4] = {"lane", "quad", "half", "wide"};
    for (uint32_t mb : {2u, 32u, 512u}) {
        const uint32_t n = mb * (1u << 20) / 64u;   // records
        std::vector<uint4> h((size_t)n * 4u);
        for (size_t i = 0; i < h.size(); ++i) { uint32_t r = (uint32_t)i * 2654435761u; r ^= r >> 15; r *= 0x2c1b3c6du; r ^= r >> 12; h[i] = make_uint4(r, r * 3u, r * 5u, r * 7u); }
        uint4* d_t; CHECK(hipMalloc(&d_t, h.size() * 16u));
        CHECK(hipMemcpy(d_t, h.data(), h.size() * 16u, hipMemcpyHostToDevice));
        for (int sparse = 0; sparse < 2; ++sparse)
            for (int mode = 0; mode < 4; ++mode) {
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0));
                    const uint32_t m = mode == 2 ? 2u * n - 1u : (mode == 3 ? n / 2u - 1u : n - 1u);
                    if (mode == 0) hipLaunchKernelGGL(k_chase<0>, dim3(grid), dim3(256), 0, 0, d_t, m, steps, sparse, d_out);
                    else if (mode == 1) hipLaunchKernelGGL(k_chase<1>, dim3(grid), dim3(256), 0, 0, d_t, m, steps, sparse, d_out);
                    else if (mode == 2) hipLaunchKernelGGL(k_chase<2>, dim3(grid), dim3(256), 0, 0, d_t, m, steps, sparse, d_out);
                    else hipLaunchKernelGGL(k_chase<3>, dim3(grid), dim3(256), 0, 0, d_t, m, steps, sparse, d_out);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                const double lanes = (double)grid * 256.0 * (sparse ? 22.0 / 64.0 : 1.0);
                printf("table %4u MB  %s  %-5s  %8.3f ms  %7.2f G records/s  %7.1f ns per step of a wave\n", mb, sparse ? "22/64 lanes" : "64/64 lanes",
                       names[mode], best, lanes * steps / best / 1e6, best * 1e6 / steps);
            }
        CHECK(hipFree(d_t));
    }
    return 0;
}
