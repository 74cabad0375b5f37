// What does a queue append cost on MI355X? Same-address device-scope atomics (one per wave, as wf_enqueue issues them) against
// one per workgroup and against appends spread over S counters.   hipcc -O3 --offload-arch=gfx950 tools/ubench_atomics.hip -o tools/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_wave(uint32_t* ctr, uint32_t* out, uint32_t segs) {   // one atomic per wave (with return, like a queue append)
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(ctr + ((blockIdx.x * 4 + (threadIdx.x >> 6)) % segs) * 32, 64u);
    base = __shfl(base, 0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = base + lane;
}
__global__ void k_wg(uint32_t* ctr, uint32_t* out, uint32_t segs) {   // one atomic per workgroup
    __shared__ uint32_t s_base;
    if (threadIdx.x == 0) s_base = atomicAdd(ctr + (blockIdx.x % segs) * 32, 256u);
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s_base + threadIdx.x;
}
__global__ void k_none(uint32_t* ctr, uint32_t* out, uint32_t segs) { out[blockIdx.x * blockDim.x + threadIdx.x] = blockIdx.x * blockDim.x + threadIdx.x; }
int main() {
    const uint32_t n_wg = 19500;
    uint32_t *ctr, *out;
    hipMalloc(&ctr, 4096 * 32 * 4); hipMalloc(&out, (size_t)n_wg * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, void (*k)(uint32_t*, uint32_t*, uint32_t), uint32_t segs) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(ctr, 0, 4096 * 32 * 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(n_wg), dim3(256), 0, 0, ctr, out, segs);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-10s segments %4u: %8.1f us for %u workgroups of 256 (%.1f ns per atomic)\n", name, segs, best * 1e3f, n_wg,
               best * 1e6f / (k == k_wave ? n_wg * 4.0f : (float)n_wg));
    };
    run("none", k_none, 1);
    for (uint32_t s : {1u, 8u, 64u, 512u}) run("per wave", k_wave, s);
    for (uint32_t s : {1u, 8u, 64u}) run("per wg", k_wg, s);
    return 0;
}
