// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access pattern of register spills (scratch_store / scratch_load of
// one dword per lane): a kernel that moves a KNOWN number of bytes through a private array too large for registers.
//   hipcc -O3 --offload-arch=gfx950 tools/scratch_calib.hip -o tools/scratch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/scratch_calib   (and WRITE_SIZE in its own pass)
// Each lane owns WORDS dwords of scratch (dynamic, wave-uniform indexing keeps them out of registers); per round it stores all of them and then
// loads all of them. Working set = waves in flight x 64 x WORDS x 4 B; with 12 waves per CU and WORDS = 128 that is 100 MB on
// 256 CUs: beyond the 32 MB of L2, inside the 256 MB Infinity Cache -- like the tile kernel's spill working set.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#ifndef WORDS
#define WORDS 128
#endif

__global__ __launch_bounds__(256) void k_scratch(float* out, int rounds, int stride) {
    float priv[WORDS];
    float acc = 0.0f;
    int idx = (blockIdx.x * 7 + stride) % WORDS;   // wave-uniform like the offset of a spill slot: each access of a wave is 64 x 4 contiguous bytes
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < WORDS; ++k) { priv[idx] = acc + (float)k; idx = (idx + stride) % WORDS; }   // WORDS stores per lane
        for (int k = 0; k < WORDS; ++k) { acc += priv[idx]; idx = (idx + stride) % WORDS; }             // WORDS loads per lane
    }
    if (acc == 12345.0f) out[threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 64;
    const int blocks = 256 * 3;   // 3 workgroups of 4 waves per CU
    float* d = nullptr;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_scratch, dim3(blocks), dim3(256), 0, 0, d, 1, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_scratch, dim3(blocks), dim3(256), 0, 0, d, rounds, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double lanes = (double)blocks * 256.0;
    const double bytes_each_way = lanes * WORDS * 4.0 * rounds;
    printf("scratch_calib: %d workgroups x 256 lanes x %d dwords x %d rounds: %.6g bytes stored, %.6g bytes loaded (timed launch; the warm-up launch adds 1/%d of that), "
           "working set %.1f MB, %.3f ms, %.1f GB/s each way\n", blocks, WORDS, rounds, bytes_each_way, bytes_each_way, rounds, lanes * WORDS * 4.0 / 1e6, ms,
           bytes_each_way / (ms * 1e-3) / 1e9);
    return 0;
}
