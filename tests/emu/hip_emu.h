// TEST INFRASTRUCTURE ONLY. Host shim that lets g++ compile the DEVICE source of tray_rust_amd/csrc/hip (the same files hipcc
// compiles for gfx950) so that per-lane device code can be checked against the oracle without a GPU. Semantics: every thread
// is its own wave of ONE lane (ballot = own bit, shuffles return the own value, any / all = the predicate), threads of a block
// run one after the other to completion. That is exact for kernels whose threads only meet through atomics and queues (the
// traversal kernels, the debug kernels); kernels that cooperate through LDS and barriers (k_path_tiles, k_wf_advance,
// mesh_leaf_coop) are NOT emulated. Nothing in the product includes this file.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#define TR_HOST_EMU 1
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__ static
#define __shared__
#define __launch_bounds__(...)

using std::isfinite; using std::isinf; using std::isnan; using std::max; using std::min;   // global in HIP device code

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct dim3 { uint32_t x = 1, y = 1, z = 1; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

namespace hip_emu { struct Idx { uint32_t x = 0, y = 0, z = 0; }; }
static thread_local hip_emu::Idx threadIdx, blockIdx;
static thread_local dim3 blockDim, gridDim;

inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline uint32_t __brev(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
}
// one-lane waves
inline unsigned long long __ballot(int p) { return p ? (1ull << (threadIdx.x & 63u)) : 0ull; }
inline int __any(int p) { return p; }
inline int __all(int p) { return p; }
template <class T> inline T __shfl(T v, int) { return v; }
template <class T> inline T __shfl_xor(T v, int) { return v; }
template <class T> inline T __shfl_down(T, int) { return T(0); }   // used by lane-0 reductions only: the other lanes of the wave do not exist
inline void __syncthreads() {}
inline void __builtin_amdgcn_wave_barrier() {}
template <class T> inline T atomicAdd(T* p, T v) { T old = *p; *p = old + v; return old; }
inline uint32_t atomicAdd(uint32_t* p, int v) { uint32_t old = *p; *p = old + (uint32_t)v; return old; }
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
template <class T, class V> inline T __hip_atomic_fetch_add(T* p, V v, int, int) { T old = *p; *p = old + (T)v; return old; }
inline long long clock64() { return 0; }
