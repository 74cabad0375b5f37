// TEST INFRASTRUCTURE ONLY. Host shim that lets g++ compile the DEVICE source of tray_rust_amd/csrc/hip (the same files hipcc
// compiles for gfx950) so that device code can be checked against the oracle without a GPU. Two execution modes:
//
//   hip_emu::launch        every thread is its own wave of ONE lane (ballot = own bit, shuffles return the own value, any / all =
//                          the predicate); the threads of a block run one after the other to completion. Exact for kernels whose
//                          threads only meet through atomics and queues (traversal kernels, debug kernels). Fast.
//   hip_emu::launch_simt   every thread of a block is a FIBER (ucontext) on one OS thread; wave intrinsics (ballot, any, all,
//                          shuffles, wave_barrier) are rendezvous of the live lanes of a 64-lane wave, __syncthreads a rendezvous
//                          of the live threads of the block, __shared__ variables are shared by the block (blocks run one at a
//                          time). Between rendezvous the lanes run one after the other, so code is exact here iff every LDS /
//                          global exchange between lanes is separated by one of those intrinsics -- which is also what makes it
//                          independent of lockstep execution on the device. A rendezvous that can never complete (an intrinsic
//                          reached by only part of a wave while the rest waits elsewhere) is reported, not hung on.
//
// Nothing in the product includes this file.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define TR_HOST_EMU 1
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__ static
#define __shared__ static
#define __launch_bounds__(...)
#define TR_DYN_LDS(T, name) T* const name = reinterpret_cast<T*>(hip_emu::dyn_lds())
#ifdef TR_EMU_PROFILE
namespace hip_emu { static uint32_t prof_phase_of[1024]; }   // per fiber: the lanes of a wave run one after the other between two rendezvous
#define TR_EMU_PHASE(k) (hip_emu::prof_phase_of[threadIdx.x & 1023u] = (uint32_t)(k))   // calls made in different phases of a segment are not executed together
#else
#define TR_EMU_PHASE(k) ((void)0)
#endif

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
static hip_emu::Idx threadIdx, blockIdx;   // (one OS thread: the fiber scheduler rewrites threadIdx at every switch)
static dim3 blockDim, gridDim;

namespace hip_emu {

constexpr uint32_t WAVE = 64, MAX_THREADS = 1024, FIBER_STACK = 1u << 20;

struct Wave {
    uint32_t live = 0, arrived = 0;
    uint64_t gen = 0;
    uint64_t slots[2][WAVE];
    uint64_t present[2] = {0, 0};
};
// Fiber switch. On x86-64 a hand-written switch of the callee-saved registers and the stack pointer: swapcontext() makes two
// rt_sigprocmask system calls per switch, and a SIMT-emulated kernel switches at every wave intrinsic of every lane (the CPU
// suite spent more time in the kernel than in the tests). Other hosts keep ucontext.
#if defined(__x86_64__)
#define HIP_EMU_ASM_SWITCH 1
extern "C" void hip_emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .p2align 4
    .type hip_emu_switch,@function
hip_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hip_emu_switch, .-hip_emu_switch
)");
#endif
struct Fiber {
#ifdef HIP_EMU_ASM_SWITCH
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr;
    bool done = false;
};
struct Block {
    bool simt = false;
    uint32_t n_threads = 0, current = 0, live = 0, arrived = 0;
    uint64_t gen = 0, progress = 0;
    Wave waves[MAX_THREADS / WAVE];
    Fiber fibers[MAX_THREADS];
#ifdef HIP_EMU_ASM_SWITCH
    void* sched_sp = nullptr;
#else
    ucontext_t sched;
#endif
    std::function<void()> body;
    std::vector<uint64_t> lds;   // dynamic LDS of the block
};
inline Block& block() { static Block b; return b; }
inline void* dyn_lds() { return block().lds.data(); }

// ---- divergence profile (builds with -DTR_EMU_PROFILE -finstrument-functions, tools/divergence_profile.py): between two
// rendezvous a lane runs one SEGMENT; a function the lanes of a wave call in the same segment would be executed together on
// the device, as often as the lane that calls it most often. prof_lane_done() closes the calling lane's segment,
// prof_wave_done() the wave's.
#ifdef TR_EMU_PROFILE
void prof_lane_done(uint32_t wave);
void prof_wave_done(uint32_t wave);
#else
inline void prof_lane_done(uint32_t) {}
inline void prof_wave_done(uint32_t) {}
#endif

#ifdef HIP_EMU_ASM_SWITCH
inline void yield() { Block& b = block(); hip_emu_switch(&b.fibers[b.current].sp, b.sched_sp); }
#else
inline void yield() { Block& b = block(); swapcontext(&b.fibers[b.current].ctx, &b.sched); }
#endif

// all live lanes of the calling lane's wave contribute `mine`; returns after every one of them has arrived
inline void wave_exchange(uint64_t mine, uint64_t* all, uint64_t& present) {
    Block& b = block();
    const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1);
    Wave& w = b.waves[tid / WAVE];
    const uint64_t my_gen = w.gen;
    const int buf = (int)(my_gen & 1u);
    if (w.arrived == 0) w.present[buf] = 0;
    w.slots[buf][lane] = mine;
    w.present[buf] |= 1ull << lane;
    prof_lane_done(tid / WAVE);
    if (++w.arrived == w.live) { w.arrived = 0; ++w.gen; ++b.progress; prof_wave_done(tid / WAVE); }
    else while (w.gen == my_gen) yield();
    std::memcpy(all, w.slots[buf], sizeof w.slots[buf]);
    present = w.present[buf];
}
inline void block_barrier() {
    Block& b = block();
    const uint64_t my_gen = b.gen;
    prof_lane_done(threadIdx.x / WAVE);
    if (++b.arrived == b.live) { b.arrived = 0; ++b.gen; ++b.progress; }
    else while (b.gen == my_gen) yield();
}
inline void fiber_exit() {   // a thread that leaves the kernel no longer takes part in rendezvous (inactive lanes on the device)
    Block& b = block();
    Wave& w = b.waves[threadIdx.x / WAVE];
    --w.live; --b.live; ++b.progress;
    prof_lane_done(threadIdx.x / WAVE);
    if (w.live == 0 || w.arrived == w.live) prof_wave_done(threadIdx.x / WAVE);
    if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; ++w.gen; }
    if (b.live > 0 && b.arrived == b.live) { b.arrived = 0; ++b.gen; }
    b.fibers[b.current].done = true;
}
inline void fiber_main() {
    block().body();
    fiber_exit();
    yield();   // never resumed
}

template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8, "shuffle of a value wider than 8 bytes"); uint64_t u = 0; std::memcpy(&u, &v, sizeof(T)); return u; }
template <class T> inline T from_bits(uint64_t u) { T v; std::memcpy(&v, &u, sizeof(T)); return v; }

// one-lane launch: threads run to completion one after the other
template <class K>
inline void launch(uint32_t blocks, uint32_t threads, K&& kernel, size_t dyn_lds_bytes = 1u << 20) {
    Block& b = block();
    b.simt = false;
    b.lds.assign((dyn_lds_bytes + 7) / 8, 0u);
    gridDim.x = blocks; blockDim.x = threads;
    for (uint32_t bi = 0; bi < blocks; ++bi)
        for (uint32_t t = 0; t < threads; ++t) { blockIdx.x = bi; threadIdx.x = t; kernel(); }
}

// SIMT launch: the threads of a block are fibers that meet at wave / block intrinsics; blocks run one at a time
template <class K>
inline int launch_simt(uint32_t blocks, uint32_t threads, K&& kernel, size_t dyn_lds_bytes = 1u << 20) {
    Block& b = block();
    if (threads > MAX_THREADS || threads % WAVE != 0) return -1;
    gridDim.x = blocks; blockDim.x = threads;
    b.body = kernel;
    b.lds.assign((dyn_lds_bytes + 7) / 8, 0u);
    for (uint32_t t = 0; t < threads; ++t)
        if (!b.fibers[t].stack) b.fibers[t].stack = static_cast<char*>(std::malloc(FIBER_STACK));
    int rc = 0;
    for (uint32_t bi = 0; bi < blocks && rc == 0; ++bi) {
        blockIdx.x = bi;
        b.simt = true; b.n_threads = threads; b.live = threads; b.arrived = 0; b.gen = 0; b.progress = 0;
        for (uint32_t wv = 0; wv < threads / WAVE; ++wv) { b.waves[wv] = Wave(); b.waves[wv].live = WAVE; }
        for (uint32_t t = 0; t < threads; ++t) {
            Fiber& f = b.fibers[t];
            f.done = false;
#ifdef HIP_EMU_ASM_SWITCH
            // a fresh stack as hip_emu_switch expects it: six callee-saved registers, then the address it returns to; fiber_main
            // starts with the alignment of a called function (rsp = 16 n + 8) and never returns
            void** sp = reinterpret_cast<void**>(reinterpret_cast<uintptr_t>(f.stack + FIBER_STACK) & ~uintptr_t(15));
            *--sp = nullptr;
            *--sp = reinterpret_cast<void*>(&fiber_main);
            for (int r = 0; r < 6; ++r) *--sp = nullptr;
            f.sp = sp;
#else
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = FIBER_STACK; f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())fiber_main, 0);
#endif
        }
        uint64_t last_progress = 0;
        uint32_t idle_rounds = 0;
        while (b.live > 0) {
            for (uint32_t t = 0; t < threads; ++t) {
                if (b.fibers[t].done) continue;
                b.current = t; threadIdx.x = t;
#ifdef HIP_EMU_ASM_SWITCH
                hip_emu_switch(&b.sched_sp, b.fibers[t].sp);
#else
                swapcontext(&b.sched, &b.fibers[t].ctx);
#endif
            }
            if (b.progress == last_progress) {
                if (++idle_rounds > 4) {   // every live fiber waits at a rendezvous that cannot complete
                    std::fprintf(stderr, "hip_emu: deadlock in block %u: a wave / block intrinsic was reached by only part of its live lanes\n", bi);
                    for (uint32_t wv = 0; wv < threads / WAVE; ++wv)
                        std::fprintf(stderr, "  wave %u: %u live, %u arrived\n", wv, b.waves[wv].live, b.waves[wv].arrived);
                    std::fprintf(stderr, "  block barrier: %u live, %u arrived\n", b.live, b.arrived);
                    rc = -3;
                    break;
                }
            } else { idle_rounds = 0; last_progress = b.progress; }
        }
    }
    b.simt = false;
    return rc;
}

}  // namespace hip_emu

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

inline unsigned long long __ballot(int p) {
    if (!hip_emu::block().simt) return p ? (1ull << (threadIdx.x & 63u)) : 0ull;
    uint64_t all[hip_emu::WAVE], present, m = 0;
    hip_emu::wave_exchange(p ? 1u : 0u, all, present);
    for (uint32_t l = 0; l < hip_emu::WAVE; ++l) if (((present >> l) & 1u) && all[l]) m |= 1ull << l;
    return m;
}
inline int __any(int p) { return __ballot(p) != 0ull; }
inline int __all(int p) { return __ballot(!p) == 0ull; }
template <class T> inline T __shfl(T v, int src) {
    if (!hip_emu::block().simt) return v;
    uint64_t all[hip_emu::WAVE], present;
    hip_emu::wave_exchange(hip_emu::to_bits(v), all, present);
    const uint32_t s = (uint32_t)src & 63u;
    return ((present >> s) & 1u) ? hip_emu::from_bits<T>(all[s]) : v;
}
template <class T> inline T __shfl_xor(T v, int mask) {
    if (!hip_emu::block().simt) return v;
    uint64_t all[hip_emu::WAVE], present;
    hip_emu::wave_exchange(hip_emu::to_bits(v), all, present);
    const uint32_t s = ((threadIdx.x & 63u) ^ (uint32_t)mask) & 63u;
    return ((present >> s) & 1u) ? hip_emu::from_bits<T>(all[s]) : v;
}
template <class T> inline T __shfl_down(T v, int delta) {
    if (!hip_emu::block().simt) return T(0);   // lane-0 reductions only: the other lanes of a one-lane wave do not exist
    uint64_t all[hip_emu::WAVE], present;
    hip_emu::wave_exchange(hip_emu::to_bits(v), all, present);
    const uint32_t s = (threadIdx.x & 63u) + (uint32_t)delta;
    if (s >= hip_emu::WAVE) return v;
    return ((present >> s) & 1u) ? hip_emu::from_bits<T>(all[s]) : T(0);   // a lane that left the kernel contributes nothing
}
inline void __builtin_amdgcn_wave_barrier() {
    if (!hip_emu::block().simt) return;
    uint64_t all[hip_emu::WAVE], present;
    hip_emu::wave_exchange(0u, all, present);
}
inline void __syncthreads() { if (hip_emu::block().simt) hip_emu::block_barrier(); }
inline void __threadfence_block() {}   // (fibers run one at a time: every store is visible at once)
template <class T> inline T atomicAdd(T* p, T v) { T old = *p; *p = old + v; return old; }
inline uint32_t atomicAdd(uint32_t* p, int v) { uint32_t old = *p; *p = old + (uint32_t)v; return old; }
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_AGENT 1
template <class T> inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T, class V> inline T __hip_atomic_fetch_add(T* p, V v, int, int) { T old = *p; *p = old + (T)v; return old; }
inline long long clock64() { return 0; }
