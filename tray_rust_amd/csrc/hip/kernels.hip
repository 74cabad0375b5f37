// gfx950 kernels of libtrayhip.so (the device-side C ABI that launches them: device_api.hip; which instantiations exist: kernel_list.h).
//
// k_path_tiles — the tile worker (exec/multithreaded.rs:72-114) as a persistent-threads kernel:
//   * a workgroup (4 x wave64) owns one 8x8 tile at a time, pulled from a global atomic counter in
//     Morton order (block_queue.rs:52-59); lane l of every wave owns pixel l of the tile, wave w
//     takes the samples s = w, w+4, ... of that pixel;
//   * paths are regenerated in place: a lane whose path ended splats its sample and starts the next
//     one while its neighbours keep bouncing, so waves stay full without a global ray queue;
//   * the film lives in LDS while the tile is rendered: a 17x17 RGBW window (tile + 4 px filter halo)
//     updated with ds_add_f32 (RenderTarget::write, render_target.rs:77-165), flushed once per tile
//     with global f32 atomics into the caller's RGBW buffer.
#ifndef TR_HOST_EMU   // tests/emu compiles the device code of this file for the host (single-lane semantics) behind its own shim
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <string>
#include <type_traits>
#include <utility>
#include <algorithm>
#include <vector>

#include "../../../include/trayhip.h"
#include "../host/validate.hpp"
#include "dev_integrator.h"
#include "dev_whitted.h"

namespace trayh { void set_error(const std::string& msg); }
using trayh::set_error;
using namespace tr;

#ifndef TR_MIN_WAVES
#define TR_MIN_WAVES 3
#endif
#ifndef TR_MIN_WAVES_ANIM   // waves per SIMD the instantiations for moving scenes are compiled for (they carry the two-level traversal and the cached transforms)
#define TR_MIN_WAVES_ANIM 3
#endif
#ifndef TR_MIN_WAVES_SIDE   // the every-lobe instantiations (textured / GGX materials) and Whitted. At 2 (256 VGPRs) the per-hit copy of a textured material and
#define TR_MIN_WAVES_SIDE 3 // Whitted's locals stop spilling (105 - 145 spilled VGPRs -> 0 - 3) and the kernels get SLOWER: textured_box 646 -> 597, smallpt Whitted 3884 -> 3324
#endif                      // Msamples/s (profiles/r05_side_instantiations_ab.txt): the third wave per SIMD is worth more than the scratch traffic costs
#define WIN_MAX 17          // 8 + 2*4 + 1 window columns/rows
#define WIN_STRIDE 24       // row stride in floats: 4 rows of 8 lanes land on 32 distinct banks
#define WIN_PLANE (WIN_MAX * WIN_STRIDE)

struct DevStats { unsigned long long samples, vertices, rays, trav[18]; };   // trav[stage * 6 + k]: traversal counters of builds with -DWF_TRACE_STATS

// RenderTarget::write for one sample into the LDS window (render_target.rs:118-146).
// win origin = (x0 - fpw, y0 - fph); ranges already clipped to the image.
__device__ __forceinline__ void film_splat(const DevScene& sc, float* __restrict__ s_win, const float* __restrict__ s_table,
                                           int x0, int y0, float sx, float sy, f3 c) {
    const int fpw = sc.fpw, fph = sc.fph;
    const int xr0 = max(x0 - fpw, 0), xr1 = min(x0 + 8 + fpw, (int)sc.width - 1);
    const int yr0 = max(y0 - fph, 0), yr1 = min(y0 + 8 + fph, (int)sc.height - 1);
    const float img_x = sx - 0.5f, img_y = sy - 0.5f;
    const int bx = (int)floorf(img_x), by = (int)floorf(img_y);
    const int ix_lo = max(xr0, bx - fpw), ix_hi = min(xr1, bx + fpw + 1);
    const int iy_lo = max(yr0, by - fph), iy_hi = min(yr1, by + fph + 1);
    const int wx0 = x0 - fpw, wy0 = y0 - fph;
    for (int iy = iy_lo; iy <= iy_hi; ++iy) {
        float fy = fabsf((float)iy - img_y) * sc.inv_h;
        if (fy > sc.filter_h) continue;
        int fy_idx = min((int)(fy * (float)TRAY_FILTER_TABLE_SIZE), TRAY_FILTER_TABLE_SIZE - 1);
        for (int ix = ix_lo; ix <= ix_hi; ++ix) {
            float fx = fabsf((float)ix - img_x) * sc.inv_w;
            if (fx > sc.filter_w) continue;
            int fx_idx = min((int)(fx * (float)TRAY_FILTER_TABLE_SIZE), TRAY_FILTER_TABLE_SIZE - 1);
            float weight = s_table[fy_idx * TRAY_FILTER_TABLE_SIZE + fx_idx];
            int o = (iy - wy0) * WIN_STRIDE + (ix - wx0);
            atomicAdd(&s_win[o], weight * c.x);
            atomicAdd(&s_win[o + WIN_PLANE], weight * c.y);
            atomicAdd(&s_win[o + 2 * WIN_PLANE], weight * c.z);
            atomicAdd(&s_win[o + 3 * WIN_PLANE], weight);
        }
    }
}

// Row-binned film (filter_h == 2, separable table). With inv_h = 1/2 the y table index of pixel row
// iy, min(int(|iy - img_y| * 8), 15), only depends on which eighth of its pixel the sample lies in
// (class cy = floor((img_y - py) * 8)), so the y half of the filter can be applied once per
// (pixel row, class) instead of once per sample: a sample adds table_x[fx] * rgb to 8-9 columns of
// rowbin[py][cy] (36 LDS atomics instead of 324), and at the end of the tile every bin is spread
// over its 8-9 rows with table_y. Samples exactly on a class boundary (img_y * 8 integral, about
// 1 in 1000) take the direct path, which is RenderTarget::write verbatim. Products are grouped as
// ty * sum(tx * c) instead of sum((tx * ty) * c): same real number, last-bit rounding differs.
#define ROW_W WIN_MAX
#define ROWBIN_SIZE (8 * 8 * ROW_W * 4)
__device__ __forceinline__ void film_splat_global(const DevScene& sc, float* __restrict__ rgbw, const float* __restrict__ s_table,
                                                  int x0, int y0, float sx, float sy, f3 c);
__device__ __forceinline__ void film_splat_rows(const DevScene& sc, float* __restrict__ s_rowbin, const float* __restrict__ s_tx,
                                                float* __restrict__ rgbw, const float* __restrict__ s_table,
                                                int x0, int y0, int py_l, float sx, float sy, f3 c) {
    const float img_x = sx - 0.5f, img_y = sy - 0.5f;
    const float e8 = (img_y - (float)(y0 + py_l)) * 8.0f;   // exact in f32
    const float fl8 = floorf(e8);
    // (the ~1 in 1000 samples on a class boundary go straight to the caller's film: the LDS window only exists while a tile is resolved)
    if (e8 == fl8 || fl8 < -4.0f || fl8 > 3.0f) { film_splat_global(sc, rgbw, s_table, x0, y0, sx, sy, c); return; }
    const int cy = (int)fl8 + 4;
    const int fpw = sc.fpw;
    const int xr0 = max(x0 - fpw, 0), xr1 = min(x0 + 8 + fpw, (int)sc.width - 1);
    const int bx = (int)floorf(img_x);
    const int ix_lo = max(xr0, bx - fpw), ix_hi = min(xr1, bx + fpw + 1);
    const int wx0 = x0 - fpw;
    float* __restrict__ row = s_rowbin + (py_l * 8 + cy) * ROW_W * 4;
    for (int ix = ix_lo; ix <= ix_hi; ++ix) {
        float fx = fabsf((float)ix - img_x) * sc.inv_w;
        if (fx > sc.filter_w) continue;
        int fx_idx = min((int)(fx * (float)TRAY_FILTER_TABLE_SIZE), TRAY_FILTER_TABLE_SIZE - 1);
        float wx = s_tx[fx_idx];
        float* __restrict__ o = row + (ix - wx0) * 4;
        atomicAdd(o + 0, wx * c.x);
        atomicAdd(o + 1, wx * c.y);
        atomicAdd(o + 2, wx * c.z);
        atomicAdd(o + 3, wx);
    }
}
// Spread the row bins of a finished tile over the LDS window (all threads of the workgroup).
__device__ __forceinline__ void film_resolve_rows(const DevScene& sc, const float* __restrict__ s_rowbin, const float* __restrict__ s_ty,
                                                  float* __restrict__ s_win, int y0, uint32_t tid) {
    const int fph = sc.fph;
    const int yr0 = max(y0 - fph, 0), yr1 = min(y0 + 8 + fph, (int)sc.height - 1);
    const int wy0 = y0 - fph;
    for (int i = (int)tid; i < 8 * 8 * ROW_W; i += TR_BLOCK) {
        const float* __restrict__ b = s_rowbin + i * 4;
        const float r = b[0], g = b[1], bl = b[2], w = b[3];
        if (r == 0.0f && g == 0.0f && bl == 0.0f && w == 0.0f) continue;
        const int col = i % ROW_W, cyi = (i / ROW_W) & 7, py_l = i / (ROW_W * 8);
        const int c = cyi - 4;   // samples of this bin have (img_y - py) * 8 strictly inside (c, c + 1)
        for (int ky = -4; ky <= 4; ++ky) {
            if ((ky == -4 && c >= 0) || (ky == 4 && c < 0)) continue;   // |iy - img_y| <= 4 px
            const int iy = y0 + py_l + ky;
            if (iy < yr0 || iy > yr1) continue;
            const int d8 = 8 * ky - c;                       // floor(|iy - img_y| * 8)
            const int fy_idx = min(d8 > 0 ? d8 - 1 : -d8, TRAY_FILTER_TABLE_SIZE - 1);
            const float wy = s_ty[fy_idx];
            const int o = (iy - wy0) * WIN_STRIDE + col;
            atomicAdd(&s_win[o], wy * r);
            atomicAdd(&s_win[o + WIN_PLANE], wy * g);
            atomicAdd(&s_win[o + 2 * WIN_PLANE], wy * bl);
            atomicAdd(&s_win[o + 3 * WIN_PLANE], wy * w);
        }
    }
}

// ---- film helpers of the wavefront kernels: bins live in global memory, private to one workgroup
__device__ __forceinline__ void wg_add(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // stays in this XCD's L2
}
// RenderTarget::write of one sample straight into the caller's film (class-boundary samples and
// non-separable filters): render_target.rs:118-146 verbatim, with device-scope atomics
__device__ __forceinline__ void film_splat_global(const DevScene& sc, float* __restrict__ rgbw, const float* __restrict__ s_table,
                                                  int x0, int y0, float sx, float sy, f3 c) {
    const int fpw = sc.fpw, fph = sc.fph;
    const int xr0 = max(x0 - fpw, 0), xr1 = min(x0 + 8 + fpw, (int)sc.width - 1);
    const int yr0 = max(y0 - fph, 0), yr1 = min(y0 + 8 + fph, (int)sc.height - 1);
    const float img_x = sx - 0.5f, img_y = sy - 0.5f;
    const int bx = (int)floorf(img_x), by = (int)floorf(img_y);
    const int ix_lo = max(xr0, bx - fpw), ix_hi = min(xr1, bx + fpw + 1);
    const int iy_lo = max(yr0, by - fph), iy_hi = min(yr1, by + fph + 1);
    for (int iy = iy_lo; iy <= iy_hi; ++iy) {
        float fy = fabsf((float)iy - img_y) * sc.inv_h;
        if (fy > sc.filter_h) continue;
        int fy_idx = min((int)(fy * (float)TRAY_FILTER_TABLE_SIZE), TRAY_FILTER_TABLE_SIZE - 1);
        for (int ix = ix_lo; ix <= ix_hi; ++ix) {
            float fx = fabsf((float)ix - img_x) * sc.inv_w;
            if (fx > sc.filter_w) continue;
            int fx_idx = min((int)(fx * (float)TRAY_FILTER_TABLE_SIZE), TRAY_FILTER_TABLE_SIZE - 1);
            float weight = s_table[fy_idx * TRAY_FILTER_TABLE_SIZE + fx_idx];
            float* dst = rgbw + ((size_t)iy * sc.width + ix) * 4;
            atomicAdd(dst + 0, weight * c.x);
            atomicAdd(dst + 1, weight * c.y);
            atomicAdd(dst + 2, weight * c.z);
            atomicAdd(dst + 3, weight);
        }
    }
}
// bins: the chunk's row bins in global memory (LDS_BINS = false, workgroup-scope atomics) or this round's partial sums in
// LDS (true: ds_add_f32; the caller adds them to the global bins once per round)
template <bool LDS_BINS>
__device__ __forceinline__ void film_splat_rows_global(const DevScene& sc, float* __restrict__ bins, const float* __restrict__ s_tx,
                                                       float* __restrict__ rgbw, const float* __restrict__ s_table,
                                                       int x0, int y0, int py_l, float sx, float sy, f3 c) {
    const float img_x = sx - 0.5f, img_y = sy - 0.5f;
    const float e8 = (img_y - (float)(y0 + py_l)) * 8.0f;   // exact in f32
    const float fl8 = floorf(e8);
    if (e8 == fl8 || fl8 < -4.0f || fl8 > 3.0f) { film_splat_global(sc, rgbw, s_table, x0, y0, sx, sy, c); return; }
    const int cy = (int)fl8 + 4;
    const int fpw = sc.fpw;
    const int xr0 = max(x0 - fpw, 0), xr1 = min(x0 + 8 + fpw, (int)sc.width - 1);
    const int bx = (int)floorf(img_x);
    const int ix_lo = max(xr0, bx - fpw), ix_hi = min(xr1, bx + fpw + 1);
    const int wx0 = x0 - fpw;
    float* __restrict__ row = bins + (py_l * 8 + cy) * ROW_W * 4;
    for (int ix = ix_lo; ix <= ix_hi; ++ix) {
        float fx = fabsf((float)ix - img_x) * sc.inv_w;
        if (fx > sc.filter_w) continue;
        int fx_idx = min((int)(fx * (float)TRAY_FILTER_TABLE_SIZE), TRAY_FILTER_TABLE_SIZE - 1);
        float wx = s_tx[fx_idx];
        float* __restrict__ o = row + (ix - wx0) * 4;
        if (LDS_BINS) {
            atomicAdd(o + 0, wx * c.x); atomicAdd(o + 1, wx * c.y); atomicAdd(o + 2, wx * c.z); atomicAdd(o + 3, wx);
        } else {
            wg_add(o + 0, wx * c.x); wg_add(o + 1, wx * c.y); wg_add(o + 2, wx * c.z); wg_add(o + 3, wx);
        }
    }
}

#include "wavefront.h"

TR_DEV const float* sc_filter_table(const DevScene& sc) { return sc.filter_table; }

// Work item w (0 <= w < n_work) maps to queue entry (w / chunk) * chunk_stride * chunk + (w % chunk):
// contiguous ranges use chunk_stride = 1; multi-GPU sharding interleaves chunks round-robin.
// INTEG: the scene's integrator. TRAY_INTEGRATOR_WHITTED runs every camera sample of a wave to its end between two
// regenerations (dev_whitted.h) inside the same tile / film skeleton; NormalsDebug is a branch of vertex_begin.
// LFILT: compile mis_ray_filter in (scenes with a sphere light or specular lobes: the only ones it can act on; its mere presence costs
// the others 2 %: cornell_box 755 -> 740 Msamples/s at 64 spp).
template <int ANIM, int FEAT, int INTEG = TRAY_INTEGRATOR_PATH, bool LFILT = false>
__global__ __launch_bounds__(TR_BLOCK, (FEAT == 15 || INTEG != TRAY_INTEGRATOR_PATH) ? TR_MIN_WAVES_SIDE : ANIM ? TR_MIN_WAVES_ANIM : TR_MIN_WAVES) void k_path_tiles(const DevScene scv, const uint2* __restrict__ tiles, uint32_t tile_count,
                                                         uint32_t chunk, uint32_t chunk_stride, uint32_t spp, uint32_t kf, uint32_t levels,
                                                         float* __restrict__ rgbw, uint32_t* __restrict__ counter,
                                                         DevStats* __restrict__ stats) {
    const float* __restrict__ const s_table = sc_filter_table(scv);   // the 16 x 16 table stays in global memory (1 KB, cache resident): with the row-binned film only
                                                                       // the ~1 in 1000 samples on a class boundary read it, and the kilobyte decides whether a
                                                                       // third workgroup fits the CU's LDS on mesh scenes (42 granules of 1280 B per workgroup)
    __shared__ float s_rowbin[ROWBIN_SIZE];
    __shared__ float s_tx[TRAY_FILTER_TABLE_SIZE], s_ty[TRAY_FILTER_TABLE_SIZE];
    __shared__ uint4 s_perm[TR_PERM_BYTES / 16];   // the scene's permutation pool (dev_math.h): one byte read per LD array and vertex
    TR_DYN_LDS(uint32_t, s_stack);   // stack_depth x TR_BLOCK entries, sized per scene at launch
    __shared__ uint32_t s_tile, s_next_sample;
    // the 17 x 17 RGBW window a finished tile is resolved through. With the row-binned film nothing touches it while paths are traced, so
    // it lies over the traversal stacks (win_offset 0) and the 6.4 KB are not part of the workgroup's LDS footprint -- which is what lets
    // a third workgroup onto the CU when a mesh needs 29 stack entries; filters the row bins do not cover keep their own region
    float* const s_win = reinterpret_cast<float*>(s_stack + scv.win_offset);
    const DevScene& sc = scv;
    const DevScene* const scp = &scv;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t* const my_stack = s_stack + tid;
    if (tid < TRAY_FILTER_TABLE_SIZE) { s_tx[tid] = sc.filter_x[tid]; s_ty[tid] = sc.filter_y[tid]; }
    s_perm[tid] = reinterpret_cast<const uint4*>(sc.perm_pool)[tid];   // TR_PERM_BYTES / 16 == TR_BLOCK
    static_assert(TR_PERM_BYTES / 16 == TR_BLOCK, "one uint4 of the permutation pool per thread");
    const bool film_rows = sc.film_rows != 0u;
    Counters cnt;
    cnt.rays = 0; cnt.vertices = 0;
                          // VGPRs that live across the whole kernel
    uint32_t w_samples = 0u, w_vertices = 0u, w_rays = 0u;
#ifdef TR_STAGE_CLOCKS
    unsigned long long clk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // 0..6 stages, 8..10 parts of the BSDF queries (dev_integrator.h: TR_QCLK)
    long long clk_t = clock64();
#endif
    for (;;) {
        __syncthreads();   // previous tile fully flushed
        if (tid == 0) { s_tile = atomicAdd(counter, 1u); s_next_sample = 0u; }
        if (film_rows) { for (uint32_t i = tid; i < ROWBIN_SIZE; i += TR_BLOCK) s_rowbin[i] = 0.0f; }
        else for (uint32_t i = tid; i < 4 * WIN_PLANE; i += TR_BLOCK) s_win[i] = 0.0f;
        __syncthreads();
        // A work item is one SLICE of a tile's samples. The film is a sum and every sample is keyed by pixel and index, so the slices of a tile are
        // independent. Round 6: the slices are PROGRESSIVE and the queue is level-major -- with `levels` = L items per tile, level 0 of every tile
        // (the first half of its samples) comes first, then level 1 of every tile (the next quarter), ..., the last two levels are spp / 2^(L-1)
        // samples each: the launch starts with large items (one film resolve per item) and ENDS with small ones, so the last round of the persistent
        // workgroups is 1 / 2^(L-1) of a tile instead of a whole one. That is what a GPU's share of a frame needs when tiles differ in cost: one
        // eighth of the dragon frame ran at 0.70 of the whole frame's rate with two equal slices per tile, because a workgroup that drew a
        // mesh tile last kept the launch alive for half a heavy tile (profiles/r06_eighth_rate_all_shards.txt). L = 1: whole tiles.
        const uint32_t item = s_tile;
        if (item >= tile_count * levels) break;
        const uint32_t level = item / tile_count, ti = item - level * tile_count, last = levels - 1u;
        const uint32_t s_per_slice = level < last ? spp >> (level + 1u) : spp >> last, s_lo = level < last ? spp - (spp >> level) : spp - (spp >> last);
        const uint2 tile = tiles[(ti / chunk) * chunk_stride * chunk + (ti % chunk)];
        const int x0 = (int)tile.x * 8, y0 = (int)tile.y * 8;
        // The (pixel, sample) pairs of the slice are handed out dynamically: a lane whose path ended takes the next pair of the
        // workgroup's LDS counter -- pixel = pair % 64 in Region order (x fastest, ld.rs:47-51), sample = s_lo + pair / 64 -- whatever
        // pixel that is. With a fixed pixel per lane the lanes of a wave finish their 256 samples up to ~90 path vertices apart (the
        // sum of 256 path lengths has that spread) and idle until the last one is done; now the tile ends within one path length.
        // Every sample is keyed by its pixel and index (TRAY-CBRNG), so who computes it does not matter.
        const uint32_t n_pairs = 64u * s_per_slice;
        bool pairs_left = true;   // wave-uniform
        uint32_t row_l = 0u;       // pixel row (0..7) of the lane's current sample: row bin of the film
        bool pending = false;
        float sx = 0.0f, sy = 0.0f;
#ifdef TR_SAMPLE_DUMP
        size_t dump_idx = 0; uint32_t dump_v = 0u;
#endif
        Lane ln;
        ln.flags = 0u;
        ln.illum = mk(0.0f, 0.0f, 0.0f);
        ln.perm_lds = TR_LDS_B(s_perm);
#ifdef TR_STAGE_CLOCKS
        ln.qclk = clk + 8;
#endif
        for (;;) {   // one path vertex per live lane and step
            bool started = false;
            uint32_t kidx = 0u;
            const bool idle = !(ln.flags & LF_ALIVE);
            if (idle && pending) {   // the previous sample of this lane is finished: RenderTarget::write it
#ifdef TR_SAMPLE_DUMP   // instrumented builds only (tools/tile_sample_dump.py): the tile kernel's OWN per-sample radiance, unclamped, + the number of vertices shaded
                if (sc.sample_dump) {
                    sc.sample_dump[2 * dump_idx] = make_float4(ln.illum.x, ln.illum.y, ln.illum.z, (float)dump_v);
                    sc.sample_dump[2 * dump_idx + 1] = make_float4(ln.throughput.x, ln.throughput.y, ln.throughput.z, (float)ln.bounce);   // (where the path stopped)
                }
#endif
                if (film_rows) film_splat_rows(sc, s_rowbin, s_tx, rgbw, s_table, x0, y0, (int)row_l, sx, sy, lane_result(ln));
                else film_splat(sc, s_win, s_table, x0, y0, sx, sy, lane_result(ln));
                pending = false;
            }
            const unsigned long long idle_m = __ballot(idle);
            if (pairs_left && idle_m != 0ull) {   // ... and start the next one: one LDS atomic per wave reserves a pair for every idle lane
                const uint32_t n_idle = (uint32_t)__popcll(idle_m), leader = (uint32_t)__ffsll((long long)idle_m) - 1u;
                uint32_t base = 0u;
                if (lane == leader) base = atomicAdd(&s_next_sample, n_idle);
                base = __shfl(base, (int)leader);
                pairs_left = base + n_idle < n_pairs;
                const uint32_t pair = base + (uint32_t)__popcll(idle_m & ((1ull << lane) - 1ull));
                if (idle && pair < n_pairs) {
#ifdef TR_PAIR_BY_PIXEL   // experiment (round 5): consecutive pairs are the samples of ONE pixel, so a wave starts 64 samples of the same pixel
                    const uint32_t pix = pair / s_per_slice, s = s_lo + (pair % s_per_slice);
#else
                    const uint32_t pix = pair & 63u, s = s_lo + (pair >> 6);
#endif
                    const uint32_t px = (uint32_t)x0 + (pix & 7u), py = (uint32_t)y0 + (pix >> 3);
                    const uint32_t kp = key_pixel(kf, py * sc.width + px);
                    float t;
                    pixel_sample(kp, s, spp, px, py, sx, sy, t);
                    lane_start_sample(ln, camera_ray<ANIM>(sc, sx, sy, t), key_sample(kp, s));
                    if (ANIM) { ln.col = xf_cache_lane(); kidx = xf_time_index(t); }   // the path's column of the transform cache; its index into the frame's table (the fill copies from there)
                    row_l = pix >> 3;
                    started = true;
                    pending = true;
#ifdef TR_SAMPLE_DUMP
                    dump_idx = (size_t)(py * sc.width + px) * spp + s; dump_v = 0u;
#endif
                }
            }
            if (ANIM && idle_m != 0ull) xf_cache_fill_wave(sc, started, ln.time, ln.col, kidx);   // the paths' transforms of the moving instances, once per camera sample: the whole wave evaluates for the lanes that start one
            w_samples += (uint32_t)__popcll(__ballot(started));
            if (!__any(ln.flags & LF_ALIVE)) break;
            if (INTEG == TRAY_INTEGRATOR_WHITTED) {
                Ray cam;
                cam.o = LN_O(ln); cam.d = ln.d; cam.min_t = 0.0f; cam.max_t = TR_INF; cam.time = ln.time; cam.col = ln.col;
                ln.illum = whitted_run<ANIM>(sc, scp, my_stack, cam, ln.ks, (ln.flags & LF_ALIVE) != 0u, cnt, w_vertices, w_rays);
                ln.flags &= ~LF_ALIVE;
                continue;
            }
#ifdef TR_STAGE_CLOCKS
#define TR_CLK(slot) do { const long long now_ = clock64(); clk[slot] += (unsigned long long)(now_ - clk_t); clk_t = now_; } while (0)
            TR_CLK(6);   // regeneration + film splat
#else
#define TR_CLK(slot) ((void)0)
#endif
#pragma nounroll
            for (int stage = 0; stage < 3; ++stage) {
                const bool alive = (ln.flags & LF_ALIVE) != 0u;
                if (LFILT && stage == 2 && alive) mis_ray_filter<ANIM>(sc, ln);   // BSDF-sampled light rays that cannot hit the light are counted, not traced
                const bool want_ray = alive && (stage == 0 || (stage == 1 && (ln.flags & LF_SHADOW)) || (stage == 2 && (ln.flags & LF_MIS)));
                TraceResult tr_;
                tr_.hit = false;
                tr_.rec.t = 0.0f; tr_.rec.inst = 0xffffffffu; tr_.rec.prim = 0u; tr_.rec.b1 = 0.0f; tr_.rec.b2 = 0.0f;
                const unsigned long long wr_ = __ballot(want_ray);
                w_rays += (uint32_t)__popcll(wr_);
                if (LFILT && stage == 2) w_rays += (uint32_t)__popcll(__ballot(alive && (ln.flags & LF_MIS_MISS) != 0u));   // rays proven to miss the light (query_stage): counted like the reference's
                if (wr_ != 0ull) {
                    const Ray r = stage == 0 ? stage_a_ray(ln) : (stage == 1 ? stage_b_ray(ln) : stage_c_ray(ln));
                    tr_ = trace<ANIM, LFILT>(scp, my_stack, r, stage == 1, want_ray);
                }
                TR_CLK(stage);   // trace A / B / C
                if (stage == 0) w_vertices += (uint32_t)__popcll(__ballot(alive && tr_.hit));
#ifdef TR_SAMPLE_DUMP
                if (stage == 0 && alive && tr_.hit) ++dump_v;
#endif
                if (stage == 1) {
                    vertex_queries<ANIM, FEAT>(sc, ln, tr_.hit, alive);   // (the whole wave enters: wave-aligned query passes)
                } else if (alive) {
                    if (stage == 0) {
                        if (tr_.hit) vertex_begin<ANIM>(sc, ln, tr_.rec, cnt);
                        else ln.flags &= ~LF_ALIVE;   // camera miss: black sample; continuation miss: path ends (path.rs:112-115)
                    } else {
                        if (!vertex_end<ANIM>(sc, ln, tr_.hit, tr_.rec)) ln.flags &= ~LF_ALIVE;
                    }
                }
                TR_CLK(3 + stage);   // vertex_begin / queries / vertex_end
            }
        }
        __syncthreads();
        if (film_rows) {   // (every wave is out of the step loop: the stacks' memory is the window now)
            for (uint32_t i = tid; i < 4 * WIN_PLANE; i += TR_BLOCK) s_win[i] = 0.0f;
            __syncthreads();
            film_resolve_rows(sc, s_rowbin, s_ty, s_win, y0, tid);
            __syncthreads();
        }
        // flush the window: film::Image::add_pixels semantics on the caller's RGBW buffer
        const int wx0 = x0 - sc.fpw, wy0 = y0 - sc.fph;
        const int ww = 8 + 2 * sc.fpw + 1, wh = 8 + 2 * sc.fph + 1;
        for (int i = (int)tid; i < ww * wh; i += TR_BLOCK) {
            int wy = i / ww, wx = i - wy * ww;
            int ix = wx0 + wx, iy = wy0 + wy;
            if (ix < 0 || iy < 0 || ix >= (int)sc.width || iy >= (int)sc.height) continue;
            int o = wy * WIN_STRIDE + wx;
            float a = s_win[o + 3 * WIN_PLANE];
            if (a == 0.0f && s_win[o] == 0.0f && s_win[o + WIN_PLANE] == 0.0f && s_win[o + 2 * WIN_PLANE] == 0.0f) continue;
            float* dst = rgbw + ((size_t)iy * sc.width + ix) * 4;
            atomicAdd(dst + 0, s_win[o]);
            atomicAdd(dst + 1, s_win[o + WIN_PLANE]);
            atomicAdd(dst + 2, s_win[o + 2 * WIN_PLANE]);
            atomicAdd(dst + 3, a);
        }
    }
    if (stats) {
        if (lane == 0u) {
            atomicAdd(&stats->samples, (unsigned long long)w_samples);
            atomicAdd(&stats->vertices, (unsigned long long)w_vertices);
            atomicAdd(&stats->rays, (unsigned long long)w_rays);
        }
#ifdef TR_STAGE_CLOCKS
        if ((tid & 63u) == 0u)
            for (int k = 0; k < 11; ++k) atomicAdd(&stats->trav[k], clk[k]);
#endif
    }
}
#undef TR_CLK

// The frame's transforms by shutter-time index (dev_geom.h: xf_time_index): record (index * stride + m) holds AnimatedTransform::transform of
// moving instance m -- or, for m == n_moving, of the camera -- at the frame_time Camera::generate_ray computes for that index (camera.rs:152-153:
// the same expression as camera_ray's), evaluated by the same eval_xform_stack the per-path cache used to be filled with. blockIdx.y = m, so the
// lanes of a wave evaluate ONE stack at consecutive times; 2^24 x stride threads per frame (~15 ms for 11 instances).
#ifndef TR_DEVICE_TU   // (the non-template kernels are compiled once, with the host side: kernel_list.h)
__global__ __launch_bounds__(TR_BLOCK) void k_xf_table_build(const DevScene scv, float* __restrict__ table, uint32_t stride, uint32_t index0, uint32_t n_index) {
    const DevScene& sc = scv;
    const uint32_t i = blockIdx.x * TR_BLOCK + threadIdx.x, m = blockIdx.y;
    if (i >= n_index) return;
    const uint32_t index = index0 + i;
    const TrayCamera& c = *sc.camera_p;
    const float frame_time = (c.shutter_close - c.shutter_open) * xf_index_time(index) + c.shutter_open;
    uint32_t first, count;
    if (m < sc.n_moving) { const TrayInstance* __restrict__ in = sc.instances + sc.moving_ids[m]; first = in->xf_first; count = in->xf_count; }
    else { first = c.xf_first; count = c.xf_count; }
    float x[TR_XF_WORDS];
    eval_xform_stack(sc.xf_levels, sc.keyframes, sc.knots, first, count, frame_time, x);
    float4* __restrict__ rec = reinterpret_cast<float4*>(table + ((size_t)index * stride + m) * TR_XF_REC);
#pragma unroll
    for (int q = 0; q < TR_XF_WORDS / 4; ++q) rec[q] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}
// test hook: how many of n pseudo-random (index, record) pairs of the table differ in any bit from a fresh evaluation
__global__ __launch_bounds__(TR_BLOCK) void k_xf_table_check(const DevScene scv, const float* __restrict__ table, uint32_t stride, uint32_t n, uint32_t* __restrict__ n_bad) {
    const DevScene& sc = scv;
    const uint32_t i = blockIdx.x * TR_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t h = i * 0x9E3779B1u + 0x7feb352du; h ^= h >> 16; h *= 0x846ca68bu; h ^= h >> 15;
    const uint32_t index = h & 0xffffffu, m = (h >> 24) % stride;
    const TrayCamera& c = *sc.camera_p;
    const float frame_time = (c.shutter_close - c.shutter_open) * xf_index_time(index) + c.shutter_open;
    uint32_t first, count;
    if (m < sc.n_moving) { const TrayInstance* __restrict__ in = sc.instances + sc.moving_ids[m]; first = in->xf_first; count = in->xf_count; }
    else { first = c.xf_first; count = c.xf_count; }
    float x[TR_XF_WORDS];
    eval_xform_stack(sc.xf_levels, sc.keyframes, sc.knots, first, count, frame_time, x);
    const uint32_t* __restrict__ rec = reinterpret_cast<const uint32_t*>(table + ((size_t)index * stride + m) * TR_XF_REC);
    bool bad = false;
    for (int q = 0; q < TR_XF_WORDS; ++q) bad = bad || rec[q] != __float_as_uint(x[q]);
    if (bad) atomicAdd(n_bad, 1u);
}
#endif  // TR_DEVICE_TU

// ---- parity / debug kernels: the same device functions, one thread per item -------------------
template <int ANIM>
__global__ __launch_bounds__(TR_BLOCK) void k_debug_intersect(const DevScene scv, uint32_t n, const TrayRay* __restrict__ rays,
                                                              TrayHit* __restrict__ hits) {
    TR_DYN_LDS(uint32_t, s_stack);   // stack_depth x TR_BLOCK entries, sized per scene at launch
    const DevScene& sc = scv;
    const DevScene* const scp = &scv;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = idx < n;   // the whole wave stays for the traversal (cooperative leaf test)
    const uint32_t i = in_range ? idx : 0u;
    Ray r;
    r.o = mk(rays[i].o[0], rays[i].o[1], rays[i].o[2]);
    r.d = mk(rays[i].d[0], rays[i].d[1], rays[i].d[2]);
    r.min_t = rays[i].min_t; r.max_t = rays[i].max_t; r.time = rays[i].time; r.col = 0u;
    TrayHit o;
    memset(&o, 0, sizeof o);
    TraceResult tr_ = trace<ANIM>(scp, s_stack + threadIdx.x, r, false, in_range);
    if (!in_range) return;
    const HitRec rec = tr_.rec;
    if (tr_.hit) {
        float uv[2];
        f3 dp_dv;
        Hit h = finish_hit<ANIM>(sc, r, rec, uv, &dp_dv);
        o.t = rec.t; o.inst = rec.inst; o.prim = rec.prim;
        o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
        o.n[0] = h.n.x; o.n[1] = h.n.y; o.n[2] = h.n.z;
        o.ng[0] = h.ng.x; o.ng[1] = h.ng.y; o.ng[2] = h.ng.z;
        o.u = uv[0]; o.v = uv[1];
        o.dp_du[0] = h.dp_du.x; o.dp_du[1] = h.dp_du.y; o.dp_du[2] = h.dp_du.z;
        o.dp_dv[0] = dp_dv.x; o.dp_dv[1] = dp_dv.y; o.dp_dv[2] = dp_dv.z;
    } else {
        o.t = r.max_t; o.inst = 0xffffffffu;
    }
    hits[i] = o;
}

// thread_work's inner loop body for individual (pixel, sample) items (multithreaded.rs:94-103),
// driven through the same lane machine as the tile kernel
template <int ANIM>
__global__ __launch_bounds__(TR_BLOCK) void k_debug_sample_radiance(const DevScene scv, uint32_t n, const uint32_t* __restrict__ px,
                                                                    const uint32_t* __restrict__ py, const uint32_t* __restrict__ si,
                                                                    uint32_t spp, uint32_t kf, float* __restrict__ out) {
    TR_DYN_LDS(uint32_t, s_stack);   // stack_depth x TR_BLOCK entries, sized per scene at launch
    const DevScene& sc = scv;
    const DevScene* const scp = &scv;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = idx < n;   // the whole wave steps together (cooperative leaf test inside the traversal)
    const uint32_t i = in_range ? idx : 0u;
    Counters cnt;
    cnt.rays = 0; cnt.vertices = 0;
    const uint32_t kp = key_pixel(kf, py[i] * sc.width + px[i]);
    float sx, sy, t;
    pixel_sample(kp, si[i], spp, px[i], py[i], sx, sy, t);
    Lane ln;
    lane_start_sample(ln, camera_ray<ANIM>(sc, sx, sy, t), key_sample(kp, si[i]));
    if (!in_range) ln.flags = 0u;
    uint32_t* const my_stack = s_stack + threadIdx.x;
    if (sc.integrator == TRAY_INTEGRATOR_WHITTED) {   // the whole recursion of the wave's samples (dev_whitted.h)
        Ray cam;
        cam.o = LN_O(ln); cam.d = ln.d; cam.min_t = 0.0f; cam.max_t = TR_INF; cam.time = ln.time; cam.col = ln.col;
        uint32_t wv = 0u, wr = 0u;
        ln.illum = whitted_run<ANIM>(sc, scp, my_stack, cam, ln.ks, in_range, cnt, wv, wr);
        ln.flags = 0u;
    }
    while (__any(ln.flags & LF_ALIVE)) {
#pragma nounroll
        for (int stage = 0; stage < 3; ++stage) {
            const bool alive = (ln.flags & LF_ALIVE) != 0u;
            if (stage == 2 && alive) mis_ray_filter<ANIM>(sc, ln);
            const bool want_ray = alive && (stage == 0 || (stage == 1 && (ln.flags & LF_SHADOW)) || (stage == 2 && (ln.flags & LF_MIS)));
            TraceResult tr_;
            tr_.hit = false;
            tr_.rec.t = 0.0f; tr_.rec.inst = 0xffffffffu; tr_.rec.prim = 0u; tr_.rec.b1 = 0.0f; tr_.rec.b2 = 0.0f;
            if (stage == 2 && alive && (ln.flags & LF_MIS_MISS)) cnt.rays++;   // proven to miss the light: counted, not traced
            if (__any(want_ray)) {
                if (want_ray) cnt.rays++;
                const Ray r = stage == 0 ? stage_a_ray(ln) : (stage == 1 ? stage_b_ray(ln) : stage_c_ray(ln));
                tr_ = trace<ANIM>(scp, my_stack, r, stage == 1, want_ray);
            }
            if (stage == 1) {
                vertex_queries<ANIM, FEAT_ALL | FEAT_TEX>(sc, ln, tr_.hit, alive);
            } else if (alive) {
                if (stage == 0) {
                    if (tr_.hit) vertex_begin<ANIM>(sc, ln, tr_.rec, cnt);
                    else ln.flags &= ~LF_ALIVE;
                } else {
                    if (!vertex_end<ANIM>(sc, ln, tr_.hit, tr_.rec)) ln.flags &= ~LF_ALIVE;
                }
            }
        }
    }
    if (!in_range) return;
    f3 c = lane_result(ln);
    float* o = out + (size_t)i * 8;
    o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = sx; o[4] = sy; o[5] = (float)cnt.vertices; o[6] = (float)cnt.rays; o[7] = 0.0f;
}

// ---- the Samplers thread_work does not construct (sampler/uniform.rs, sampler/adaptive.rs; include/trayhip.h: tray_scene_set_sampler) ----
// One launch = one get_samples() round of every pixel of a batch of tiles (multithreaded.rs:91-103): a thread per camera sample of the
// round, driven through the same lane machine as the tile kernel, the sample written into the block's film window in LDS
// (RenderTarget::write, render_target.rs:118-146) that is added to the caller's film at the end. Not a hot path: one instantiation per ANIM (0 / 2), every lobe compiled in.
//   Uniform   one round: the pixel's centre, a uniform time, uniform numbers for every array of the integrator (uniform.rs:22-47).
//   Adaptive  round j generates `count` = min_spp (j = 0) or step_size positions -- sample_02(i + samples_taken) under the round's
//             scrambles, shuffled (adaptive.rs:92-116; Kensler's hashed permutation stands for rng.shuffle as in pixel_sample) -- and
//             max_spp time values of which thread_work's zip uses the first `count` (multithreaded.rs:76,93-94, adaptive.rs:117-121);
//             the integrator's arrays start at index samples_taken as well (lane_2d). A pixel k_sampler_decide has finished sits out.
struct SamplerPass {
    uint32_t kind, pass, count;
    uint32_t taken;         // Adaptive: samples_taken while the round's numbers are generated = min_spp + pass * step
    uint32_t before;        // samples of a pixel before this round
    uint32_t min_spp, max_spp, step;
    uint32_t lum_cap;       // luminance slots per pixel
};
// RenderTarget::write of one sample into an LDS window of ww x wh pixels (RGBW interleaved) whose pixel (0, 0) is image pixel (wx0, wy0): render_target.rs:118-146
// with the image's bounds as the only clip -- the window of a group of tiles covers every footprint of a sample inside them
TR_DEV void film_splat_window(const DevScene& sc, float* __restrict__ s_win, int wx0, int wy0, int ww, const float* __restrict__ table, float sx, float sy, f3 c) {
    const int fpw = sc.fpw, fph = sc.fph;
    const float img_x = sx - 0.5f, img_y = sy - 0.5f;
    const int bx = (int)floorf(img_x), by = (int)floorf(img_y);
    const int ix_lo = max(0, bx - fpw), ix_hi = min((int)sc.width - 1, bx + fpw + 1);
    const int iy_lo = max(0, by - fph), iy_hi = min((int)sc.height - 1, by + fph + 1);
    for (int iy = iy_lo; iy <= iy_hi; ++iy) {
        const float fy = fabsf((float)iy - img_y) * sc.inv_h;
        if (fy > sc.filter_h) continue;
        const int fy_idx = min((int)(fy * (float)TRAY_FILTER_TABLE_SIZE), TRAY_FILTER_TABLE_SIZE - 1);
        for (int ix = ix_lo; ix <= ix_hi; ++ix) {
            const float fx = fabsf((float)ix - img_x) * sc.inv_w;
            if (fx > sc.filter_w) continue;
            const int fx_idx = min((int)(fx * (float)TRAY_FILTER_TABLE_SIZE), TRAY_FILTER_TABLE_SIZE - 1);
            const float weight = table[fy_idx * TRAY_FILTER_TABLE_SIZE + fx_idx];
            float* __restrict__ o = s_win + ((iy - wy0) * ww + (ix - wx0)) * 4;
            atomicAdd(o + 0, weight * c.x); atomicAdd(o + 1, weight * c.y); atomicAdd(o + 2, weight * c.z); atomicAdd(o + 3, weight);
        }
    }
}
// thread_work under sampler::Uniform / sampler::Adaptive (and under LowDiscrepancy for scenes with an AnimatedMesh): ONE get_samples() round of every pixel
// of a batch of tiles (multithreaded.rs:84-112 with the round's samples of sampler/adaptive.rs:96-131, uniform.rs:22-48).
// Round 6: persistent form with path regeneration, as k_path_tiles. A workgroup owns a GROUP of up to SP_GROUP_MAX consecutive tiles of the queue -- 16
// tiles of the Z-order queue are a 32 x 32 pixel square when the group is aligned, which it is for whole frames and for the 16-tile chunks of a shard --
// and the round's (pixel, sample) pairs of the group are handed to whichever lane is idle (an LDS counter), so a wave stays full while the paths of
// its lanes end after different numbers of vertices; the group's film is ONE window in LDS (41 x 41 pixels at most with the 4-pixel filter halo), flushed
// once. Rounds 4-5 ran one thread per sample without regeneration (lanes at a third; Uniform's 64 samples per tile went straight to the caller's film with
// ~100 global atomics each): Uniform 159, Adaptive(4, 32) 260 - 310 Msamples/s at 1080p against the tile kernel's 1053 (profiles/r05_i_side_paths.txt).
// A group whose tiles do not form a square of at most 32 x 32 pixels (a ragged tile range) is walked tile by tile through the same window.
// FEAT: the lobe set compiled in, as for the tile kernel -- none of the optional ones (FEAT_NONE: matte / plastic / metal scenes, the common
// case) or all of them with textures (which also carries the Whitted integrator).
#define SP_GROUP_MAX 16
#define SP_WIN_MAX 41   // 4 tiles x 8 pixels + 2 x 4 pixels of halo + 1
template <int ANIM, int FEAT = FEAT_ALL | FEAT_TEX>
__global__ __launch_bounds__(TR_BLOCK, TR_MIN_WAVES_SIDE) void k_sampler_pass(const DevScene scv, const uint2* __restrict__ tiles, uint32_t item0, uint32_t n_items,
                                                           uint32_t chunk, uint32_t chunk_stride, uint32_t kf, SamplerPass sp,
                                                           const uint32_t* __restrict__ px_state, float* __restrict__ px_lum,
                                                           float* __restrict__ rgbw, DevStats* __restrict__ stats, uint32_t group) {
    TR_DYN_LDS(uint32_t, s_stack);
    __shared__ float s_win[4 * SP_WIN_MAX * SP_WIN_MAX];
    __shared__ uint2 s_tiles[SP_GROUP_MAX];
    __shared__ uint32_t s_next;
    __shared__ int s_box[4];
    __shared__ uint4 s_perm[TR_PERM_BYTES / 16];   // the scene's permutation pool (dev_math.h), as in k_path_tiles
    s_perm[threadIdx.x] = reinterpret_cast<const uint4*>(scv.perm_pool)[threadIdx.x];
    const DevScene& sc = scv;
    const DevScene* const scp = &scv;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t* const my_stack = s_stack + tid;
    const uint32_t g0 = blockIdx.x * group;                                   // first tile of the group, relative to item0
    const uint32_t n_g = g0 < n_items ? min(group, n_items - g0) : 0u;       // its tiles
    if (n_g == 0u) return;
    if (tid < n_g) { const uint32_t ti = item0 + g0 + tid; s_tiles[tid] = tiles[(ti / chunk) * chunk_stride * chunk + (ti % chunk)]; }
    __syncthreads();
    if (tid == 0u) {   // the group's box in tile coordinates (sixteen tiles at most)
        int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
        for (uint32_t k = 0; k < n_g; ++k) { bx0 = min(bx0, (int)s_tiles[k].x); by0 = min(by0, (int)s_tiles[k].y); bx1 = max(bx1, (int)s_tiles[k].x); by1 = max(by1, (int)s_tiles[k].y); }
        s_box[0] = bx0; s_box[1] = by0; s_box[2] = bx1; s_box[3] = by1;
    }
    __syncthreads();
    const bool square = s_box[2] - s_box[0] < 4 && s_box[3] - s_box[1] < 4;   // the whole group fits one window
    const uint32_t n_sub = square ? 1u : n_g, per_sub = square ? n_g : 1u;
    const uint32_t per_tile = 64u * sp.count;
    Counters cnt;
    cnt.rays = 0; cnt.vertices = 0;
    uint32_t n_samples = 0u;
    for (uint32_t sub = 0; sub < n_sub; ++sub) {
        // the window of this pass: the tiles [first, first + per_sub) of the group
        const uint32_t first = square ? 0u : sub;
        const int tx0 = square ? s_box[0] : (int)s_tiles[first].x, ty0 = square ? s_box[1] : (int)s_tiles[first].y;
        const int tx1 = square ? s_box[2] : tx0, ty1 = square ? s_box[3] : ty0;
        const int wx0 = tx0 * 8 - sc.fpw, wy0 = ty0 * 8 - sc.fph;
        const int ww = (tx1 - tx0 + 1) * 8 + 2 * sc.fpw + 1, wh = (ty1 - ty0 + 1) * 8 + 2 * sc.fph + 1;
        for (int k = (int)tid; k < 4 * ww * wh; k += TR_BLOCK) s_win[k] = 0.0f;
        if (tid == 0u) s_next = 0u;
        __syncthreads();
        const uint32_t n_pairs = per_sub * per_tile;
        bool pairs_left = true, pending = false;
        float sx = 0.0f, sy = 0.0f;
        uint32_t slot = 0u, i_smp = 0u;
        Lane ln;
        ln.flags = 0u;
        ln.illum = mk(0.0f, 0.0f, 0.0f);
        ln.perm_lds = TR_LDS_B(s_perm);
        for (;;) {   // one path vertex per live lane and step (k_path_tiles)
            const bool idle = !(ln.flags & LF_ALIVE);
            if (idle && pending) {   // the lane's previous sample is finished: RenderTarget::write it into the group's window, its luminance into the pixel's list
                const f3 c = lane_result(ln);
                film_splat_window(sc, s_win, wx0, wy0, ww, sc.filter_table, sx, sy, c);
                if (sp.kind == TRAY_SAMPLER_ADAPTIVE) px_lum[(size_t)slot * sp.lum_cap + sp.before + i_smp] = 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z;   // Colorf::luminance (color.rs:43-45)
                pending = false;
            }
            const unsigned long long idle_m = __ballot(idle);
            if (pairs_left && idle_m != 0ull) {
                const uint32_t n_idle = (uint32_t)__popcll(idle_m), leader = (uint32_t)__ffsll((long long)idle_m) - 1u;
                uint32_t base = 0u;
                if (lane == leader) base = atomicAdd(&s_next, n_idle);
                base = __shfl(base, (int)leader);
                pairs_left = base + n_idle < n_pairs;
                const uint32_t pair = base + (uint32_t)__popcll(idle_m & ((1ull << lane) - 1ull));
                if (idle && pair < n_pairs) {
                    // pair -> (tile of the pass, pixel of the tile in Region order, sample of the round): sampler/mod.rs:82-98 walks x fastest
                    const uint32_t t_in = pair / per_tile, rem = pair - t_in * per_tile, pix = rem / sp.count, i = rem - pix * sp.count;
                    const uint2 tile = s_tiles[first + t_in];
                    slot = (g0 + first + t_in) * 64u + pix;   // the pixel's index in the batch (px_state / px_lum)
                    if (!(sp.kind == TRAY_SAMPLER_ADAPTIVE && (px_state[slot] & 1u))) {   // (a pixel k_sampler_decide has finished sits out: its pairs are dropped)
                        const uint32_t px = tile.x * 8u + (pix & 7u), py = tile.y * 8u + (pix >> 3);
                        const uint32_t kp = key_pixel(kf, py * sc.width + px);
                        float t;
                        uint32_t ks;
                        if (sp.kind == TRAY_SAMPLER_LOW_DISCREPANCY) {                          // the tile kernel's samples (scenes with an AnimatedMesh)
                            pixel_sample(kp, i, sp.count, px, py, sx, sy, t);
                            ks = key_sample(kp, i);
                        } else if (sp.kind == TRAY_SAMPLER_UNIFORM) {
                            sx = (float)px + 0.5f; sy = (float)py + 0.5f;                      // uniform.rs:28
                            t = (float)(draw(kp, PD_SCR_T) >> 8) / 16777216.0f;                // uniform.rs:42-46
                            ks = key_sample(kp, 0u);
                        } else {
                            const uint32_t kq = key_pass(kp, sp.pass);
                            const uint32_t n_xy = permute(i, sp.count, draw(kq, PD_PERM_XY)) + sp.taken;
                            sx = van_der_corput(n_xy, draw(kq, PD_SCR_X)) + (float)px;         // adaptive.rs:106-110
                            sy = sobol(n_xy, draw(kq, PD_SCR_Y)) + (float)py;
                            t = van_der_corput(permute(i, sp.max_spp, draw(kq, PD_PERM_T)) + sp.taken, draw(kq, PD_SCR_T));
                            ks = key_sample(kq, i);
                        }
                        lane_start_sample(ln, camera_ray<ANIM>(sc, sx, sy, t), ks);
                        ln.smp_kind = sp.kind; ln.smp_offset = sp.taken;
                        i_smp = i;
                        pending = true;
                        ++n_samples;
                    }
                }
            }
            if (!__any(ln.flags & LF_ALIVE)) { if (pairs_left) continue; break; }   // (dropped pairs can leave a wave without a live lane while pairs remain)
            if (FEAT == (FEAT_ALL | FEAT_TEX) && sc.integrator == TRAY_INTEGRATOR_WHITTED) {   // every camera sample of the wave to its end (dev_whitted.h)
                Ray cam;
                cam.o = LN_O(ln); cam.d = ln.d; cam.min_t = 0.0f; cam.max_t = TR_INF; cam.time = ln.time; cam.col = ln.col;
                uint32_t wv = 0u, wr = 0u;
                ln.illum = whitted_run<ANIM>(sc, scp, my_stack, cam, ln.ks, (ln.flags & LF_ALIVE) != 0u, cnt, wv, wr, sp.kind, sp.taken);
                ln.flags &= ~LF_ALIVE;
                continue;
            }
#pragma nounroll
            for (int stage = 0; stage < 3; ++stage) {
                const bool alive = (ln.flags & LF_ALIVE) != 0u;
                if (stage == 2 && alive) mis_ray_filter<ANIM>(sc, ln);
                const bool want_ray = alive && (stage == 0 || (stage == 1 && (ln.flags & LF_SHADOW)) || (stage == 2 && (ln.flags & LF_MIS)));
                TraceResult tr_;
                tr_.hit = false;
                tr_.rec.t = 0.0f; tr_.rec.inst = 0xffffffffu; tr_.rec.prim = 0u; tr_.rec.b1 = 0.0f; tr_.rec.b2 = 0.0f;
                if (stage == 2 && alive && (ln.flags & LF_MIS_MISS)) cnt.rays++;
                if (__any(want_ray)) {
                    if (want_ray) cnt.rays++;
                    const Ray r = stage == 0 ? stage_a_ray(ln) : (stage == 1 ? stage_b_ray(ln) : stage_c_ray(ln));
                    tr_ = trace<ANIM>(scp, my_stack, r, stage == 1, want_ray);
                }
                if (stage == 1) {
                    vertex_queries<ANIM, FEAT>(sc, ln, tr_.hit, alive);
                } else if (alive) {
                    if (stage == 0) {
                        if (tr_.hit) vertex_begin<ANIM>(sc, ln, tr_.rec, cnt);
                        else ln.flags &= ~LF_ALIVE;
                    } else {
                        if (!vertex_end<ANIM>(sc, ln, tr_.hit, tr_.rec)) ln.flags &= ~LF_ALIVE;
                    }
                }
            }
        }
        __syncthreads();
        // flush the window: film::Image::add_pixels semantics on the caller's RGBW buffer (as k_path_tiles)
        for (int k = (int)tid; k < ww * wh; k += TR_BLOCK) {
            const int wy = k / ww, wx = k - wy * ww;
            const int ix = wx0 + wx, iy = wy0 + wy;
            if (ix < 0 || iy < 0 || ix >= (int)sc.width || iy >= (int)sc.height) continue;
            const float* __restrict__ o = s_win + 4 * k;
            if (o[3] == 0.0f && o[0] == 0.0f && o[1] == 0.0f && o[2] == 0.0f) continue;
            float* dst = rgbw + ((size_t)iy * sc.width + ix) * 4;
            atomicAdd(dst + 0, o[0]); atomicAdd(dst + 1, o[1]); atomicAdd(dst + 2, o[2]); atomicAdd(dst + 3, o[3]);
        }
        __syncthreads();
    }
    if (stats) {   // one update per wave
        for (int off = 32; off > 0; off >>= 1) { n_samples += __shfl_down(n_samples, off); cnt.vertices += __shfl_down(cnt.vertices, off); cnt.rays += __shfl_down(cnt.rays, off); }
        if (lane == 0u && n_samples) {
            atomicAdd(&stats->samples, (unsigned long long)n_samples);
            atomicAdd(&stats->vertices, (unsigned long long)cnt.vertices);
            atomicAdd(&stats->rays, (unsigned long long)cnt.rays);
        }
    }
}
// Adaptive::report_results / needs_supersampling (adaptive.rs:55-75, 133-143) of every unfinished pixel of the batch, after round
// `sp.pass`: one thread per pixel walks the luminances of ALL the pixel's samples so far in the order they were taken.
#ifndef TR_DEVICE_TU
__global__ __launch_bounds__(TR_BLOCK) void k_sampler_decide(uint32_t n_pixels, SamplerPass sp, uint32_t* __restrict__ px_state,
                                                             float* __restrict__ px_avg, const float* __restrict__ px_lum) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_pixels || (px_state[slot] & 1u)) return;
    const float* __restrict__ lum = px_lum + (size_t)slot * sp.lum_cap;
    const uint32_t n = sp.before + sp.count;
    bool more = false;
    if (sp.taken < sp.max_spp) {   // (report_results tests samples_taken >= max_spp first: the average is not touched then)
        float avg = px_avg[slot];
        if (sp.taken == sp.min_spp) {   // first round: the plain mean
            float ac = 0.0f;
            for (uint32_t k = 0; k < n; ++k) ac = ac + lum[k];
            avg = ac / (float)n;
        } else {                        // the new samples enter a running average -- with (i - 1) / i where a mean would have i / (i + 1)
            for (uint32_t k = n - sp.step; k < n; ++k) avg = (lum[k] + (float)(k - 1u) * avg) / (float)k;
        }
        px_avg[slot] = avg;
        for (uint32_t k = 0; k < n && !more; ++k) more = fabsf(lum[k] - avg) / avg > 0.5f;
    }
    if (!more) px_state[slot] |= 1u;
}

__global__ __launch_bounds__(64) void k_debug_bsdf(const DevScene scv, uint32_t flags_sel, uint32_t n,
                                                   const float* __restrict__ dirs, const float* __restrict__ u3, float* __restrict__ out) {
    const DevScene& sc = scv;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // canonical frame: n = +z, dp_du = +x; instance 0 of the (private) scene copy carries the material
    Hit h;
    h.p = mk(0.0f, 0.0f, 0.0f); h.n = mk(0.0f, 0.0f, 1.0f); h.ng = mk(0.0f, 0.0f, 1.0f); h.dp_du = mk(1.0f, 0.0f, 0.0f);
    h.inst = 0; h.u = 0.5f; h.v = 0.5f;
    Bsdf b = make_bsdf(sc, h);
    DevMaterial hit_mat;   // a textured material: its lobes at the centre of the texture, frame time 0
    if (b.mat->textured) { resolve_textured(sc, b.mat, h.u, h.v, 0.0f, hit_mat); b.mat = &hit_mat; }
    uint32_t flags = flags_sel == 0 ? BX_ALL : BX_NON_SPECULAR;
    f3 wo = mk(dirs[6 * i], dirs[6 * i + 1], dirs[6 * i + 2]), wi = mk(dirs[6 * i + 3], dirs[6 * i + 4], dirs[6 * i + 5]);
    float* o = out + (size_t)i * 12;
    f3 e = bsdf_eval<FEAT_ALL | FEAT_TEX>(b, wo, wi, flags);
    o[0] = e.x; o[1] = e.y; o[2] = e.z; o[3] = bsdf_pdf<FEAT_ALL | FEAT_TEX>(b, wo, wi, flags);
    f3 swi;
    float spdf;
    uint32_t st;
    f3 f = bsdf_sample(b, wo, flags, u3[3 * i], u3[3 * i + 1], u3[3 * i + 2], swi, spdf, st);
    o[4] = f.x; o[5] = f.y; o[6] = f.z; o[7] = swi.x; o[8] = swi.y; o[9] = swi.z; o[10] = spdf; o[11] = (float)st;
}

#endif  // TR_DEVICE_TU
