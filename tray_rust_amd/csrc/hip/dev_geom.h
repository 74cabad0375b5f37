// Device-side geometry: one-loop two-level BVH traversal (closest / any hit), primitive tests,
// deferred differential geometry, area-light sampling. One thread = one ray. Mirrors, per function:
//   Scene::intersect scene.rs:148-150        BVH::intersect bvh.rs:81-130     fast_intersect bbox.rs:75-104
//   Instance/Receiver/Emitter::intersect receiver.rs:29-44, emitter.rs:118-137
//   intersect_triangle mesh.rs:136-198       Sphere sphere.rs:33-140          Rectangle rectangle.rs:38-104
//   Disk disk.rs:42-110                      Light for Emitter emitter.rs:140-203
// Differences by design (results identical):
//  * the reference builds a DifferentialGeometry for every accepted candidate; here traversal only
//    keeps (t, instance, primitive, barycentrics) and the differential geometry of the final hit is
//    rebuilt once from exactly the same inputs;
//  * the reference nests BVH<Triangle>::intersect inside BVH<Instance>::intersect; here both levels
//    run in ONE loop over a shared stack (instances of a leaf and an "exit mesh" sentinel are stack
//    entries), so lanes that are in different levels still execute the same node-test code. The
//    visiting order of nodes, instances and triangles is the reference's.
#pragma once
#include "../../../include/trayhip.h"
#include "dev_math.h"
#include "dev_anim.h"
#include "../host/gates.hpp"

namespace tr {

// Lobes of a material, precomputed on the host at scene creation (dev_bsdf.h)
struct DevLobe {
    uint32_t kind, type;
    float color[3];
    float eta_t;   // Dielectric::new(1.0, eta_t)
    float width;   // Beckmann width | Oren-Nayar A
    float ob;      // Oren-Nayar B
};
struct DevMaterial {
    uint32_t n_lobes;
    uint32_t mat_kind;      // TRAY_MAT_*: the key of the wavefront schedule's material sort
    uint64_t merl_offset;   // float offset into merl_data
    DevLobe lobe[2];
    float eta[4];           // conductor eta (rgb)
    float k[4];             // conductor k (rgb)
    // image textures (dev_tex.h): a material with a textured parameter keeps its source parameters; its lobes are lowered per hit
    // from the values sampled at the hit's (u, v, time), exactly as Material::bsdf does it in the reference
    uint32_t textured;      // 1 => lobe[] / eta / k above are placeholders
    uint32_t tex_c0, tex_c1, tex_f0, tex_f1;
    float c0[3], c1[3], f0, f1;
    uint32_t microfacet;    // TRAY_MF_*
    uint32_t pad_tex[2];
};

struct DevScene {
    const TrayInstance* __restrict__ instances;
    const TrayBvhNode* __restrict__ top_nodes;
    const uint32_t* __restrict__ top_order;
    const TrayMesh* __restrict__ meshes;
    const TrayBvhNode* __restrict__ mesh_nodes;
    const TrayTriVerts* __restrict__ tri_verts;
    const TrayTriAttrs* __restrict__ tri_attrs;
    const DevMaterial* __restrict__ materials;
    const float* __restrict__ merl_data;
    const uint32_t* __restrict__ lights;
    const float* __restrict__ filter_table;
    const float* __restrict__ filter_x;   // separable factors: table[y*16+x] == filter_x[x] * filter_y[y]
    const float* __restrict__ filter_y;
    const TrayXformLevel* __restrict__ xf_levels;   // spline stacks of moving instances / camera (dev_anim.h)
    const TrayKeyframe* __restrict__ keyframes;
    const float* __restrict__ knots;
    const TrayColorKey* __restrict__ color_keys;
    // Per-path transform cache of the moving instances (ANIM kernels): every ray of a path carries the camera ray's time, so
    // each lane evaluates the spline stacks ONCE per camera sample and keeps mat/inv rows in HBM, laid out
    // [moving_slot][TR_XF_WORDS floats][lane] so that a wave reads / writes 256 contiguous bytes per float.
    float* __restrict__ xf_cache;              // nullptr: evaluate at every use (debug kernels)
    const uint32_t* __restrict__ moving_ids;   // instance ids of the moving instances, by moving_slot
    uint32_t n_moving, xf_cache_lanes;
    uint32_t xf_aos;                           // layout of xf_cache: 0 = [instance][word][column] (tile kernel: a wave's columns are consecutive, every
                                               // word is one coalesced load), 1 = [column][instance][24 words] (wavefront: the lanes of a wave hold
                                               // arbitrary pool slots, a transform is six 16-byte loads of one 96-byte record instead of 24 cache lines)
    uint32_t n_instances, n_lights, min_depth, max_depth;
    uint32_t width, height, frame, film_rows;   // film_rows: 1 = row-binned film (separable, filter_h == 2)
    uint32_t integrator;                        // TRAY_INTEGRATOR_*
    uint32_t xf_table;                          // 1: xf_cache IS the frame's table of transforms by shutter-time index (xf_tab below), a path's column is its time index
    uint32_t coop_offset;   // word offset of the cooperative leaf test's LDS area behind the traversal stacks (0 = none)
    const tray::FlatLeaf* __restrict__ flat_leaves;   // the flat instance loop's view of the scene: BVH<Instance> leaves ...
    const tray::FlatInst* __restrict__ flat_insts;    // ... and their instances, one 128-B record each (host/gates.hpp)
    uint32_t n_flat_leaves;
    uint32_t win_offset;                        // tile kernel: word offset of the 17 x 17 RGBW film window in the dynamic LDS (0: it shares the traversal stacks' memory,
                                                // which is free while a finished tile is resolved -- row-binned film only)
    const TrayTexture* __restrict__ textures;      // image textures (dev_tex.h); null when the scene has none
    const TrayTexFrame* __restrict__ tex_frames;
    const uint8_t* __restrict__ tex_data;
    uint32_t* __restrict__ retraced;               // counter of rays the flat loop handed to trace_bvh (tied candidates); may be null
    const uint8_t* __restrict__ tri_leaf;          // per triangle of a mesh with <= TR_COOP_MAX_TRIS triangles: index of its BVH<Triangle> leaf node (mesh_leaf_coop's gate)
    float filter_w, filter_h, inv_w, inv_h;
    int32_t fpw, fph;
                       // where it held scalar registers for the whole kernel (193 -> 140 SGPRs spilled to VGPR lanes in k_path_tiles)
    const TrayCamera* __restrict__ camera_p;
    const uint8_t* __restrict__ perm_pool;   // TR_PERM_BYTES: the shuffles the per-path LD arrays draw from (dev_math.h: perm_pool_build)
    const tray::WfInst* __restrict__ wf_insts;   // one 64-B record per BVH<Instance> leaf slot (host/gates.hpp): the wavefront traversal's instance entry
    const float4* __restrict__ quads;            // the wavefront traversal's trees (host/gates.hpp: QuadTrees): 128-byte records of up to four (box, descriptor)
                                                 // slots = two levels of the binary trees above. ONE buffer -- the BVH<Triangle>s first, then this frame's
                                                 // BVH<Instance> from record top_quad_first -- so that a record's address is a uniform base + a 32-bit offset
    uint32_t top_quad_first;
    uint32_t xf_stride;                          // records per column of xf_cache in the record layout (xf_aos): n_moving, + 1 in table mode when the camera moves (its record is the last)
    // AnimatedMesh (geometry/animated_mesh.rs; include/trayhip.h: TrayMeshKeys): per mesh its keyframe count and times, or null. Only the
    // ANIM = 3 instantiations (debug kernels, k_sampler_pass) read them.
    const TrayMeshKeys* __restrict__ mesh_keys;
    const float* __restrict__ key_times;
    // the frame's transform table by shutter-time index (below: xf_time_index), or null; xf_tab_stride records per index: the moving instances, then the
    // camera if it moves. The wavefront kernels index it directly (xf_table = 1: xf_cache == xf_tab), the tile kernel copies a path's records into its cache column
    const float* __restrict__ xf_tab;
    uint32_t xf_tab_stride, pad_tab;
#ifdef TR_SAMPLE_DUMP   // instrumented builds only: [pixel][sample] records the tile kernel writes when a sample is finished (kernels.hip)
    float4* __restrict__ sample_dump;
#endif
};

struct Ray {
    f3 o, d;
    float min_t, max_t;
    float time;   // ray.time (linalg/ray.rs:17): only read by the kernels built for moving scenes (ANIM)
    uint32_t col; // column of the per-path transform cache that belongs to this ray's path (ANIM)
};

// Transform of an instance at a ray's time as rows 0..2 of mat (x) and of inv (x + 12). Instance transforms are products
// of TRS keyframes (AnimatedTransform::unanimated decomposes static ones too), so row 3 is (0,0,0,1) and the affine
// point transform equals Transform * Point.
#ifdef TR_XF_COL_MASK   // experiment (round 6, profiles/r06_moving_box_fill_ab.txt): the threads share a few columns that stay in L2 -- WRONG pictures, the time cache-resident columns would give
TR_DEV uint32_t xf_cache_lane() { return (blockIdx.x * blockDim.x + threadIdx.x) & (uint32_t)(TR_XF_COL_MASK); }
#else
TR_DEV uint32_t xf_cache_lane() { return blockIdx.x * blockDim.x + threadIdx.x; }   // megakernel: one column per thread
#endif
// Round 5: the frame's transforms as a TABLE over the shutter-time index. A camera sample's time is one of 2^24 values -- van_der_corput
// returns (bits >> 8) / 2^24 (ld.rs:100-104; the clamp to 1 - f32::EPSILON lands on such a value too) -- and every ray of the path inherits it
// (path.rs:110, mod.rs:154), so AnimatedTransform::transform(ray.time) of an instance is a function of that 24-bit index alone. A 512-spp frame
// at 1920x1080 takes 1.06e9 camera samples, i.e. it evaluates every moving instance's spline stack 63 times per index (k_wf_regen: 17 % of the
// frame with 11 moving instances, VALU-bound; the tile kernel's moving scenes: a third of their extra instructions) and keeps the results in
// a per-path cache of 112 B per path and instance. The table holds each of them ONCE: xf_cache[(index * xf_stride + moving_slot) * TR_XF_WORDS],
// built by k_xf_table_build (kernels.hip) per frame with the same eval_xform_stack at the same frame_time -- the same bits by construction --,
// 1.9 GB per moving instance. The wavefront kernels index it directly: a path's `column` is its time index, nothing is evaluated or stored per
// path (C5 stand-in, frame 64: 186 -> 213 Msamples/s). The tile kernel keeps its per-thread cache columns (coalesced reads in the flat instance
// loop; gathered 112-byte records there measured only +3.5 % on moving_box) and FILLS them from the table instead of evaluating. The host builds
// the table for launches of enough samples (kernels.hip: xf_table_prepare), the per-path evaluation serves the others.
#ifndef TR_XF_FILL_COOP   // the tile kernel's fill of its cache columns from the frame's table: 1 = dealt out to the whole wave (xf_cache_fill_wave), 0 = by the starting lanes
#define TR_XF_FILL_COOP 1
#endif
#ifdef TR_XF_KIDX_MASK   // experiment (round 5, profiles/r05_xf_table_locality_ceiling.txt): every path reads one of a few table records -- WRONG pictures, the time a perfectly cache-resident table would give
TR_DEV uint32_t xf_time_index(float t) { return (uint32_t)(t * 16777216.0f) & (uint32_t)(TR_XF_KIDX_MASK); }
#else
TR_DEV uint32_t xf_time_index(float t) { return (uint32_t)(t * 16777216.0f); }   // t = index / 2^24 exactly
#endif
TR_DEV float xf_index_time(uint32_t index) { return (float)index * (1.0f / 16777216.0f); }
// start of a camera sample: evaluate every moving instance at the path's time into the path's cache column
TR_DEV void xf_cache_fill(const DevScene& sc, float time, uint32_t lane) {
    if (!sc.xf_cache || sc.xf_table) return;
    const uint32_t lanes = sc.xf_cache_lanes;
    for (uint32_t m = 0; m < sc.n_moving; ++m) {
        const TrayInstance* __restrict__ in = sc.instances + sc.moving_ids[m];
        float x[TR_XF_WORDS];
        eval_xform_stack(sc.xf_levels, sc.keyframes, sc.knots, in->xf_first, in->xf_count, time, x);
        if (sc.xf_aos) {
            float4* __restrict__ rec = reinterpret_cast<float4*>(sc.xf_cache + ((size_t)lane * sc.xf_stride + m) * TR_XF_REC);
#pragma unroll
            for (int q = 0; q < TR_XF_WORDS / 4; ++q) rec[q] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
        } else {
            float* __restrict__ col = sc.xf_cache + (size_t)m * TR_XF_WORDS * lanes + lane;
#pragma unroll
            for (int k = 0; k < 26; ++k) col[(size_t)k * lanes] = x[k];
        }
    }
}
// The same for the lanes of a wave that start a camera sample in the same step of the tile kernel (`started`; called by ALL lanes). A step
// regenerates a third of the wave's lanes or fewer, and the evaluation of a spline stack is long: run under the starting lanes' mask it
// occupied 44 % of the moving test scene's wave cycles (profiles/r04_moving_box_cooperative_fill_ab.txt). Here the (starting lane,
// moving instance) pairs are dealt out to all the lanes of the wave -- instance-major, so neighbouring lanes evaluate the SAME stack at
// different times and stay in step -- and every lane writes its result into the column of the lane it worked for. Values and layout are
// exactly xf_cache_fill's; the stores are made visible to the wave before anybody reads its column.
TR_DEV void xf_cache_fill_wave(const DevScene& sc, bool started, float time, uint32_t column, uint32_t time_index) {
    if (!sc.xf_cache || sc.xf_table) return;
#if TR_XF_FILL_COOP
    if (sc.xf_tab) {   // the frame's table has the path's transforms: copied, not evaluated. The (starting lane, moving instance) pairs are dealt out to all
        // the lanes of the wave, instance-major: every lane fetches ONE record from its random place in the table -- one trip to HBM for the wave
        // instead of one per moving instance for the starting lanes, and a third of the lanes start a sample in EVERY step -- and writes it into the
        // column of the lane it works for (neighbouring lanes work for neighbouring columns of the same instance: the stores of a word share
        // their lines). moving_box: 459 -> 524 Msamples/s at 32 spp, 475 -> 557 at 128 (profiles/r06_moving_box_fill_ab.txt).
        const unsigned long long start_m = __ballot(started), exec_m = __ballot(1);
        if (start_m == 0ull) return;
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t n_exec = (uint32_t)__popcll(exec_m), my_rank = (uint32_t)__popcll(exec_m & ((1ull << lane) - 1ull));
        const uint32_t n_start = (uint32_t)__popcll(start_m), n_tasks = n_start * sc.n_moving;
        const uint32_t lanes = sc.xf_cache_lanes;
        for (uint32_t base = 0; base < n_tasks; base += n_exec) {
            const uint32_t task = base + my_rank;
            const bool valid = task < n_tasks;
            const uint32_t m = valid ? task / n_start : 0u, which = valid ? task % n_start : 0u;
            uint32_t src = 0u, k = 0u;   // the which-th starting lane of the wave (a scalar walk over the mask's bits)
            for (unsigned long long rest = start_m; rest != 0ull; rest &= rest - 1ull, ++k) {
                const uint32_t b = (uint32_t)__ffsll((long long)rest) - 1u;
                if (k == which) src = b;
            }
            const uint32_t kidx = __shfl(time_index, (int)src);
            const uint32_t col = __shfl(column, (int)src);
            if (valid) {
                const float4* __restrict__ rec = reinterpret_cast<const float4*>(sc.xf_tab + ((size_t)kidx * sc.xf_tab_stride + m) * TR_XF_REC);
                float* __restrict__ dst = sc.xf_cache + (size_t)m * TR_XF_WORDS * lanes + col;
                const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5], r6 = rec[6];
                const float v[26] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w,
                                     r4.x, r4.y, r4.z, r4.w, r5.x, r5.y, r5.z, r5.w, r6.x, r6.y};
#pragma unroll
                for (int q = 0; q < 26; ++q) dst[(size_t)q * lanes] = v[q];
            }
        }
        __builtin_amdgcn_wave_barrier();   // (as below: the host emulation's lanes meet here)
        __threadfence_block();             // (a lane reads its column next: written by another lane of this wave)
        return;
    }
#endif
    if (sc.xf_tab) {   // the frame's table has the path's transforms: copied, not evaluated (the lanes that start a sample, each its own records)
        if (started) {
            const uint32_t lanes = sc.xf_cache_lanes;
            for (uint32_t m = 0; m < sc.n_moving; ++m) {
                const float4* __restrict__ rec = reinterpret_cast<const float4*>(sc.xf_tab + ((size_t)time_index * sc.xf_tab_stride + m) * TR_XF_REC);
                float* __restrict__ dst = sc.xf_cache + (size_t)m * TR_XF_WORDS * lanes + column;
#pragma unroll
                for (int q = 0; q < 6; ++q) { const float4 v = rec[q]; dst[(size_t)(4 * q) * lanes] = v.x; dst[(size_t)(4 * q + 1) * lanes] = v.y; dst[(size_t)(4 * q + 2) * lanes] = v.z; dst[(size_t)(4 * q + 3) * lanes] = v.w; }
                const float4 w = rec[6];
                dst[(size_t)24 * lanes] = w.x; dst[(size_t)25 * lanes] = w.y;
            }
        }
        return;
    }
    const unsigned long long start_m = __ballot(started), exec_m = __ballot(1);
    if (start_m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_exec = (uint32_t)__popcll(exec_m), my_rank = (uint32_t)__popcll(exec_m & ((1ull << lane) - 1ull));
    const uint32_t n_start = (uint32_t)__popcll(start_m), n_tasks = n_start * sc.n_moving;
    const uint32_t lanes = sc.xf_cache_lanes;
    for (uint32_t base = 0; base < n_tasks; base += n_exec) {
        const uint32_t task = base + my_rank;
        const bool valid = task < n_tasks;
        const uint32_t m = valid ? task / n_start : 0u, which = valid ? task % n_start : 0u;
        // the which-th starting lane of the wave (a scalar walk over the mask's bits)
        uint32_t src = 0u, k = 0u;
        for (unsigned long long rest = start_m; rest != 0ull; rest &= rest - 1ull, ++k) {
            const uint32_t b = (uint32_t)__ffsll((long long)rest) - 1u;
            if (k == which) src = b;
        }
        const float t = __shfl(time, (int)src);
        const uint32_t col = __shfl(column, (int)src);
        if (valid) {
            const TrayInstance* __restrict__ in = sc.instances + sc.moving_ids[m];
            float x[TR_XF_WORDS];
            eval_xform_stack(sc.xf_levels, sc.keyframes, sc.knots, in->xf_first, in->xf_count, t, x);
            float* __restrict__ dst = sc.xf_cache + (size_t)m * TR_XF_WORDS * lanes + col;
#pragma unroll
            for (int q = 0; q < 26; ++q) dst[(size_t)q * lanes] = x[q];
        }
    }
    __builtin_amdgcn_wave_barrier();   // (no instruction on the device, where a wave's lanes move together; the host emulation's lanes meet here)
    __threadfence_block();   // (a lane reads its column next: written by another lane of this wave)
}
// ANIM template values: 0 = nothing moves within the frame; 1 = moving instances are read from the per-path cache (tile and
// wavefront kernels: no function call in their hot loops, a call would raise their register allocation to the callee's);
// 2 = the spline stacks are evaluated at every use (debug kernels, whose grids are not sized by the cache)
// 3 = as 2, and the scene may hold AnimatedMeshes: the triangles of such a mesh are interpolated at ray.time in the flat loop's mesh traversal,
//     in the two-level traversal and in finish_hit; the scenes' renders go through k_sampler_pass<3> whatever the sampler
// rows of inv and its [3][3] only (x + 12 .. x + 24 are written)
template <int ANIM>
TR_DEV void instance_inv_at(const DevScene& sc, const TrayInstance* __restrict__ in, float time, uint32_t column, float* x) {
    if (ANIM == 1) {
        if (sc.xf_aos) {
            const float4* __restrict__ rec = reinterpret_cast<const float4*>(sc.xf_cache + ((size_t)column * sc.xf_stride + in->moving_slot) * TR_XF_REC + 12u);
#pragma unroll
            for (int q = 0; q < 3; ++q) { const float4 v = rec[q]; x[12 + 4 * q] = v.x; x[13 + 4 * q] = v.y; x[14 + 4 * q] = v.z; x[15 + 4 * q] = v.w; }
            x[24] = rec[3].x;
        } else {
            const uint32_t lanes = sc.xf_cache_lanes;
            const float* __restrict__ col = sc.xf_cache + ((size_t)in->moving_slot * TR_XF_WORDS + 12u) * lanes + column;
#pragma unroll
            for (int k = 0; k < 13; ++k) x[12 + k] = col[(size_t)k * lanes];
        }
    } else {
        eval_xform_stack(sc.xf_levels, sc.keyframes, sc.knots, in->xf_first, in->xf_count, time, x);
    }
}
// rows of inv of moving instance `moving_slot` in the path's column of the wavefront kernels' transform cache (xf_aos = 1)
TR_DEV void instance_inv_cached(const DevScene& sc, uint32_t moving_slot, uint32_t column, float* x) {
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(sc.xf_cache + ((size_t)column * sc.xf_stride + moving_slot) * TR_XF_REC + 12u);
#pragma unroll
    for (int q = 0; q < 3; ++q) { const float4 v = rec[q]; x[12 + 4 * q] = v.x; x[13 + 4 * q] = v.y; x[14 + 4 * q] = v.z; x[15 + 4 * q] = v.w; }
    x[24] = rec[3].x;
}
// rows of inv and its [3][3] (x + 12 .. x + 24) of ANY instance in a kernel built for moving scenes: the path's own if the instance moves
template <int ANIM>
TR_DEV void instance_inv_any(const DevScene& sc, const TrayInstance* __restrict__ in, float time, uint32_t column, float* x) {
    if (in->animated) instance_inv_at<ANIM>(sc, in, time, column, x);
    else {
#pragma unroll
        for (int k = 0; k < 12; ++k) x[12 + k] = in->inv[k];
        x[24] = in->inv[15];
    }
}
template <int ANIM>
TR_DEV void instance_xf_at(const DevScene& sc, const TrayInstance* __restrict__ in, float time, uint32_t column, float* x) {
    if (in->animated) {
        if (ANIM == 1) {
            if (sc.xf_aos) {
                const float4* __restrict__ rec = reinterpret_cast<const float4*>(sc.xf_cache + ((size_t)column * sc.xf_stride + in->moving_slot) * TR_XF_REC);
#pragma unroll
                for (int q = 0; q < TR_XF_WORDS / 4; ++q) { const float4 v = rec[q]; x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
            } else {
                const uint32_t lanes = sc.xf_cache_lanes;
                const float* __restrict__ col = sc.xf_cache + (size_t)in->moving_slot * TR_XF_WORDS * lanes + column;
#pragma unroll
                for (int k = 0; k < 26; ++k) x[k] = col[(size_t)k * lanes];
            }
        } else {
            eval_xform_stack(sc.xf_levels, sc.keyframes, sc.knots, in->xf_first, in->xf_count, time, x);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) { x[k] = in->mat[k]; x[12 + k] = in->inv[k]; }
        x[24] = in->inv[15]; x[25] = in->mat[15];
    }
}

struct HitRec {   // what traversal keeps for the closest candidate
    float t;
    uint32_t inst;
    uint32_t prim;
    float b1, b2;
};

struct Hit {   // DifferentialGeometry in world space (differential_geometry.rs:9-27) + instance id
    f3 p, n, ng, dp_du;
    float u, v;   // texture coordinates (read only by kernels built with image-texture support)
    uint32_t inst;
};

struct Counters { uint32_t rays, vertices; };

// BBox::fast_intersect (bbox.rs:75-104); comparison directions kept so NaNs fall the same way. Straight-line form: the reference's
// early `return false`s only skip work, so the slab updates run unconditionally here and the early verdicts are and-ed in at the end --
// the same boolean for every input, without the four nested branches per box (each a saveexec / cbranch pair on the scalar unit
// that the 16 waves of a CU share: the node step of k_wf_trace_dyn was 80 scalar instructions for 24 of box arithmetic).
TR_DEV bool bbox_hit(const float4 lo, const float4 hi, const f3 o, const f3 inv_dir, const bool nx, const bool ny, const bool nz,
                     float min_t, float max_t) {
    // lo = (bmin.x, bmin.y, bmin.z, bmax.x), hi = (bmax.y, bmax.z, offset, meta)
    float bminx = lo.x, bminy = lo.y, bminz = lo.z, bmaxx = lo.w, bmaxy = hi.x, bmaxz = hi.y;
    float tmin = ((nx ? bmaxx : bminx) - o.x) * inv_dir.x;
    float tmax = ((nx ? bminx : bmaxx) - o.x) * inv_dir.x;
    float tymin = ((ny ? bmaxy : bminy) - o.y) * inv_dir.y;
    float tymax = ((ny ? bminy : bmaxy) - o.y) * inv_dir.y;
    const bool miss_y = tmin > tymax || tymin > tmax;
    tmin = tymin > tmin ? tymin : tmin;
    tmax = tymax < tmax ? tymax : tmax;
    float tzmin = ((nz ? bmaxz : bminz) - o.z) * inv_dir.z;
    float tzmax = ((nz ? bminz : bmaxz) - o.z) * inv_dir.z;
    const bool miss_z = tmin > tzmax || tzmin > tmax;
    tmin = tzmin > tmin ? tzmin : tmin;
    tmax = tzmax < tmax ? tzmax : tmax;
    return !miss_y & !miss_z & (tmin < max_t) & (tmax > min_t);
}

// BBox::fast_intersect that also reports the entry distance it compared with max_t (the flat loop's and the cooperative
// leaf test's gates keep it: a candidate whose box is entered at or behind its own hit distance is a hazard, see trace_flat).
// tmin_out is only read when the box was hit.
TR_DEV bool bbox_hit_t(const float4 lo, const float4 hi, const f3 o, const f3 inv_dir, const bool nx, const bool ny, const bool nz,
                       float min_t, float max_t, float& tmin_out) {
    float bminx = lo.x, bminy = lo.y, bminz = lo.z, bmaxx = lo.w, bmaxy = hi.x, bmaxz = hi.y;
    float tmin = ((nx ? bmaxx : bminx) - o.x) * inv_dir.x;
    float tmax = ((nx ? bminx : bmaxx) - o.x) * inv_dir.x;
    float tymin = ((ny ? bmaxy : bminy) - o.y) * inv_dir.y;
    float tymax = ((ny ? bminy : bmaxy) - o.y) * inv_dir.y;
    const bool miss_y = tmin > tymax || tymin > tmax;
    tmin = tymin > tmin ? tymin : tmin;
    tmax = tymax < tmax ? tymax : tmax;
    float tzmin = ((nz ? bmaxz : bminz) - o.z) * inv_dir.z;
    float tzmax = ((nz ? bminz : bmaxz) - o.z) * inv_dir.z;
    const bool miss_z = tmin > tzmax || tzmin > tmax;
    tmin = tzmin > tmin ? tzmin : tmin;
    tmax = tzmax < tmax ? tzmax : tmax;
    tmin_out = tmin;
    return !miss_y & !miss_z & (tmin < max_t) & (tmax > min_t);
}

// Test part of intersect_triangle (mesh.rs:136-171). Inclusive range test, no back-face culling.
TR_DEV bool triangle_test(const TrayTriVerts* __restrict__ tv, f3 o, f3 d, float min_t, float max_t, float& t_out, float& b1_out, float& b2_out) {
    const float4* q = reinterpret_cast<const float4*>(tv);
    float4 A = q[0], B = q[1], C = q[2];
    f3 pa = mk(A.x, A.y, A.z), pb = mk(B.x, B.y, B.z), pc = mk(C.x, C.y, C.z);
    f3 e0 = pb - pa, e1 = pc - pa;
    f3 s0 = cross(d, e1);
    float dv = dot(s0, e0);
#ifdef TR_TRI_BRANCHES
    if (dv == 0.0f) return false;
    float div = 1.0f / dv;
    f3 dd = o - pa;
    float b1 = dot(dd, s0) * div;
    if (b1 < 0.0f || b1 > 1.0f) return false;
    f3 s1 = cross(dd, e0);
    float b2 = dot(d, s1) * div;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return false;
    float t = dot(e1, s1) * div;
    if (t < min_t || t > max_t) return false;
    t_out = t; b1_out = b1; b2_out = b2;
    return true;
#else
    // straight-line form of mesh.rs:136-171: the early `return None`s only skip work, so every quantity is computed and the verdicts
    // are and-ed (after a rejected test the later values are garbage that nobody reads) -- the lanes of a wave test different
    // triangles, so the four exits were four divergent branches per test
    float div = rcp_rn(dv);
    f3 dd = o - pa;
    float b1 = dot(dd, s0) * div;
    f3 s1 = cross(dd, e0);
    float b2 = dot(d, s1) * div;
    float t = dot(e1, s1) * div;
    const bool ok = (dv != 0.0f) & !(b1 < 0.0f || b1 > 1.0f) & !(b2 < 0.0f || b1 + b2 > 1.0f) & !(t < min_t || t > max_t);
    if (ok) { t_out = t; b1_out = b1; b2_out = b2; }
    return ok;
#endif
}

// AnimatedMeshData::active_keyframes (animated_mesh.rs:56-70) and the interpolation factor of position / normal / texcoord (:72-107) for
// one ray.time: keyframes lo and hi (hi == lo: one keyframe applies -- the time IS a keyframe's, or lies outside the keyframes' range)
struct KeyPair { uint32_t lo, hi; float x; };
TR_DEV KeyPair active_keyframes(const DevScene& sc, uint32_t mesh_id, float time) {
    const TrayMeshKeys mk = sc.mesh_keys[mesh_id];
    const float* __restrict__ times = sc.key_times + mk.time_first;
    uint32_t i = 0u;   // times.binary_search_by(|t| t.partial_cmp(&time)) over ascending, distinct times: Ok(i) if times[i] == time, else Err(first i with times[i] > time)
    while (i < mk.n_keys && times[i] < time) ++i;
    KeyPair kp;
    kp.x = 0.0f;
    if (i < mk.n_keys && times[i] == time) { kp.lo = kp.hi = i; }
    else if (i == mk.n_keys) { kp.lo = kp.hi = i - 1u; }
    else if (i == 0u) { kp.lo = kp.hi = 0u; }
    else { kp.lo = i - 1u; kp.hi = i; kp.x = (time - times[kp.lo]) / (times[kp.hi] - times[kp.lo]); }
    return kp;
}
TR_DEV f3 key_lerp(float x, f3 a, f3 b) { return a * (1.0f - x) + b * x; }   // linalg::lerp (linalg/mod.rs:47-49)
// the triangle in slot `tv` of keyframe 0 (leaf order; keyframe k lies k * stride records on) at the ray's time: AnimatedTriangle::intersect's
// pa, pb, pc (animated_mesh.rs:160-163)
TR_DEV void key_triangle(const TrayTriVerts* __restrict__ tv, uint32_t stride, KeyPair kp, f3& pa, f3& pb, f3& pc) {
    const float4* q = reinterpret_cast<const float4*>(tv + (size_t)kp.lo * stride);
    const float4 A = q[0], B = q[1], C = q[2];
    pa = mk(A.x, A.y, A.z); pb = mk(B.x, B.y, B.z); pc = mk(C.x, C.y, C.z);
    if (kp.hi != kp.lo) {
        const float4* r = reinterpret_cast<const float4*>(tv + (size_t)kp.hi * stride);
        const float4 D = r[0], E = r[1], F = r[2];
        pa = key_lerp(kp.x, pa, mk(D.x, D.y, D.z)); pb = key_lerp(kp.x, pb, mk(E.x, E.y, E.z)); pc = key_lerp(kp.x, pc, mk(F.x, F.y, F.z));
    }
}
TR_DEV bool key_triangle_test(const TrayTriVerts* __restrict__ tv, uint32_t stride, KeyPair kp, f3 o, f3 d, float min_t, float max_t, float& t_out, float& b1_out, float& b2_out) {
    alignas(16) TrayTriVerts now;
    f3 pa, pb, pc;
    key_triangle(tv, stride, kp, pa, pb, pc);
    now.pa[0] = pa.x; now.pa[1] = pa.y; now.pa[2] = pa.z; now.tri_id = 0u;
    now.pb[0] = pb.x; now.pb[1] = pb.y; now.pb[2] = pb.z; now.pad0 = 0u;
    now.pc[0] = pc.x; now.pc[1] = pc.y; now.pc[2] = pc.z; now.pad1 = 0u;
    return triangle_test(&now, o, d, min_t, max_t, t_out, b1_out, b2_out);
}

TR_DEV bool sphere_test(float radius, f3 o, f3 d, float min_t, float max_t, float& t_out) {   // sphere.rs:33-53
    float a = length_sqr(d);
    float b = 2.0f * dot(d, o);
    float c = dot(o, o) - radius * radius;
    float t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return false;
    if (t0 > max_t || t1 < min_t) return false;
    float t_hit = t0;
    if (t_hit < min_t) {
        t_hit = t1;
        if (t_hit > max_t) return false;
    }
    t_out = t_hit;
    return true;
}
TR_DEV bool rect_test(float width, float height, f3 o, f3 d, float min_t, float max_t, float& t_out) {   // rectangle.rs:38-52
    if (fabsf(d.z) < 1e-8f) return false;
    float t = -o.z / d.z;
    if (t < min_t || t > max_t) return false;
    f3 p = o + d * t;
    float hw = width / 2.0f, hh = height / 2.0f;
    if (p.x >= -hw && p.x <= hw && p.y >= -hh && p.y <= hh) { t_out = t; return true; }
    return false;
}
TR_DEV bool disk_test(float radius, float inner_radius, f3 o, f3 d, float min_t, float max_t, float& t_out) {   // disk.rs:42-66
    if (fabsf(d.z) == 0.0f) return false;
    float t = -o.z / d.z;
    if (t < min_t || t > max_t) return false;
    f3 p = o + d * t;
    float dist_sqr = p.x * p.x + p.y * p.y;
    if (dist_sqr > radius * radius || dist_sqr < inner_radius * inner_radius) return false;
    float phi = lm_atan2(p.y, p.x);
    if (phi < 0.0f) phi += kPi * 2.0f;
    if (phi > kPi * 2.0f) return false;
    t_out = t;
    return true;
}

// Per-thread traversal stack in LDS: entry e of thread t lives at base[e * TR_BLOCK + t], so the 64
// lanes of a wave always touch 64 consecutive dwords (conflict free).
#ifndef TR_STACK
#define TR_STACK 24
#endif
#ifndef TR_BLOCK
#define TR_BLOCK 256
#endif
enum : uint32_t { STK_NODE = 0u, STK_INSTANCE = 1u << 30, STK_EXIT_MESH = 2u << 30, STK_KIND_MASK = 3u << 30 };
// Word 7 of a device node (host/gates.hpp: pair_tree): the node's descriptor -- all a traversal needs of a node whose box it has
// tested: where its children (interior: the sibling pair at `offset`) or its primitives (leaf: `count` of them from `offset`) lie and the
// split axis. The top two bits are 0, so a descriptor is a STK_NODE stack entry as it stands.
TR_DEV uint32_t nd_offset(uint32_t desc) { return desc & 0x7fffffu; }
TR_DEV uint32_t nd_count(uint32_t desc) { return (desc >> 23) & 31u; }
TR_DEV uint32_t nd_axis(uint32_t desc) { return (desc >> 28) & 3u; }

// BVH<Triangle>::intersect over one mesh (bvh.rs:81-130, leaf <= 16). Returns true if a triangle was
// accepted; max_t shrinks as candidates are accepted. any_hit: stop at the first accepted candidate.
// Node step: both children of a hit node are fetched and tested together, the far one is pushed only if its
// box is hit now and is re-tested when popped, so the accepted candidates and their order are the reference's
// (bvh.rs:89-127) in about 0.65x the dependent iterations of the one-node-per-step form (measured +6 .. 17 % on
// mesh scenes). leaf_tmin: entry distance of the leaf box of the last accepted triangle (trace_flat's hazard test).
// Called by ALL lanes of the wave (`participate` = this lane has a ray for the mesh; the instance is wave-uniform in trace_flat).
// Written in while-while form -- the node step may repeat TR_WW_NODE_STEPS times while TR_WW_NODE_MIN lanes still have a node to
// test before the (longer) triangle loop runs for the lanes that reached a leaf. Measured on the C4 stand-in at full size
// (871 200 triangles, 32 spp) the setting hardly matters: 1 / 2 / 4 / 8 / 16 steps = 353.0 / 349.9 / 348.8 / 349.8 / 349.6 Msamples/s.
// 4 steps / 12 lanes is kept because the cornell_box kernel (which never runs this code: its meshes take the cooperative test) came
// out 3 % faster with it than with 1 / 1 (772 vs 747 Msamples/s at 64 spp): the tile kernel's speed moves by +-2 % with code it does
// not even execute (register allocation and scheduling of a 14 000-instruction kernel; not the instruction cache, whose miss
// rate is 0.1 %, tools/pmc_icache.sh).
// Whatever the setting, every lane performs the same node tests and triangle tests in the same order.
#ifndef TR_WW_NODE_STEPS
#define TR_WW_NODE_STEPS 4
#endif
#ifndef TR_WW_NODE_MIN
#define TR_WW_NODE_MIN 12
#endif
// DEFORM (ANIM = 3 kernels only): the mesh is an AnimatedMesh, its triangles are taken at the ray's time (`keys`, see active_keyframes)
template <int DEFORM = 0>
TR_DEV bool mesh_traverse_ww(const DevScene& sc, uint32_t* __restrict__ stack, const TrayMesh m, bool participate, f3 o, f3 d, float min_t, float& max_t,
                             bool any_hit, uint32_t& prim, float& b1, float& b2, float& leaf_tmin, KeyPair keys = KeyPair{0u, 0u, 0.0f}) {
    const TrayBvhNode* __restrict__ tree = sc.mesh_nodes + m.node_offset;
    const TrayTriVerts* __restrict__ tris = sc.tri_verts + m.tri_offset;
    const f3 inv_dir = rcp_rn3(d);
    const bool nx = d.x < 0.0f, ny = d.y < 0.0f, nz = d.z < 0.0f;
    enum : uint32_t { MW_NODE = 0u, MW_LEAF = 1u, MW_DONE = 2u };
    int sp = 0;
    bool any = false;
    const uint32_t no_node = 0xffffffffu;
    uint32_t node_a = 0u, node_b = no_node, mode = participate ? MW_NODE : MW_DONE;
    uint32_t leaf_offset = 0u, leaf_count = 0u;
    float leaf_t = 0.0f;
    for (;;) {
#pragma nounroll
        for (int it = 0; it < TR_WW_NODE_STEPS; ++it) {
            const bool in_node = mode == MW_NODE;
            const uint32_t n_node = (uint32_t)__popcll(__ballot(in_node));
            if (n_node == 0u) break;
            if (it > 0 && n_node < TR_WW_NODE_MIN && __any(mode == MW_LEAF)) break;
            if (in_node) {
                const bool two = node_b != no_node;
                const float4* qa = reinterpret_cast<const float4*>(tree + node_a);
                const float4* qb = reinterpret_cast<const float4*>(tree + (two ? node_b : node_a));
                const float4 alo = qa[0], ahi = qa[1], blo = qb[0], bhi = qb[1];
                float ta, tb;
                const bool ha = bbox_hit_t(alo, ahi, o, inv_dir, nx, ny, nz, min_t, max_t, ta);
                const bool hb = bbox_hit_t(blo, bhi, o, inv_dir, nx, ny, nz, min_t, max_t, tb) && two;
                if (ha || hb) {
                    if (ha && hb) { stack[sp * TR_BLOCK] = node_b; ++sp; }
                    const uint32_t offset = __float_as_uint(ha ? ahi.z : bhi.z);
                    const uint32_t meta = __float_as_uint(ha ? ahi.w : bhi.w);
                    const uint32_t count = nd_count(meta), axis = nd_axis(meta);
                    if (count == 0u) {
                        // any-hit rays take the child on the light's side first (the boolean does not depend on the order; C4 stand-in +2.5 %)
                        const bool neg = (axis == 0u ? nx : (axis == 1u ? ny : nz)) != any_hit;
                        node_a = neg ? offset + 1u : offset;   // (device order: the children are the pair at `offset`, host/gates.hpp)
                        node_b = neg ? offset : offset + 1u;
                    } else {
                        mode = MW_LEAF; leaf_offset = offset; leaf_count = count; leaf_t = ha ? ta : tb;
                    }
                } else if (sp == 0) {
                    mode = MW_DONE;
                } else {
                    --sp;
                    node_a = stack[sp * TR_BLOCK]; node_b = no_node;
                }
            }
        }
        if (mode == MW_LEAF) {
            bool stop = false;
            for (uint32_t k = 0; k < leaf_count; ++k) {
                float t, bb1, bb2;
                if (DEFORM ? key_triangle_test(tris + leaf_offset + k, m.tri_count, keys, o, d, min_t, max_t, t, bb1, bb2)
                           : triangle_test(tris + leaf_offset + k, o, d, min_t, max_t, t, bb1, bb2)) {
                    max_t = t; prim = m.tri_offset + leaf_offset + k; b1 = bb1; b2 = bb2; any = true;
                    leaf_tmin = leaf_t;
                    if (any_hit) { stop = true; break; }
                }
            }
            if (stop || sp == 0) {
                mode = MW_DONE;
            } else {
                --sp;
                node_a = stack[sp * TR_BLOCK]; node_b = no_node; mode = MW_NODE;
            }
        }
        if (__all(mode == MW_DONE)) break;
    }
    return any;
}

// Meshes of at most TR_COOP_MAX_TRIS triangles (the reference's cube.obj: 12) in the flat instance loop. Their BVH<Triangle>
// is a dozen nodes deep enough that per-lane traversal costs a wave ~2000 instructions per instance however few of its lanes
// hold a ray that passes the root box (measured: the two cubes were 32 % of the cornell_box kernel). Instead, after the
// root box test (the reference's first test), the lanes of a wave share a test of ALL the mesh's triangles: the n rays that
// pass are staged in LDS and every ray is tested by FOUR lanes, each taking a quarter of the triangles in leaf order, so one
// pass of ceil(T/4) triangle tests serves 16 rays. Returned: the valid triangle of minimal t over ALL triangles (t_out, prim,
// b1, b2), the entry distance of its BVH<Triangle> leaf's box (leaf_tmin) and `hazard`. The reference reaches a triangle only
// through its leaf's box (ancestors contain it and are hit whenever it is); that gate is tested for the SURVIVOR only, with the
// ray's original max_t as the reference tests it before any candidate was accepted: if it passes, the survivor is also the
// minimum of the gated candidates (a subset). `hazard` is set when it fails (only rounding at a flat leaf box produces that), or
// when another valid triangle lies at or below max(t, leaf_tmin) -- a tie, or a rival inside the window in which the survivor's
// gate depends on the order of the reference's traversal (trace_flat explains the rule; testing ungated rivals only adds
// caution); the caller then re-traces the ray the reference's way.
#define TR_COOP_MAX_TRIS 16
#define TR_COOP_WORDS 576   // per wave: ray o, d, min_t, gate max_t (rows 0-7) + the candidate marker k (row 8); the results t, b1, b2, second t of ray r overwrite
                            // rows 0-3 of ITS column once its quad has read them (no other quad ever reads that column): 9 rows instead of 13 keep a
                            // third workgroup within the CU's LDS
// value of lane (l ^ step) of the same quad, step 1 or 2: a DPP quad_perm move (one VALU instruction; __shfl_xor is a ds_bpermute)
TR_DEV float quad_xor(float v, int step) {
#ifdef TR_HOST_EMU
    return __shfl_xor(v, step);
#else
    const int i = __float_as_int(v);
    return __int_as_float(step == 1 ? __builtin_amdgcn_mov_dpp(i, 0xB1, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(i, 0x4E, 0xf, 0xf, true));
#endif
}
TR_DEV uint32_t quad_xor(uint32_t v, int step) {
#ifdef TR_HOST_EMU
    return __shfl_xor(v, step);
#else
    return (uint32_t)(step == 1 ? __builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));
#endif
}
// value of lane (l ^ 7) (wide == 8: DPP row_half_mirror, lane i of an aligned group of eight <-> lane 7 - i) or of lane (l ^ 15) (wide == 16: row_mirror):
// once the four lanes of every quad agree, these pair quad with quad and half-row with half-row
TR_DEV float row_xor(float v, int wide) {
#ifdef TR_HOST_EMU
    return __shfl_xor(v, wide - 1);
#else
    const int i = __float_as_int(v);
    return __int_as_float(wide == 8 ? __builtin_amdgcn_mov_dpp(i, 0x141, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(i, 0x140, 0xf, 0xf, true));
#endif
}
TR_DEV uint32_t row_xor(uint32_t v, int wide) {
#ifdef TR_HOST_EMU
    return __shfl_xor(v, wide - 1);
#else
    return (uint32_t)(wide == 8 ? __builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));
#endif
}
TR_DEV bool mesh_leaf_coop(const DevScene& sc, const TrayMesh m, LdsF w_lds, bool participate, f3 o, f3 d, float min_t,
                           float gate_max_t, float accept_max_t, float& t_out, uint32_t& prim, float& b1, float& b2, float& leaf_tmin, bool& hazard) {
    const uint32_t lane = threadIdx.x & 63u;
    const TrayBvhNode* __restrict__ tree = sc.mesh_nodes + m.node_offset;
    const float4* nq = reinterpret_cast<const float4*>(tree);
    const float4 lo = nq[0], hi = nq[1];
    const uint32_t T = m.tri_count;
    const TrayTriVerts* __restrict__ tris = sc.tri_verts + m.tri_offset;
    const f3 inv_dir = rcp_rn3(d);   // (every lane holds a transformed ray here, wanted or not)
    const bool dnx = d.x < 0.0f, dny = d.y < 0.0f, dnz = d.z < 0.0f;
    const bool need = participate && bbox_hit(lo, hi, o, inv_dir, dnx, dny, dnz, min_t, gate_max_t);
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) return false;
    const uint32_t n = (uint32_t)__popcll(mask);
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
#if defined(TR_HOST_EMU) && defined(TR_COOP_HIST)   // tools/coop_histogram.py: how many rays of a wave reach a small mesh's triangles, how many lanes asked
    { extern unsigned long long tr_coop_hist[65 * 65]; const uint32_t asked = (uint32_t)__popcll(__ballot(participate)); if (lane == (uint32_t)__ffsll((long long)mask) - 1u) tr_coop_hist[n * 65 + asked]++; }
#endif
#ifdef TR_COOP_WIDE_PAYLOAD
    if (need) {
        w_lds[0 * 64 + rank] = o.x; w_lds[1 * 64 + rank] = o.y; w_lds[2 * 64 + rank] = o.z;
        w_lds[3 * 64 + rank] = d.x; w_lds[4 * 64 + rank] = d.y; w_lds[5 * 64 + rank] = d.z;
        w_lds[6 * 64 + rank] = min_t; w_lds[7 * 64 + rank] = gate_max_t;
        w_lds[8 * 64 + rank] = -1.0f;   // no candidate yet
    }
    TR_WAVE_SYNC();
    // Lanes per ray by the number of staged rays (wave-uniform): a wave of incoherent paths stages 5 rays on average, four or fewer in 61 % of the
    // calls and eight or fewer in 88 % (tools/coop_histogram.py, cornell_box at 1080p) -- with four lanes per ray such a call runs ceil(T / 4)
    // triangle tests in sequence on a quarter of the wave. Sixteen lanes per ray (n <= 4) test a cube's twelve triangles at once, eight (n <= 8) in two
    // rounds. The result is the same set function of the ray's valid triangles whatever the width: the minimal t with its triangle, and the smallest
    // other t (c2); between equal t's the choice differs, and a tie sets `hazard` (c2 == t) either way.
#ifdef TR_COOP_QUADS_ONLY
    const uint32_t lsh = 2u;
#else
    const uint32_t lsh = n <= 4u ? 4u : n <= 8u ? 3u : 2u;
#endif
    const uint32_t lpr = 1u << lsh, per = (T + lpr - 1u) >> lsh, g = lane & (lpr - 1u);
    for (uint32_t base = 0; base < n; base += 64u >> lsh) {
        const uint32_t r = base + (lane >> lsh);
        // best candidate of this lane's quarter and the smallest t among its other candidates
        float ct = 0.0f, cb1 = 0.0f, cb2 = 0.0f, ck = -1.0f, c2 = TR_INF;
        if (r < n) {
            const f3 ro = mk(w_lds[0 * 64 + r], w_lds[1 * 64 + r], w_lds[2 * 64 + r]);
            const f3 rd = mk(w_lds[3 * 64 + r], w_lds[4 * 64 + r], w_lds[5 * 64 + r]);
            const float rmin = w_lds[6 * 64 + r], rmax = w_lds[7 * 64 + r];
            for (uint32_t j = 0; j < per; ++j) {
                const uint32_t k = g * per + j;
                if (k < T) {
                    float t, bb1, bb2;
                    if (triangle_test(tris + k, ro, rd, rmin, rmax, t, bb1, bb2)) {
                        if (ck < 0.0f || t < ct) { c2 = ck < 0.0f ? c2 : fminf(c2, ct); ct = t; cb1 = bb1; cb2 = bb2; ck = (float)k; }
                        else c2 = fminf(c2, t);
                    }
                }
            }
        }
        // reduction over the ray's lanes: inside the quads first (all lanes of an aligned group of `lpr` are active together), then quad with quad
#pragma unroll
        for (int step = 1; step <= 2; step <<= 1) {
            const float ot = quad_xor(ct, step), ok = quad_xor(ck, step), ob1 = quad_xor(cb1, step), ob2 = quad_xor(cb2, step),
                        o2 = quad_xor(c2, step);
            const bool both = ok >= 0.0f && ck >= 0.0f;
            const bool take = ok >= 0.0f && (ck < 0.0f || ot < ct);
            c2 = fminf(c2, o2);
            if (both) c2 = fminf(c2, take ? ct : ot);   // the loser of the two bests is the other side's closest rival
            if (take) { ct = ot; ck = ok; cb1 = ob1; cb2 = ob2; }
        }
#ifndef TR_COOP_QUADS_ONLY
        if (lsh >= 3u) {
#pragma unroll
            for (int wide = 8; wide <= 16; wide <<= 1) {
                if (wide == 16 && lsh < 4u) break;
                const float ot = row_xor(ct, wide), ok = row_xor(ck, wide), ob1 = row_xor(cb1, wide), ob2 = row_xor(cb2, wide), o2 = row_xor(c2, wide);
                const bool both = ok >= 0.0f && ck >= 0.0f;
                const bool take = ok >= 0.0f && (ck < 0.0f || ot < ct);
                c2 = fminf(c2, o2);
                if (both) c2 = fminf(c2, take ? ct : ot);
                if (take) { ct = ot; ck = ok; cb1 = ob1; cb2 = ob2; }
            }
        }
#endif
        if (r < n && g == 0u && ck >= 0.0f) {   // (the four lanes of the quad have read column r above; DPP moves below are register-only)
            w_lds[0 * 64 + r] = ct; w_lds[8 * 64 + r] = ck; w_lds[1 * 64 + r] = cb1; w_lds[2 * 64 + r] = cb2; w_lds[3 * 64 + r] = c2;
        }
    }
    TR_WAVE_SYNC();
    bool hit = false;
    if (need) {
        const float ck = w_lds[8 * 64 + rank];
        if (ck >= 0.0f && w_lds[0 * 64 + rank] <= accept_max_t) {
            const float t = w_lds[0 * 64 + rank], c2 = w_lds[3 * 64 + rank];
            const uint32_t k = (uint32_t)ck;
            // the survivor's gate: the box of its BVH<Triangle> leaf, with the ray's original max_t
            const float4* lq = reinterpret_cast<const float4*>(tree + sc.tri_leaf[m.tri_offset + k]);
            float cbox;
            const bool gate = bbox_hit_t(lq[0], lq[1], o, inv_dir, dnx, dny, dnz, min_t, gate_max_t, cbox);
            t_out = t; prim = m.tri_offset + k;
            b1 = w_lds[1 * 64 + rank]; b2 = w_lds[2 * 64 + rank];
            leaf_tmin = cbox;
            hazard = !gate || !(c2 > t && c2 > cbox);   // (also true when cbox is NaN)
            hit = true;
        }
    }
#else
    // The candidate of a lane / of a group of lanes is the pair (t, k) in lexicographic order, "none" = (+inf, COOP_NONE) -- a valid t lies below the
    // ray's max_t, so "none" loses against every candidate and two of them compare equal; c2 = the smallest t among the ray's OTHER valid triangles.
    // The lanes exchange (t, k, c2) only: the barycentrics stay with the lane that found them, which recognises itself as the winner afterwards.
    constexpr uint32_t COOP_NONE = 0x7fffffffu;
    if (need) {
        w_lds[0 * 64 + rank] = o.x; w_lds[1 * 64 + rank] = o.y; w_lds[2 * 64 + rank] = o.z;
        w_lds[3 * 64 + rank] = d.x; w_lds[4 * 64 + rank] = d.y; w_lds[5 * 64 + rank] = d.z;
        w_lds[6 * 64 + rank] = min_t; w_lds[7 * 64 + rank] = gate_max_t;
        w_lds[8 * 64 + rank] = __uint_as_float(COOP_NONE);   // no candidate yet
    }
    TR_WAVE_SYNC();
    // Lanes per ray by the number of staged rays (wave-uniform): a wave of incoherent paths stages 5 rays on average, four or fewer in 61 % of the
    // calls and eight or fewer in 88 % (tools/coop_histogram.py, cornell_box at 1080p) -- with four lanes per ray such a call runs ceil(T / 4)
    // triangle tests in sequence on a quarter of the wave. Sixteen lanes per ray (n <= 4) test a cube's twelve triangles at once, eight (n <= 8) in two
    // rounds. The result is the same set function of the ray's valid triangles whatever the width: the minimal (t, k), and the smallest other t (c2);
    // a tie in t sets `hazard` (c2 == t).
    const uint32_t lsh = n <= 4u ? 4u : n <= 8u ? 3u : 2u;
    const uint32_t lpr = 1u << lsh, per = (T + lpr - 1u) >> lsh, g = lane & (lpr - 1u);
    for (uint32_t base = 0; base < n; base += 64u >> lsh) {
        const uint32_t r = base + (lane >> lsh);
        float ct = TR_INF, cb1 = 0.0f, cb2 = 0.0f, c2 = TR_INF;
        uint32_t ck = COOP_NONE;
        if (r < n) {
            const f3 ro = mk(w_lds[0 * 64 + r], w_lds[1 * 64 + r], w_lds[2 * 64 + r]);
            const f3 rd = mk(w_lds[3 * 64 + r], w_lds[4 * 64 + r], w_lds[5 * 64 + r]);
            const float rmin = w_lds[6 * 64 + r], rmax = w_lds[7 * 64 + r];
            for (uint32_t j = 0; j < per; ++j) {
                const uint32_t k = g * per + j;
                if (k < T) {
                    float t, bb1, bb2;
                    if (triangle_test(tris + k, ro, rd, rmin, rmax, t, bb1, bb2)) {
                        if (t < ct) { c2 = fminf(c2, ct); ct = t; cb1 = bb1; cb2 = bb2; ck = k; }   // (k ascends: of equal t's the first stays)
                        else c2 = fminf(c2, t);
                    }
                }
            }
        }
        const uint32_t own_k = ck;
        // reduction over the ray's lanes: inside the quads first (all lanes of an aligned group of `lpr` are active together), then quad with quad
#define TR_COOP_MERGE(MOVE, ARG) do { \
            const float ot = MOVE(ct, ARG), o2 = MOVE(c2, ARG); const uint32_t ok = MOVE(ck, ARG); \
            const bool take = ot < ct || (ot == ct && ok < ck); \
            c2 = fminf(fminf(c2, o2), take ? ct : ot);   /* the loser of the two bests is a rival of the winner (+inf if there is none) */ \
            if (take) { ct = ot; ck = ok; } } while (0)
        TR_COOP_MERGE(quad_xor, 1);
        TR_COOP_MERGE(quad_xor, 2);
        if (lsh >= 3u) {
            TR_COOP_MERGE(row_xor, 8);
            if (lsh >= 4u) TR_COOP_MERGE(row_xor, 16);
        }
#undef TR_COOP_MERGE
        if (r < n && ck != COOP_NONE) {   // (the lanes of the group have read column r above; the moves are register-only)
            if (g == 0u) { w_lds[0 * 64 + r] = ct; w_lds[8 * 64 + r] = __uint_as_float(ck); w_lds[3 * 64 + r] = c2; }
            if (own_k == ck) { w_lds[1 * 64 + r] = cb1; w_lds[2 * 64 + r] = cb2; }   // exactly one lane of the group holds triangle ck
        }
    }
    TR_WAVE_SYNC();
    bool hit = false;
    if (need) {
        const uint32_t k = __float_as_uint(w_lds[8 * 64 + rank]);
        if (k != COOP_NONE && w_lds[0 * 64 + rank] <= accept_max_t) {
            const float t = w_lds[0 * 64 + rank], c2 = w_lds[3 * 64 + rank];
            // the survivor's gate: the box of its BVH<Triangle> leaf, with the ray's original max_t
            const float4* lq = reinterpret_cast<const float4*>(tree + sc.tri_leaf[m.tri_offset + k]);
            float cbox;
            const bool gate = bbox_hit_t(lq[0], lq[1], o, inv_dir, dnx, dny, dnz, min_t, gate_max_t, cbox);
            t_out = t; prim = m.tri_offset + k;
            b1 = w_lds[1 * 64 + rank]; b2 = w_lds[2 * 64 + rank];
            leaf_tmin = cbox;
            hazard = !gate || !(c2 > t && c2 > cbox);   // (also true when cbox is NaN)
            hit = true;
        }
    }
#endif
    return hit;
}

// Scene::intersect for scenes with a handful of instances: every lane tests the instances, BVH<Instance> leaf by leaf
// (host/gates.hpp), inside one wave-uniform loop. The instance index is uniform, so the transform and the geometry
// parameters are scalar loads and the primitive type never diverges; only BVH<Triangle> traversal is
// per lane.
//
// Exactness versus BVH<Instance>::intersect (bvh.rs:81-130), which visits the instances leaf by leaf in a
// ray-dependent order and reaches one only if its leaf's box passes fast_intersect with the ray's CURRENT max_t.
// Call C the candidates that are valid for the ray's original [min_t, max_t] and whose gates -- the box of the
// instance's BVH<Instance> leaf and, for triangles, the box of their BVH<Triangle> leaf; ancestors contain them and
// are hit whenever they are (the slab arithmetic is monotone in the bounds) -- pass with the ORIGINAL max_t.
//  * any-hit rays (OcclusionTester): until a candidate is accepted the reference's max_t is the original one, so it
//    accepts one iff C is not empty. The loop below returns at the first member of C it meets: the same boolean.
//  * closest-hit rays: let a be the minimum of C (t_a = B), G the largest entry distance of a's gates. If every other
//    member of C has t > max(B, G), the reference returns a whatever its order: when it reaches a its max_t is the
//    original one or the t' of an accepted candidate, t' > G, so a's gates pass and a is accepted; earlier accepted
//    candidates are overwritten, later ones have t > B. (G is usually far below B; for the flat box of an axis-aligned
//    wall G and B are the same number up to rounding, which is why the window is max(B, G) and not B.)
//    The loop tests every instance whose gate passes with the original max_t against the bound max(closest so far, its
//    gate entry): a candidate accepted at or behind the closest one (R1), or a new closest one whose gate is entered at or
//    behind the previous closest (R2), shows that a second candidate lies inside the window -- a tie or a rounding
//    coincidence, about one ray in 1e7 -- and sets `hazard`; trace() re-traces those rays with trace_bvh, which IS the
//    reference's traversal. No member of C inside the window escapes: one tested after a meets the bound max(B, G); one
//    tested before a either was the closest so far when a arrived (R2) or lies behind a closest-so-far that is (R2).
//    BVH<Triangle> traversals start from the ORIGINAL max_t (a start value the reference did not have would prune
//    differently) and report the entry distance of the winning triangle's leaf box, which is part of G.
// The flat loop therefore returns the reference's hit record bit for bit; tests/test_device_emulation.py and
// tests/test_gpu_parity.py check it on random scenes and on scenes built to produce ties (coincident walls).
#ifndef TR_FLAT_MAX
#define TR_FLAT_MAX 16
#endif
// conservative cull by the instance's own (inflated) world box: rejects only when a slab comparison says so, a NaN passes
TR_DEV bool own_box_pass(const float* __restrict__ lo, const float* __restrict__ hi, const f3 o, const f3 inv_dir, const bool nx, const bool ny, const bool nz,
                         float min_t, float max_t) {
    float tmin = ((nx ? hi[0] : lo[0]) - o.x) * inv_dir.x;
    float tmax = ((nx ? lo[0] : hi[0]) - o.x) * inv_dir.x;
    const float tymin = ((ny ? hi[1] : lo[1]) - o.y) * inv_dir.y;
    const float tymax = ((ny ? lo[1] : hi[1]) - o.y) * inv_dir.y;
    const float tzmin = ((nz ? hi[2] : lo[2]) - o.z) * inv_dir.z;
    const float tzmax = ((nz ? lo[2] : hi[2]) - o.z) * inv_dir.z;
    if (tymin > tmin) tmin = tymin;
    if (tzmin > tmin) tmin = tzmin;
    if (tymax < tmax) tmax = tymax;
    if (tzmax < tmax) tmax = tzmax;
    return !(tmin > tmax) && !(tmin >= max_t) && !(tmax <= min_t);
}
// ANIM: an instance that moves within the frame takes the path's own transform (per-path cache) instead of the scalar record's -- the
// loop over leaves and instances stays wave-uniform, only the world -> object transform is per lane. Its gate is the box of its
// BVH<Instance> leaf exactly as for the others (the reference's swept bounds with quirk Q12, whatever they cover: the reference reaches
// the instance through nothing else); the own-box cull of occlusion rays is off for it (host/gates.hpp).
// GUARD_ACTIVE: the range guard of the world-space reciprocal direction (dev_math.h: rcp_rn3) looks at the lanes that hold a ray only. Measured per
// instantiation of the tile kernel (profiles/r05_c2_exact_reciprocal_ab.txt): the LFILT kernels (smallpt) lose 1.2 % of their instructions to the
// compiler's division without it, the others (cornell_box) pay 1.3 % for the mask's scalar registers with it.
// Round 6: FLAT instances (a rectangle or disk that is alone in a flat BVH<Instance> leaf box and does not move: FlatInst::lane_pass, host/gates.hpp)
// are not tested where the wave-uniform loop meets them. The loop only notes, per
// lane, which of them the lane's ray has to test (`pend`, a bit per FlatInst: its gate passed, for occlusion rays its own box too), and a second,
// PER-LANE pass runs transform + primitive test once per noted instance: every lane takes its lowest pending instance, fetches that instance's record
// itself, and the pass repeats while any lane has one left. Why: a BVH<Instance> leaf box of an axis-aligned wall is flat -- a ray passes it only
// where it crosses the wall's rectangle --, so of cornell_box's six rectangles a ray has ONE pending (sometimes the light as well), while the uniform
// loop ran the object-space transform for the whole wave and the rectangle test at 14 - 33 % of the lanes six times per trace
// (profiles/r05_divergence_profiles.txt); the per-lane pass runs 1 - 2 times. Exactness: the accept rule below is the one of the uniform loop and does
// not depend on the order in which a lane meets its candidates (see the argument above: whatever order the reference or this loop visits them in, the
// unique closest candidate wins or `hazard` is set), meshes are still met in loop order, and the gate's entry distance is recomputed from the same
// box with the same arithmetic.
#ifndef TR_FLAT_PEND
#define TR_FLAT_PEND 1
#endif
template <int ANIM, bool GUARD_ACTIVE = false>
TR_DEV bool trace_flat(const DevScene& sc, uint32_t* __restrict__ stack, const Ray& ray, bool any_hit, bool active, HitRec& rec, bool& hazard) {
    const float min_t = ray.min_t, gate_max_t = ray.max_t;
    float max_t = ray.max_t;          // closest accepted candidate so far
    float best_gate = -TR_INF;        // G of that candidate
    bool any = false, done = !active;   // lanes without a ray run along: the cooperative leaf test uses their ALUs
    uint32_t pend = 0u;               // simple instances this lane's ray still has to test (bit = FlatInst index; TR_FLAT_MAX <= 32)
    static_assert(TR_FLAT_MAX <= 32, "one bit per instance of the flat loop");
    // the accept rule of a candidate (R1 / R2 above), shared by the uniform loop and the per-lane pass
#define TR_FLAT_ACCEPT(t_, prim_, b1_, b2_, leaf_t_, hz_, box_t_, inst_id_) do {                                                          \
        const float gate_new_ = fmaxf((box_t_), (leaf_t_));                                                                                \
        const bool closer_ = !any || (t_) < max_t;                                                                                         \
        if (!any_hit) {                                                                                                                    \
            if (!closer_) hazard = true;                              /* R1: a second candidate inside the window of the closest one */    \
            else if (any && !(gate_new_ < max_t)) hazard = true;      /* R2: the previous closest one lies inside the window of the new one */ \
            if ((hz_) || (box_t_) != (box_t_) || (leaf_t_) != (leaf_t_)) hazard = true;   /* rivals inside a small mesh; NaN entry distances */ \
        }                                                                                                                                  \
        if (closer_) {                                                                                                                     \
            max_t = (t_); best_gate = gate_new_;                                                                                           \
            rec.t = (t_); rec.inst = (inst_id_); rec.prim = (prim_); rec.b1 = (b1_); rec.b2 = (b2_);                                       \
        }                                                                                                                                  \
        any = true;                                                                                                                        \
        done = any_hit;                                                                                                                    \
    } while (0)
    const f3 w_inv_dir = rcp_rn3(ray.d, !GUARD_ACTIVE || active);
    const bool wnx = ray.d.x < 0.0f, wny = ray.d.y < 0.0f, wnz = ray.d.z < 0.0f;
#ifdef TR_FLAT_PK
    const f2 wpx = mk2(ray.o.x, ray.d.x), wpy = mk2(ray.o.y, ray.d.y), wpz = mk2(ray.o.z, ray.d.z);
#endif
    // the leaf and instance indices are wave-uniform: the records are read through the constant address space so that boxes,
    // transforms and geometry parameters arrive as scalar loads (SGPRs), not 64 identical lane loads
    typedef const __attribute__((address_space(4))) tray::FlatLeaf* ConstLeaf;
    typedef const __attribute__((address_space(4))) tray::FlatInst* ConstInst;
    const uint32_t n_leaves = sc.n_flat_leaves;
    for (uint32_t l = 0; l < n_leaves; ++l) {
        ConstLeaf lf = (ConstLeaf)(sc.flat_leaves + l);
        const float4 leaf_lo = make_float4(lf->bmin[0], lf->bmin[1], lf->bmin[2], lf->bmax[0]), leaf_hi = make_float4(lf->bmax[1], lf->bmax[2], 0.0f, 0.0f);
        const uint32_t first = lf->first, count = lf->count;
        float box_t = 0.0f;
        // the gate of every instance of this leaf: its box with the ORIGINAL max_t (bvh.rs:89-98)
        const bool gate = bbox_hit_t(leaf_lo, leaf_hi, ray.o, w_inv_dir, wnx, wny, wnz, min_t, gate_max_t, box_t) && !done;
        if (!__any(gate)) continue;
        for (uint32_t k = 0; k < count; ++k) {
            ConstInst in = (ConstInst)(sc.flat_insts + first + k);
            float own_lo[3], own_hi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { own_lo[c] = in->lo[c]; own_hi[c] = in->hi[c]; }
            // occlusion segments are short and end at the light: most instances lie outside the segment's reach for every lane of the
            // wave, and the own-box cull skips them wholesale (measured: trace B 17.3 -> 15.3 % of the tile kernel's wave cycles);
            // unbounded rays pass most boxes, the cull would only cost them its slab test
            #ifdef TR_OWN_BOX_ALWAYS
            const bool own_redundant = false;
#else
            const bool own_redundant = count == 1u && !(ANIM && in->animated != 0u);   // the leaf holds this instance alone: its box lies inside the instance's own inflated box, the gate has said it all
#endif
            const bool wanted = gate && !done && (!any_hit || own_redundant || own_box_pass(own_lo, own_hi, ray.o, w_inv_dir, wnx, wny, wnz, min_t, gate_max_t));
            if (!__any(wanted)) continue;   // nobody's ray comes near this instance
            const uint32_t gt = in->geom_type, mesh_id = in->mesh_id, inst_id = in->inst;
            if (TR_FLAT_PEND && in->lane_pass != 0u) {   // (host/gates.hpp: a rectangle / disk alone behind a flat gate, not moving)
                if (wanted) pend |= 1u << (first + k);   // the per-lane pass below tests it
                continue;
            }
            float inv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) inv[c] = in->inv[c];
            const float gp0 = in->gp0, gp1 = in->gp1;
            // Instance::intersect (receiver.rs:29-35): world ray -> object ray by `inv`, direction not renormalised
            f3 o, d;
            if (ANIM && in->animated != 0u) {   // transform.transform(ray.time) of a moving instance (receiver.rs:30), from the path's cache column
                float x[TR_XF_WORDS];
                instance_inv_at<ANIM>(sc, sc.instances + inst_id, ray.time, ray.col, x);
                o = xf_point_affine_w(x + 12, x[24], ray.o);
                d = xf_vector(x + 12, ray.d);
            } else {
#ifdef TR_FLAT_PK   // (experiment, profiles/r04_tile_kernel_packed_ab.txt: origin and direction side by side in packed arithmetic; the matrix is scalar)
                const f2 a0 = wpx * splat2(inv[0]) + wpy * splat2(inv[1]) + wpz * splat2(inv[2]);
                const f2 a1 = wpx * splat2(inv[4]) + wpy * splat2(inv[5]) + wpz * splat2(inv[6]);
                const f2 a2 = wpx * splat2(inv[8]) + wpy * splat2(inv[9]) + wpz * splat2(inv[10]);
                o = mk(a0.x + inv[3], a1.x + inv[7], a2.x + inv[11]);
                d = mk(a0.y, a1.y, a2.y);
                const float w = inv[12] * ray.o.x + inv[13] * ray.o.y + inv[14] * ray.o.z + inv[15];
                if (w != 1.0f && fabsf(w - 1.0f) < kEps) o = o / w;   // quirk Q5 (xf_point)
#else
                o = xf_point(inv, ray.o);
                d = xf_vector(inv, ray.d);
#endif
            }
            const float bound = fmaxf(max_t, best_gate);   // (best_gate is -inf until a candidate was accepted)
            float t = bound;
            bool hit = false, hz = false;
            uint32_t prim = 0u;
            float b1 = 0.0f, b2 = 0.0f, leaf_t = -TR_INF;
            if (gt == TRAY_GEOM_MESH && sc.coop_offset != 0u && sc.meshes[mesh_id].tri_count <= TR_COOP_MAX_TRIS) {
                // small mesh: the whole wave enters, lanes without a pending ray only lend their ALUs
                const LdsF w_lds = TR_LDS_F(stack - threadIdx.x + sc.coop_offset) + (threadIdx.x >> 6) * TR_COOP_WORDS;
                hit = mesh_leaf_coop(sc, sc.meshes[mesh_id], w_lds, wanted, o, d, min_t, gate_max_t, bound, t, prim, b1, b2, leaf_t, hz);
            } else if (gt == TRAY_GEOM_MESH || (ANIM == 3 && gt == TRAY_GEOM_ANIMATED_MESH)) {   // (the instance is wave-uniform: the whole wave enters the while-while traversal)
                float mt = gate_max_t;   // the mesh's own traversal, from the original max_t
                if (ANIM == 3 && gt == TRAY_GEOM_ANIMATED_MESH)   // AnimatedMesh::intersect: the same traversal over its one tree, triangles at ray.time
                    hit = mesh_traverse_ww<1>(sc, stack, sc.meshes[mesh_id], wanted, o, d, min_t, mt, any_hit, prim, b1, b2, leaf_t, active_keyframes(sc, mesh_id, ray.time)) && mt <= bound;
                else hit = mesh_traverse_ww(sc, stack, sc.meshes[mesh_id], wanted, o, d, min_t, mt, any_hit, prim, b1, b2, leaf_t) && mt <= bound;
                t = mt;
            } else if (wanted) {
                if (gt == TRAY_GEOM_RECT) hit = rect_test(gp0, gp1, o, d, min_t, bound, t);
                else if (gt == TRAY_GEOM_SPHERE) hit = sphere_test(gp0, o, d, min_t, bound, t);
                else hit = disk_test(gp0, gp1, o, d, min_t, bound, t);
            }
            if (hit) TR_FLAT_ACCEPT(t, prim, b1, b2, leaf_t, hz, box_t, inst_id);
        }
        if (__all(done)) break;
    }
    // the per-lane pass over the simple instances noted above
    while (TR_FLAT_PEND && __any(pend != 0u && !done)) {
        if (pend != 0u && !done) {
            const uint32_t k = (uint32_t)__ffsll((long long)pend) - 1u;
            pend &= pend - 1u;
            const tray::FlatInst* __restrict__ in = sc.flat_insts + k;   // (the lane's own record: vector loads; sixteen records at most, cache resident)
            const float4* __restrict__ q = reinterpret_cast<const float4*>(in);
            const float4 m0 = q[0], m1 = q[1], m2 = q[2], m3 = q[3], g1 = q[5], g2 = q[6];   // inv rows 0 .. 3; (hi.y, hi.z, gp0, gp1); (geom_type, mesh_id, inst, animated)
            const uint32_t leaf = in->leaf;
            static_assert(offsetof(tray::FlatInst, gp0) == 88 && offsetof(tray::FlatInst, geom_type) == 96 && offsetof(tray::FlatInst, inst) == 104, "the quarters of a FlatInst record");
            const float inv[16] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w, m3.x, m3.y, m3.z, m3.w};
            const float gp0 = g1.z, gp1 = g1.w;
            const uint32_t gt = __float_as_uint(g2.x), inst_id = __float_as_uint(g2.z);
            const float4* __restrict__ lq = reinterpret_cast<const float4*>(sc.flat_leaves + leaf);
            const float4 l0 = lq[0], l1 = lq[1];   // (bmin, bmax.x), (bmax.y, bmax.z, first, count)
            float box_t = 0.0f;
            // (the gate passed in the uniform loop -- that is why the instance is pending --; this is its entry distance again: same box, same ray, same arithmetic)
            (void)bbox_hit_t(l0, make_float4(l1.x, l1.y, 0.0f, 0.0f), ray.o, w_inv_dir, wnx, wny, wnz, min_t, gate_max_t, box_t);
            const f3 o = xf_point(inv, ray.o), d = xf_vector(inv, ray.d);   // Instance::intersect (receiver.rs:29-35)
            const float bound = fmaxf(max_t, best_gate);
            float t = bound;
            bool hit;
            if (gt == TRAY_GEOM_RECT) hit = rect_test(gp0, gp1, o, d, min_t, bound, t);
            else hit = disk_test(gp0, gp1, o, d, min_t, bound, t);
            if (hit) TR_FLAT_ACCEPT(t, 0u, 0.0f, 0.0f, -TR_INF, false, box_t, inst_id);
        }
    }
#undef TR_FLAT_ACCEPT
    return any;
}

// Scene::intersect. Returns true on hit; rec = closest candidate (the last accepted one,
// bvh.rs:93-98). any_hit: return at the first accepted candidate (OcclusionTester::occluded only
// needs the boolean, light/mod.rs:30-37; the first accepted candidate is the same in both modes).
template <int ANIM>
TR_DEV bool trace_bvh(const DevScene& sc, uint32_t* __restrict__ stack, const Ray& ray, bool any_hit, HitRec& rec) {
    const TrayBvhNode* __restrict__ tree = sc.top_nodes;
    f3 o = ray.o, d = ray.d;
    f3 inv_dir = rcp_rn3(d);
    bool nx = d.x < 0.0f, ny = d.y < 0.0f, nz = d.z < 0.0f;
    const float min_t = ray.min_t;
    float max_t = ray.max_t;
    int sp = 0;
    uint32_t current = 0;          // node index in the current tree
    uint32_t cur_inst = 0;         // instance being traversed when in a mesh
    const TrayTriVerts* __restrict__ tris = nullptr;
    uint32_t tri_base = 0;
    bool in_mesh = false;
    bool any = false;
    bool deforming = false;        // (ANIM = 3) the mesh being traversed is an AnimatedMesh: its triangles at ray.time
    KeyPair keys;
    keys.lo = keys.hi = 0u; keys.x = 0.0f;
    uint32_t key_stride = 0u;
    for (;;) {
        // ---- node test (both levels)
        const float4* nq = reinterpret_cast<const float4*>(tree + current);
        float4 lo = nq[0], hi = nq[1];
        uint32_t offset = __float_as_uint(hi.z);
        uint32_t meta = __float_as_uint(hi.w);   // the node's descriptor
        uint32_t count = nd_count(meta), axis = nd_axis(meta);
        bool descend = false;
        if (bbox_hit(lo, hi, o, inv_dir, nx, ny, nz, min_t, max_t)) {
            if (count == 0u) {   // interior: near child first by the sign of the split axis (bvh.rs:105-119)
                bool neg = (axis == 0u ? nx : (axis == 1u ? ny : nz)) != any_hit;   // (any-hit: light's side first, see mesh_traverse_ww)
                uint32_t far_child = neg ? offset : offset + 1u;   // (device order: the children are the pair at `offset`, host/gates.hpp)
                current = neg ? offset + 1u : offset;
                stack[sp * TR_BLOCK] = far_child;
                ++sp;
                descend = true;
            } else if (in_mesh) {   // BVH<Triangle> leaf (<= 16 triangles), tested in order
                for (uint32_t k = 0; k < count; ++k) {
                    float t, bb1, bb2;
                    const bool tri_hit = (ANIM == 3 && deforming) ? key_triangle_test(tris + offset + k, key_stride, keys, o, d, min_t, max_t, t, bb1, bb2)
                                                                  : triangle_test(tris + offset + k, o, d, min_t, max_t, t, bb1, bb2);
                    if (tri_hit) {
                        max_t = t;
                        rec.t = t; rec.inst = cur_inst; rec.prim = tri_base + offset + k; rec.b1 = bb1; rec.b2 = bb2;
                        any = true;
                        if (any_hit) return true;
                    }
                }
            } else {   // BVH<Instance> leaf (<= 4 instances): queue them so that pops come in leaf order
                for (uint32_t k = count; k > 0u; --k) {
                    stack[sp * TR_BLOCK] = STK_INSTANCE | (offset + k - 1u);
                    ++sp;
                }
            }
        }
        if (descend) continue;
        // ---- pop until there is a node to test
        bool have_node = false;
        while (sp > 0) {
            --sp;
            uint32_t e = stack[sp * TR_BLOCK];
            uint32_t kind = e & STK_KIND_MASK;
            if (kind == STK_NODE) { current = e; have_node = true; break; }
            if (kind == STK_EXIT_MESH) {   // back to world space and the top-level tree
                in_mesh = false;
                tree = sc.top_nodes;
                o = ray.o; d = ray.d;
                inv_dir = rcp_rn3(d);
                nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f;
                continue;
            }
            // Instance::intersect (receiver.rs:29-35): world ray -> object ray by `inv`; the direction is
            // not renormalised, so t is shared between the two spaces
            uint32_t i = sc.top_order[e & ~STK_KIND_MASK];
            const TrayInstance* __restrict__ in = sc.instances + i;
            if (in->kind == TRAY_INST_POINT_EMITTER) continue;   // emitter.rs:120
            f3 lo_, ld;
            if (ANIM && in->animated) {   // transform.transform(ray.time) per visit (receiver.rs:30)
                float x[TR_XF_WORDS];
                instance_inv_at<ANIM>(sc, in, ray.time, ray.col, x);
                lo_ = xf_point_affine_w(x + 12, x[24], ray.o);
                ld = xf_vector(x + 12, ray.d);
            } else {
                lo_ = xf_point(in->inv, ray.o);
                ld = xf_vector(in->inv, ray.d);
            }
            uint32_t gt = in->geom_type;
            if (gt == TRAY_GEOM_MESH || (ANIM == 3 && gt == TRAY_GEOM_ANIMATED_MESH)) {
                const TrayMesh m = sc.meshes[in->mesh_id];
                if (ANIM == 3) {   // AnimatedMesh::intersect (animated_mesh.rs:130-134): the same traversal over ITS tree, triangles at ray.time
                    deforming = gt == TRAY_GEOM_ANIMATED_MESH;
                    if (deforming) { keys = active_keyframes(sc, in->mesh_id, ray.time); key_stride = m.tri_count; }
                }
                stack[sp * TR_BLOCK] = STK_EXIT_MESH;
                ++sp;
                in_mesh = true;
                cur_inst = i;
                tree = sc.mesh_nodes + m.node_offset;
                tris = sc.tri_verts + m.tri_offset;
                tri_base = m.tri_offset;
                o = lo_; d = ld;
                inv_dir = rcp_rn3(d);
                nx = d.x < 0.0f; ny = d.y < 0.0f; nz = d.z < 0.0f;
                current = 0;
                have_node = true;
                break;
            }
            float t;
            bool hit;
            if (gt == TRAY_GEOM_RECT) hit = rect_test(in->geom_params[0], in->geom_params[1], lo_, ld, min_t, max_t, t);
            else if (gt == TRAY_GEOM_SPHERE) hit = sphere_test(in->geom_params[0], lo_, ld, min_t, max_t, t);
            else hit = disk_test(in->geom_params[0], in->geom_params[1], lo_, ld, min_t, max_t, t);
            if (hit) {
                max_t = t;
                rec.t = t; rec.inst = i; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
                any = true;
                if (any_hit) return true;
            }
        }
        if (!have_node) break;
    }
    return any;
}

// Scene::intersect (scene.rs:148-150). The tile kernel calls this from exactly one site (every
// lane traces one ray per step of its phase machine), so it is inlined there.
struct TraceResult { HitRec rec; bool hit; };
// Called by ALL lanes of a wave that has at least one ray; `active` = this lane has one.
template <int ANIM, bool GUARD_ACTIVE = false>
TR_DEV TraceResult trace(const DevScene* __restrict__ scp, uint32_t* __restrict__ stack, Ray ray, bool any_hit, bool active) {
    const DevScene& sc = *scp;
    TraceResult r;
    r.rec.t = 0.0f; r.rec.inst = 0xffffffffu; r.rec.prim = 0u; r.rec.b1 = 0.0f; r.rec.b2 = 0.0f;
    // (moving scenes too since round 4: the flat loop's gates are the boxes of the BVH<Instance> leaves -- for a moving scene the reference's
    // swept bounds of animated_transform.rs:58-71 with quirk Q12 -- so it reaches exactly the instances the reference's traversal can reach)
    if (sc.n_instances <= TR_FLAT_MAX) {
        bool hazard = false;
        r.hit = trace_flat<ANIM, GUARD_ACTIVE>(sc, stack, ray, any_hit, active, r.rec, hazard);
        if (__any(hazard)) {   // (about one ray in 1e7: tied candidates, or a box entered behind its own hit) the reference's traversal decides
            if (hazard) {
                if (sc.retraced) atomicAdd(sc.retraced, 1u);
                r.rec.t = 0.0f; r.rec.inst = 0xffffffffu; r.rec.prim = 0u; r.rec.b1 = 0.0f; r.rec.b2 = 0.0f;
                r.hit = trace_bvh<ANIM>(sc, stack, ray, false, r.rec);
            }
        }
    } else r.hit = active ? trace_bvh<ANIM>(sc, stack, ray, any_hit, r.rec) : false;
    return r;
}

// Rebuilds the DifferentialGeometry of the final candidate in object space and moves it to world
// space (receiver.rs:36-42; DifferentialGeometry::{new,with_normal} differential_geometry.rs:32-64).
template <int ANIM>
TR_DEV Hit finish_hit(const DevScene& sc, const Ray& ray, const HitRec& rec, float* uv_out = nullptr, f3* dp_dv_out = nullptr) {
    const TrayInstance* __restrict__ in = sc.instances + rec.inst;
    float x[TR_XF_WORDS];
    f3 o, d;
    if (ANIM) {
        instance_xf_at<ANIM>(sc, in, ray.time, ray.col, x);
        o = xf_point_affine_w(x + 12, x[24], ray.o);
        d = xf_vector(x + 12, ray.d);
    } else {
        o = xf_point(in->inv, ray.o);
        d = xf_vector(in->inv, ray.d);
    }
    f3 p = o + d * rec.t;
    f3 n, ng, dp_du, dp_dv;
    float u = 0.0f, v = 0.0f;
    uint32_t gt = in->geom_type;
    if (gt == TRAY_GEOM_RECT) {   // rectangle.rs:53-60
        float hw = in->geom_params[0] / 2.0f, hh = in->geom_params[1] / 2.0f;
        u = (p.x + hw) / (2.0f * hw); v = (p.y + hh) / (2.0f * hh);
        dp_du = mk(hw * 2.0f, 0.0f, 0.0f); dp_dv = mk(0.0f, hh * 2.0f, 0.0f);
        n = normalized(cross(dp_du, dp_dv));
        ng = normalized(mk(0.0f, 0.0f, 1.0f));
    } else if (gt == TRAY_GEOM_SPHERE) {   // sphere.rs:56-81
        float radius = in->geom_params[0];
        float theta = lm_acos(clampf(p.z / radius, -1.0f, 1.0f));
        float inv_z = 1.0f / sqrtf(p.x * p.x + p.y * p.y);
        float cos_phi = p.x * inv_z, sin_phi = p.y * inv_z;
        u = lm_atan2(p.x, p.y) / (2.0f * kPi);
        if (u < 0.0f) u = u + 1.0f;
        v = theta / kPi;
        dp_du = mk(-kPi * 2.0f * p.y, kPi * 2.0f * p.x, 0.0f);
        dp_dv = mk(p.z * cos_phi, p.z * sin_phi, -radius * lm_sin(theta)) * kPi;
        n = normalized(p);
        ng = n;
    } else if (gt == TRAY_GEOM_MESH || (ANIM == 3 && gt == TRAY_GEOM_ANIMATED_MESH)) {   // mesh.rs:172-197
        const float4* q = reinterpret_cast<const float4*>(sc.tri_verts + rec.prim);
        float4 A = q[0], B = q[1], C = q[2];
        f3 pa = mk(A.x, A.y, A.z), pb = mk(B.x, B.y, B.z), pc = mk(C.x, C.y, C.z);
        const float4* aq = reinterpret_cast<const float4*>(sc.tri_attrs + rec.prim);
        float4 a0 = aq[0], a1 = aq[1], a2 = aq[2], a3 = aq[3];
        f3 na = mk(a0.x, a0.y, a0.z), nb = mk(a0.w, a1.x, a1.y), nc = mk(a1.z, a1.w, a2.x);
        f3 ta = mk(a2.y, a2.z, 0.0f), tb = mk(a2.w, a3.x, 0.0f), tc = mk(a3.y, a3.z, 0.0f);
        if (ANIM == 3 && gt == TRAY_GEOM_ANIMATED_MESH) {   // AnimatedTriangle::intersect (animated_mesh.rs:160-172): everything at ray.time (rec.prim: the slot in keyframe 0)
            const KeyPair kp = active_keyframes(sc, in->mesh_id, ray.time);
            const uint32_t stride = sc.meshes[in->mesh_id].tri_count;
            key_triangle(sc.tri_verts + rec.prim, stride, kp, pa, pb, pc);
            const float4* lq = reinterpret_cast<const float4*>(sc.tri_attrs + rec.prim + (size_t)kp.lo * stride);
            a0 = lq[0]; a1 = lq[1]; a2 = lq[2]; a3 = lq[3];
            na = mk(a0.x, a0.y, a0.z); nb = mk(a0.w, a1.x, a1.y); nc = mk(a1.z, a1.w, a2.x);
            ta = mk(a2.y, a2.z, 0.0f); tb = mk(a2.w, a3.x, 0.0f); tc = mk(a3.y, a3.z, 0.0f);
            if (kp.hi != kp.lo) {
                const float4* hq = reinterpret_cast<const float4*>(sc.tri_attrs + rec.prim + (size_t)kp.hi * stride);
                const float4 h0 = hq[0], h1 = hq[1], h2 = hq[2], h3 = hq[3];
                na = key_lerp(kp.x, na, mk(h0.x, h0.y, h0.z)); nb = key_lerp(kp.x, nb, mk(h0.w, h1.x, h1.y)); nc = key_lerp(kp.x, nc, mk(h1.z, h1.w, h2.x));
                ta = key_lerp(kp.x, ta, mk(h2.y, h2.z, 0.0f)); tb = key_lerp(kp.x, tb, mk(h2.w, h3.x, 0.0f)); tc = key_lerp(kp.x, tc, mk(h3.y, h3.z, 0.0f));
            }
        }
        float b1 = rec.b1, b2 = rec.b2;
        float b0 = 1.0f - b1 - b2;
        n = normalized(normalized(b0 * na + b1 * nb + b2 * nc));   // normalised in mesh.rs:172 AND in DifferentialGeometry::with_normal (differential_geometry.rs:51)
        ng = n;
        f3 texcoord = b0 * ta + b1 * tb + b2 * tc;
        u = texcoord.x; v = texcoord.y;
        float du0 = ta.x - tc.x, du1 = tb.x - tc.x;
        float dv0 = ta.y - tc.y, dv1 = tb.y - tc.y;
        float det = du0 * dv1 - dv0 * du1;
        if (det == 0.0f) {
            f3 e0 = pb - pa, e1 = pc - pa;
            coordinate_system(normalized(cross(e1, e0)), dp_du, dp_dv);
        } else {
            det = 1.0f / det;
            f3 dp0 = pa - pc, dp1 = pb - pc;
            dp_du = (dv1 * dp0 - dv0 * dp1) * det;
            dp_dv = (-du1 * dp0 + du0 * dp1) * det;
        }
    } else {   // disk.rs:67-75
        float radius = in->geom_params[0], inner_radius = in->geom_params[1];
        float dist_sqr = p.x * p.x + p.y * p.y;
        float phi = lm_atan2(p.y, p.x);
        if (phi < 0.0f) phi += kPi * 2.0f;
        float hit_radius = sqrtf(dist_sqr);
        u = phi / (2.0f * kPi);
        v = 1.0f - (hit_radius - inner_radius) / (radius - inner_radius);
        dp_du = mk(-kPi * 2.0f * p.y, kPi * 2.0f * p.x, 0.0f);
        dp_dv = ((inner_radius - radius) / hit_radius) * mk(p.x, p.y, 0.0f);
        n = normalized(cross(dp_du, dp_dv));
        ng = normalized(mk(0.0f, 0.0f, 1.0f));
    }
    Hit h;
    if (ANIM) {
        h.p = xf_point_affine_w(x, x[25], p);
        h.n = xf_normal_t(x + 12, n);
        h.ng = xf_normal_t(x + 12, ng);
        h.dp_du = xf_vector(x, dp_du);
        if (dp_dv_out) *dp_dv_out = xf_vector(x, dp_dv);
    } else {
        h.p = xf_point(in->mat, p);
        h.n = xf_normal_t(in->inv, n);
        h.ng = xf_normal_t(in->inv, ng);
        h.dp_du = xf_vector(in->mat, dp_du);
        if (dp_dv_out) *dp_dv_out = xf_vector(in->mat, dp_dv);
    }
    h.inst = rec.inst;
    h.u = u; h.v = v;
    if (uv_out) { uv_out[0] = u; uv_out[1] = v; }
    return h;
}

// Geometry normal of the final candidate only (what estimate_direct's BSDF half needs from the hit,
// mod.rs:159): same arithmetic as finish_hit restricted to ng.
template <int ANIM>
TR_DEV f3 finish_hit_ng(const DevScene& sc, const Ray& ray, const HitRec& rec) {
    const TrayInstance* __restrict__ in = sc.instances + rec.inst;
    uint32_t gt = in->geom_type;
    f3 ng;
    float x[TR_XF_WORDS];
    if (ANIM) instance_inv_any<ANIM>(sc, in, ray.time, ray.col, x);   // (only the inverse is used below)
    if (gt == TRAY_GEOM_SPHERE) {
        f3 o = ANIM ? xf_point_affine_w(x + 12, x[24], ray.o) : xf_point(in->inv, ray.o);
        f3 d = ANIM ? xf_vector(x + 12, ray.d) : xf_vector(in->inv, ray.d);
        ng = normalized(o + d * rec.t);
    } else if (gt == TRAY_GEOM_MESH) {
        const float4* aq = reinterpret_cast<const float4*>(sc.tri_attrs + rec.prim);
        float4 a0 = aq[0], a1 = aq[1], a2 = aq[2];
        f3 na = mk(a0.x, a0.y, a0.z), nb = mk(a0.w, a1.x, a1.y), nc = mk(a1.z, a1.w, a2.x);
        float b1 = rec.b1, b2 = rec.b2;
        float b0 = 1.0f - b1 - b2;
        ng = normalized(normalized(b0 * na + b1 * nb + b2 * nc));   // twice, as in finish_hit
    } else {
        ng = normalized(mk(0.0f, 0.0f, 1.0f));
    }
    return ANIM ? xf_normal_t(x + 12, ng) : xf_normal_t(in->inv, ng);
}

// ---- Sampleable (object space) -------------------------------------------------------------
TR_DEV f3 uniform_sample_sphere(float u0, float u1) {   // mc.rs:84-89
    float z = 1.0f - 2.0f * u0;
    float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float phi = kPi * 2.0f * u1;
    float sn, cs;
    lm_sincos(phi, sn, cs);
    return mk(cs * r, sn * r, z);
}
TR_DEV f3 uniform_sample_cone_frame(float u0, float u1, float cos_theta_max, f3 wx, f3 wy, f3 wz) {   // mc.rs:76-82
    float cos_theta = lerpf(u0, cos_theta_max, 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = u1 * kPi * 2.0f;
    float sn, cs;
    lm_sincos(phi, sn, cs);
    return cs * sin_theta * wx + sn * sin_theta * wy + cos_theta * wz;
}

// Sampleable::sample: point + normal on the emitter's geometry as seen from object-space p
TR_DEV void geom_sample(const TrayInstance* __restrict__ in, f3 p, float u0, float u1, f3& ps, f3& ns) {
    uint32_t gt = in->geom_type;
    if (gt == TRAY_GEOM_RECT) {   // rectangle.rs:77-83
        float w = in->geom_params[0], h = in->geom_params[1];
        ps = mk(u0 * w - w / 2.0f, u1 * h - h / 2.0f, 0.0f);
        ns = mk(0.0f, 0.0f, 1.0f);
    } else if (gt == TRAY_GEOM_DISK) {   // disk.rs:85-93
        float dx, dy;
        concentric_sample_disk(u0, u1, dx, dy);
        ps = mk(dx * in->geom_params[0], dy * in->geom_params[0], 0.0f);
        ns = mk(0.0f, 0.0f, 1.0f);
    } else {   // sphere.rs:92-123
        float radius = in->geom_params[0];
        float dist_sqr = length_sqr(p - mk(0.0f, 0.0f, 0.0f));
        if (dist_sqr - radius * radius < 0.0001f) {
            ps = mk(0.0f, 0.0f, 0.0f) + radius * uniform_sample_sphere(u0, u1);
            ns = normalized(ps);
            return;
        }
        f3 w_z = normalized(mk(0.0f, 0.0f, 0.0f) - p);
        f3 w_x, w_y;
        coordinate_system(w_z, w_x, w_y);
        float cos_theta_max = sqrtf(fmaxf(0.0f, 1.0f - radius * radius / dist_sqr));
        f3 dir = normalized(uniform_sample_cone_frame(u0, u1, cos_theta_max, w_x, w_y, w_z));
        float t;
        if (sphere_test(radius, p, dir, 0.0f, TR_INF, t)) {
            ps = p + dir * t;
            ns = normalized(ps);   // dg.ng of with_normal
        } else {
            float tt = dot(mk(0.0f, 0.0f, 0.0f) - p, dir);
            ps = p + dir * tt;
            ns = normalized(ps);
        }
    }
}
// Sampleable::pdf (rectangle.rs:91-104, disk.rs:97-110, sphere.rs:131-140)
TR_DEV float geom_pdf(const TrayInstance* __restrict__ in, f3 p, f3 w_i) {
    uint32_t gt = in->geom_type;
    if (gt == TRAY_GEOM_SPHERE) {
        float radius = in->geom_params[0];
        float dist_sqr = length_sqr(p - mk(0.0f, 0.0f, 0.0f));
        if (dist_sqr - radius * radius < 0.0001f) return 1.0f / (4.0f * kPi * radius);   // quirk Q7
        float cos_theta_max = sqrtf(fmaxf(0.0f, 1.0f - radius * radius / dist_sqr));
        return 1.0f / (kPi * 2.0f * (1.0f - cos_theta_max));   // mc::uniform_cone_pdf
    }
    float t, area;
    bool hit;
    f3 n;
    if (gt == TRAY_GEOM_RECT) {
        hit = rect_test(in->geom_params[0], in->geom_params[1], p, w_i, 0.001f, TR_INF, t);
        area = in->geom_params[0] * in->geom_params[1];
        float hw = in->geom_params[0] / 2.0f, hh = in->geom_params[1] / 2.0f;
        n = normalized(cross(mk(hw * 2.0f, 0.0f, 0.0f), mk(0.0f, hh * 2.0f, 0.0f)));
    } else {
        hit = disk_test(in->geom_params[0], in->geom_params[1], p, w_i, 0.001f, TR_INF, t);
        area = kPi * (in->geom_params[0] * in->geom_params[0] - in->geom_params[1] * in->geom_params[1]);
        if (hit) {
            f3 ph = p + w_i * t;
            float hit_radius = sqrtf(ph.x * ph.x + ph.y * ph.y);
            f3 dp_du = mk(-kPi * 2.0f * ph.y, kPi * 2.0f * ph.x, 0.0f);
            f3 dp_dv = ((in->geom_params[1] - in->geom_params[0]) / hit_radius) * mk(ph.x, ph.y, 0.0f);
            n = normalized(cross(dp_du, dp_dv));
        } else {
            n = mk(0.0f, 0.0f, 1.0f);
        }
    }
    if (!hit) return 0.0f;
    f3 w = -w_i;
    float pdf = length_sqr(p - (p + w_i * t)) / (fabsf(dot(n, w)) * area);
    return isfinite(pdf) ? pdf : 0.0f;
}

}  // namespace tr
