// Wavefront formulation of the tile worker: the path vertex step of dev_integrator.h cut into
// stage kernels over an HBM-resident SoA path pool (this is the formulation SURVEY §8(d)'s
// 368 B / vertex accounting describes).
//
//   pool      : F_COUNT f32/u32 fields x N slots, field-major (slot i of field f at pool[f*N + i]) so that
//               every load / store of a stage kernel is a fully coalesced 256-B wave access
//   chunk     : 256 consecutive slots = one workgroup = one 8x8 tile at a time (slot k <-> pixel k & 63,
//               samples (k >> 6), +4, ... exactly like k_path_tiles), pulled from the Morton queue with a
//               global atomic counter (block_queue.rs:52-59) when the previous tile of the chunk is complete
//   per round : k_wf_advance (stage C shading of the few vertices that traced a BSDF-sampled light ray, film splat of finished
//               samples, tile switch, queues A and R) -> k_wf_regen (new camera samples for queue R: camera ray, spline stacks
//               of moving instances into the per-slot cache, queue A) -> k_wf_trace_dyn<A> -> k_wf_begin (queue B) ->
//               k_wf_trace_dyn<B> -> k_wf_query (BSDF queries; vertex_end unless a stage C ray is pending; queue C) ->
//               k_wf_trace_dyn<C> (STAGE 0: camera / continuation ray, closest hit -> rec, WF_HIT_A; 1: occlusion ray of the light
//               sample, any hit -> WF_OCCLUDED; 2: BSDF-sampled ray of estimate_direct, closest hit -> rec, WF_HIT_C).
//   film      : per-chunk row bins in global memory (workgroup-private, L2 resident), resolved through an LDS
//               window and flushed with global f32 atomics once per tile (same arithmetic as k_path_tiles)
//
// Each stage kernel only touches the fields it needs, has a small register footprint (no spills, high
// occupancy) and a single divergent concern. The arithmetic per camera sample is bit-identical to the
// megakernel's: both call the same device functions.
#pragma once

namespace tr {

// counters are spread over WF_STAT_SLOTS records (one hot word would serialise 65536 wave atomics per launch)
#define WF_STAT_SLOTS 1024
enum : uint32_t {   // flags that only exist between wavefront stages
    WF_HIT_A = 32u, WF_OCCLUDED = 64u, WF_HIT_C = 128u, WF_INVERTEX = 256u, WF_FINISHED = 512u,
    WF_CFLIGHT = 4096u,  // the vertex's BSDF-sampled light ray (stage C) travels with this round's stage A rays (WF_FOLD_C)
    WF_PENDING = 8192u   // fused shading (k_wf_shade_kind): the vertex's light term was computed before its occlusion ray was traced -- (F_PEND_TV, F_PEND_DL)
                         // hold the factors, F_PEND_OCC the traversal's verdict; whoever touches the slot's radiance next applies `illum += tv * dl` if it says "free"
};
// Round 5: no traversal launch of its own for stage C. A few thousand of a round's millions of vertices trace a BSDF-sampled light ray
// (estimate_direct's second half, mod.rs:141-166, for the samples mis_ray_filter could not dismiss); their launch was 95 % tail -- 0.5 ms per round
// for the longest chain of dependent fetches, 3.3 % of a frame with its fallback launch. Now k_wf_advance of the NEXT round puts such a ray into
// queue A beside the continuation rays (marked WF_CFLIGHT; the vertex stays open), k_wf_trace_dyn<A> writes its result as WF_HIT_C, k_wf_begin and
// the query kernels pass the slot by, and k_wf_advance of the round after closes the vertex (vertex_end) as before. The slot spends one round more
// on that vertex; every value is what it was -- both traversals are the same closest-hit search over [0.001, inf).
#ifndef WF_FOLD_C
#define WF_FOLD_C 1
#endif

// Pool layout (round 5): RECORDS for what is read or written in QUEUE order, field-major arrays for the rest.
// The kernels that run one thread per pool slot (k_wf_advance, k_wf_begin) read and write consecutive slots; the others take their slots
// from a queue -- the traversal kernels in octant order, the shading kernels in material order, k_wf_regen in the order samples finished --
// i.e. scattered over the chunk. Rounds 2-4 kept every field field-major (slot i of field f at pool[f * N + i]): for the queue-ordered
// kernels every 4-byte access then touched a cache line of its own -- k_wf_trace_dyn<A> wrote 752 B per camera sample for 175 B of results,
// the query kernels fetched ~430 B per vertex for 148 B of fields (profiles/r05_c5_traffic_by_kernel.txt). Now the words that travel together
// lie together, one record per slot and group, the records of a group side by side (array of structures per group):
//   HIT   8 words  hit record of the slot's last closest-hit ray                  traversal -> k_wf_begin / k_wf_advance
//   RAY   8 words  origin, direction of the ray towards the next vertex, bounce, sample key
//                                                                                 k_wf_regen / query -> k_wf_advance (queue A) / k_wf_begin / query
//   THRU  8 words  throughput, radiance so far, ray.time                          k_wf_regen / query <-> k_wf_begin / query / k_wf_advance
//   VERT 20 words  shading frame (p, n, tan), material, the light sample          k_wf_begin -> query
// A record access of a slot-ordered kernel is as dense as before (a wave covers 64 consecutive records); the queue-ordered kernels move the
// bytes they use. Everything else (first hit's normal, film position, the extras of the few vertices with a stage C ray, texture
// coordinates) is touched by slot-ordered kernels only, or rarely, and stays field-major.
// Not stored at all: `direct` and `t_vertex` between k_wf_begin and the query kernels (vertex_begin sets them to 0 and to the throughput,
// which the query kernels have), `w_o` (it is -d), the occlusion segment's direction (the queue entry carries it; k_wf_trace_fallback reads
// the deferred ray's own record).
enum {
    F_REC_T, F_REC_INST, F_REC_PRIM, F_REC_B1, F_REC_B2, F_HIT_PAD0, F_HIT_PAD1, F_HIT_PAD2,   // HIT
    F_O, F_D = F_O + 3, F_BOUNCE = F_D + 3, F_KS,                                            // RAY
    F_T, F_ILLUM = F_T + 3, F_TIME = F_ILLUM + 3, F_KIDX,                                     // THRU (F_KIDX: the path's shutter-time index, moving scenes in table mode)
    F_P, F_N = F_P + 3, F_TAN = F_N + 3, F_MAT = F_TAN + 3, F_LINST, F_LI, F_WL = F_LI + 3, F_PDFL = F_WL + 3, F_VERT_PAD0, F_VERT_PAD1,   // VERT (bitan = cross(tan, n) is recomputed)
    F_SOA,                                                                                    // ---- field-major from here on
    F_FLAGS = F_SOA,   // (field-major: k_wf_advance scans the flags of EVERY slot every round -- as a word of the hit record that was 32 bytes read and 32 written
                       // per slot and round for 4, 44 % of the kernel's traffic; the traversal's result write is two pieces instead of one for it)
    F_SNEXT, F_SX, F_SY, F_NG,
    F_AUX = F_NG + 3, F_DIRECT = F_AUX + 3, F_MISF = F_DIRECT + 3, F_TV = F_MISF + 3,   // between the query kernels and k_wf_advance, for the few vertices with a stage C ray
    F_U = F_TV + 3, F_V,   // hit.dg.u / v of the vertex (scenes with image textures)
    F_COUNT
};
// (fused shading: the vertex record's normal / tangent / material words hold the pending light term -- nobody stores a shading frame there in that schedule)
enum { F_PEND_TV = F_N, F_PEND_DL = F_TAN, F_PEND_OCC = F_MAT };
#define WF_HIT_WORDS 8
static_assert(F_O == 8 && F_T == 16 && F_P == 24 && F_SOA == 44, "record groups of the pool: 8 (hit) + 16 (ray | throughput: one 64-byte record, pidx) + 20 (vertex) words");

struct WfPool {
    float* __restrict__ data;   // F_COUNT * n_slots dwords
    uint32_t n_slots;
    uint32_t seg_cap;           // entries per segment of every queue (wf_seg_cap(chunks of the pool)); a queue holds WF_SEGS * seg_cap
    uint32_t first = 0u;        // a view's first slot (kernels.hip: WfView): slot i of the view is slot first + i of the pool
};
struct WfChunk { uint32_t tile, done, next_pair; };   // tile: index into the work list, WF_TILE_NEED or WF_TILE_IDLE; done: finished samples of the
                                                       // tile; next_pair: first (pixel, sample) pair of the tile not handed to a slot yet
enum : uint32_t { WF_TILE_NEED = 0xfffffffeu, WF_TILE_IDLE = 0xffffffffu };

// (f is a constant at every call site: the layout test folds)
TR_DEV size_t pidx(const WfPool& p, int f, uint32_t i) {
    const size_t n = p.n_slots, k = (size_t)i + p.first;
#ifdef WF_SPLIT_RAY_THRU   // (the layout up to cycle f of round 5: the ray and the throughput group as 32-byte records of their own)
    return f < F_O ? k * 8u + (size_t)f
         : f < F_T ? (size_t)F_O * n + k * 8u + (size_t)(f - F_O)
         : f < F_P ? (size_t)F_T * n + k * 8u + (size_t)(f - F_T)
         : f < F_SOA ? (size_t)F_P * n + k * 20u + (size_t)(f - F_P)
#else
    // the ray and the throughput group side by side in ONE 64-byte record: k_wf_begin and the query kernels read both, and a 32-byte record costs a 64-byte fetch
    return f < F_O ? k * 8u + (size_t)f
         : f < F_P ? (size_t)F_O * n + k * 16u + (size_t)(f - F_O)
         : f < F_SOA ? (size_t)F_P * n + k * 20u + (size_t)(f - F_P)
#endif
         : (size_t)f * n + k;
}
TR_DEV float& pf(const WfPool& p, int f, uint32_t i) { return p.data[pidx(p, f, i)]; }
TR_DEV uint32_t& pu(const WfPool& p, int f, uint32_t i) { return reinterpret_cast<uint32_t*>(p.data)[pidx(p, f, i)]; }
// the hit record of slot i as the traversal kernels write it and k_wf_begin / k_wf_advance read it: {t, inst, prim, b1 | b2}; the flags word apart
TR_DEV void st_hit(const WfPool& p, uint32_t i, uint32_t flags, const HitRec& rec) {
    uint32_t* __restrict__ r = reinterpret_cast<uint32_t*>(p.data) + (size_t)(i + p.first) * WF_HIT_WORDS;
    *reinterpret_cast<uint4*>(r) = make_uint4(__float_as_uint(rec.t), rec.inst, rec.prim, __float_as_uint(rec.b1));
    r[4] = __float_as_uint(rec.b2);
    reinterpret_cast<uint32_t*>(p.data)[pidx(p, F_FLAGS, i)] = flags;
}
TR_DEV void ld_hit(const WfPool& p, uint32_t i, HitRec& rec) {
    const uint32_t* __restrict__ r = reinterpret_cast<const uint32_t*>(p.data) + (size_t)(i + p.first) * WF_HIT_WORDS;
    const uint4 a = *reinterpret_cast<const uint4*>(r);
    rec.t = __uint_as_float(a.x); rec.inst = a.y; rec.prim = a.z; rec.b1 = __uint_as_float(a.w); rec.b2 = __uint_as_float(r[4]);
}
TR_DEV f3 ld3(const WfPool& p, int f, uint32_t i) { return mk(pf(p, f, i), pf(p, f + 1, i), pf(p, f + 2, i)); }
TR_DEV void st3(const WfPool& p, int f, uint32_t i, f3 v) { pf(p, f, i) = v.x; pf(p, f + 1, i) = v.y; pf(p, f + 2, i) = v.z; }

TR_DEV void ld_bsdf(const DevScene& sc, const WfPool& p, uint32_t i, Bsdf& b) {
    b.p = ld3(p, F_P, i); b.n = ld3(p, F_N, i); b.tan = ld3(p, F_TAN, i);

    b.mat = sc.materials + pu(p, F_MAT, i);
    b.merl_data = sc.merl_data;
    if (sc.textures) { b.u = pf(p, F_U, i); b.v = pf(p, F_V, i); } else { b.u = 0.0f; b.v = 0.0f; }
}
TR_DEV void st_bsdf(const DevScene& sc, const WfPool& p, uint32_t i, const Bsdf& b) {
    st3(p, F_P, i, b.p); st3(p, F_N, i, b.n); st3(p, F_TAN, i, b.tan);

    pu(p, F_MAT, i) = (uint32_t)(b.mat - sc.materials);
    if (sc.textures) { pf(p, F_U, i) = b.u; pf(p, F_V, i) = b.v; }
}

// ---- stage kernels --------------------------------------------------------------------------

// ---- ray queues: per-stage compaction of the slots that have a ray to trace --------------------
// Producers append with one atomic per wave (ballot + prefix count); slots keep their place in the pool, only their
// indices are compacted, so every stage still reads and writes pool fields at the slot's own address.
// Every queue is cut into WF_SEGS segments with a counter each: appends to ONE counter serialise at 11.4 ns apiece on an MI355X
// (tools/ubench_atomics.hip: 130 000 waves of a round = 1.5 ms per append, which was most of k_wf_advance / k_wf_begin /
// k_wf_regen and the floor under the refills of k_wf_trace_dyn); over 64 counters on separate 128-B lines they cost 0.3 ns.
// The entry of pool slot i lives in segment (i / 256) % WF_SEGS of whatever queue it is in: one-thread-per-slot kernels append
// to segment blockIdx.x % WF_SEGS, one-thread-per-entry kernels read entries (blockIdx.x / WF_SEGS) * 256 ... of segment
// blockIdx.x % WF_SEGS and append to the same segment, so a segment never holds more than the slots of its own chunks (seg_cap).
// Control words of segment s: qctl[s * WF_SEG_STRIDE + k]; k = 0..2 entries in queue A / B / C, 3..5 consumer cursors of the
// traversal kernels, 6 entries in the regeneration queue, 8 + kind entries of the shading queue of a material kind (the material
// sort of k_wf_begin). All zeroed at the start of every round.
#ifndef WF_FUSED_DEFAULT
#define WF_FUSED_DEFAULT 1   // k_wf_sort + k_wf_shade_kind instead of k_wf_begin + k_wf_query_kind (TRAYHIP_WF_FUSED overrides)
#endif
#define WF_SEGS 64u
#define WF_SEG_STRIDE 32u
#define WF_QCTL_WORDS (WF_SEGS * WF_SEG_STRIDE)
#define WF_MAT_KINDS 7     // TRAY_MAT_*
#ifdef TR_HOST_EMU
inline
#else
__host__ __device__ inline
#endif
uint32_t wf_seg_cap(uint32_t n_chunks) { return (n_chunks + WF_SEGS - 1u) / WF_SEGS * TR_BLOCK; }
TR_DEV uint32_t wf_my_seg() { return blockIdx.x & (WF_SEGS - 1u); }
// An entry of a RAY queue (A, B, C) is the ray itself: 8 words = slot, origin, direction, the slot's F_FLAGS word as the producer
// left it (bit 31: a camera ray, min_t = 0). The traversal kernel's refill is then ONE coalesced fetch after its atomic (64 lanes,
// 64 consecutive 32-byte records) instead of entry -> seven fields scattered over the pool, and its result write needs no read
// of the flags. The regeneration queue and the material kinds' shading queues hold slot indices only.
#define WF_RAY_WORDS 8u
enum : uint32_t { WF_CAMERA_RAY = 1u << 31 };
TR_DEV void wf_put_ray(uint32_t* __restrict__ queue, size_t pos, uint32_t slot, f3 o, f3 d, uint32_t flags) {
    uint4* __restrict__ r = reinterpret_cast<uint4*>(queue + pos * WF_RAY_WORDS);
    r[0] = make_uint4(slot, __float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z));
    r[1] = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), flags);
}
TR_DEV void wf_enqueue_ray(const WfPool& pool, uint32_t* __restrict__ queue, uint32_t* __restrict__ qctl, uint32_t k, bool want, uint32_t slot, f3 o, f3 d, uint32_t flags) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
    const uint32_t seg = blockIdx.x & (WF_SEGS - 1u);
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(qctl + seg * WF_SEG_STRIDE + k, (uint32_t)__popcll(m));
    base = __shfl(base, (int)leader);
    if (want) wf_put_ray(queue, (size_t)seg * pool.seg_cap + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), slot, o, d, flags);
}
TR_DEV void wf_enqueue(const WfPool& pool, uint32_t* __restrict__ queue, uint32_t* __restrict__ qctl, uint32_t k, bool want, uint32_t slot) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
    const uint32_t seg = wf_my_seg();
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(qctl + seg * WF_SEG_STRIDE + k, (uint32_t)__popcll(m));
    base = __shfl(base, (int)leader);
    if (want) queue[(size_t)seg * pool.seg_cap + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = slot;
}
// Workgroup-wide append in the order of an 8-valued key (a counting sort in LDS, as the material sort of k_wf_begin: only indices move):
// the workgroup's entries land in ONE contiguous run of its segment, grouped by key. Used for the ray queues with key = octant of the
// ray direction: the traversal's near-child order (bvh.rs:105-119) depends on the direction signs only, so the lanes of a wave that
// draws 64 consecutive entries walk the trees in the same order and touch the same nodes. Every thread of the workgroup calls.
#define WF_SORT_KEYS 8u     // direction octant (adding the origin's octant of the scene box as 3 more key bits measured the same: 82.3 vs 82.2 Msamples/s)
TR_DEV uint32_t wf_octant(f3 d) { return (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u); }
#ifndef WF_NO_OCTANT_SORT
TR_DEV void wf_enqueue_ray_by_key(const WfPool& pool, uint32_t* __restrict__ queue, uint32_t* __restrict__ qctl, uint32_t k, bool want, uint32_t slot, f3 o, f3 d,
                                  uint32_t flags, uint32_t key, uint32_t* s_cnt /* WF_SORT_KEYS */, uint32_t* s_base /* WF_SORT_KEYS */) {
    if (threadIdx.x < WF_SORT_KEYS) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t rank = 0u;
    if (want) rank = atomicAdd(&s_cnt[key & (WF_SORT_KEYS - 1u)], 1u);
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t total = 0u;
        for (uint32_t b = 0; b < WF_SORT_KEYS; ++b) { const uint32_t c = s_cnt[b]; s_base[b] = total; total += c; }
        const uint32_t base = total ? atomicAdd(qctl + wf_my_seg() * WF_SEG_STRIDE + k, total) : 0u;
        for (uint32_t b = 0; b < WF_SORT_KEYS; ++b) s_base[b] += base;
    }
    __syncthreads();
    if (want) wf_put_ray(queue, (size_t)wf_my_seg() * pool.seg_cap + s_base[key & (WF_SORT_KEYS - 1u)] + rank, slot, o, d, flags);
}
#endif
// The shading kernels' append to the NEXT round's stage A queue: grouped by direction octant like k_wf_advance's own appends. Every thread of the workgroup calls.
TR_DEV void wf_append_next_a(const WfPool& pool, uint32_t* __restrict__ queue_a, uint32_t* __restrict__ qctl, bool want, uint32_t slot, f3 o, f3 d, uint32_t flags,
                             uint32_t* s_cnt, uint32_t* s_base) {
#ifndef WF_NO_OCTANT_SORT
    wf_enqueue_ray_by_key(pool, queue_a, qctl, 0u, want, slot, o, d, flags, want ? wf_octant(d) : 0u, s_cnt, s_base);
#else
    (void)s_cnt; (void)s_base;
    wf_enqueue_ray(pool, queue_a, qctl, 0u, want, slot, o, d, flags);
#endif
}
// ... from a workgroup some of whose threads have RETURNED already (the one-thread-per-entry shading kernels: threads without an entry leave before
// the vertex is shaded, and the device functions in between vote over the lanes that are left): the barriers below count the threads that are still
// there (a terminated wave is not waited for, a terminated lane is masked), so nothing may hang on a particular thread -- the counts s_cnt[0 .. 7] and
// the ticket s_cnt[8] were zeroed while every thread was present; whoever draws ticket 0 reserves the workgroup's range.
TR_DEV void wf_append_live(const WfPool& pool, uint32_t* __restrict__ queue, uint32_t* __restrict__ qctl, uint32_t k, bool want, uint32_t slot, f3 o, f3 d, uint32_t flags,
                           uint32_t* s_cnt /* WF_SORT_KEYS + 1, zeroed */, uint32_t* s_base /* WF_SORT_KEYS */);
TR_DEV void wf_append_next_a_live(const WfPool& pool, uint32_t* __restrict__ queue_a, uint32_t* __restrict__ qctl, bool want, uint32_t slot, f3 o, f3 d, uint32_t flags,
                                  uint32_t* s_cnt /* WF_SORT_KEYS + 1, zeroed */, uint32_t* s_base /* WF_SORT_KEYS */) {
    wf_append_live(pool, queue_a, qctl, 0u, want, slot, o, d, flags, s_cnt, s_base);
}
TR_DEV void wf_append_live(const WfPool& pool, uint32_t* __restrict__ queue_a, uint32_t* __restrict__ qctl, uint32_t qk, bool want, uint32_t slot, f3 o, f3 d, uint32_t flags,
                           uint32_t* s_cnt, uint32_t* s_base) {
#ifndef WF_NO_OCTANT_SORT
    const uint32_t key = wf_octant(d);
    uint32_t rank = 0u;
    if (want) rank = atomicAdd(&s_cnt[key], 1u);
    __syncthreads();
    if (atomicAdd(&s_cnt[WF_SORT_KEYS], 1u) == 0u) {
        uint32_t total = 0u;
        for (uint32_t b = 0; b < WF_SORT_KEYS; ++b) { const uint32_t c = s_cnt[b]; s_base[b] = total; total += c; }
        const uint32_t base = total ? atomicAdd(qctl + wf_my_seg() * WF_SEG_STRIDE + qk, total) : 0u;
        for (uint32_t b = 0; b < WF_SORT_KEYS; ++b) s_base[b] += base;
    }
    __syncthreads();
    if (want) wf_put_ray(queue_a, (size_t)wf_my_seg() * pool.seg_cap + s_base[key] + rank, slot, o, d, flags);
#else
    (void)s_cnt; (void)s_base;
    wf_enqueue_ray(pool, queue_a, qctl, qk, want, slot, o, d, flags);
#endif
}
// one thread per queue entry: the entry this thread owns, or false
TR_DEV bool wf_my_entry(const WfPool& pool, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ qctl, uint32_t k, uint32_t& slot) {
    const uint32_t seg = wf_my_seg(), q = (blockIdx.x / WF_SEGS) * TR_BLOCK + threadIdx.x;
    if (q >= qctl[seg * WF_SEG_STRIDE + k]) return false;
    slot = queue[(size_t)seg * pool.seg_cap + q];
    return true;
}
// Traversal kernels: the next segment after `seg` (cyclically; `seg` itself last) whose cursor has not reached its count, or
// WF_SEGS. The wave reads all 64 (count, cursor) pairs with one load each; cursors are read past the non-coherent caches.
TR_DEV uint32_t wf_next_segment(const uint32_t* __restrict__ qctl, uint32_t stage, uint32_t seg, uint32_t& count_out) {
#ifdef TR_HOST_EMU
    if (!hip_emu::block().simt) {   // one-lane waves: the same search, serially
        for (uint32_t k = 1; k <= WF_SEGS; ++k) {
            const uint32_t s2 = (seg + k) & (WF_SEGS - 1u);
            const uint32_t c = qctl[s2 * WF_SEG_STRIDE + stage];
            if (qctl[s2 * WF_SEG_STRIDE + 3u + stage] < c) { count_out = c; return s2; }
        }
        return WF_SEGS;
    }
#endif
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = qctl[lane * WF_SEG_STRIDE + stage];
    const uint32_t u = __hip_atomic_load(qctl + lane * WF_SEG_STRIDE + 3u + stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long open = __ballot(u < c);
    if (open == 0ull) return WF_SEGS;
    const unsigned long long after = seg >= 63u ? 0ull : (open >> (seg + 1u)) << (seg + 1u);
    const uint32_t next = (uint32_t)__ffsll((long long)(after ? after : open)) - 1u;
    count_out = __shfl(c, (int)next);
    return next;
}

// Persistent-threads traversal with dynamic ray fetch: a wave keeps 64 rays in flight; a lane whose ray is finished takes
// the next entry of the stage's queue instead of idling until the slowest ray of its wave is done (the tail is what
// holds the lane utilisation of one-ray-per-lane kernels near 10 % on scenes that mix walls with million-triangle
// meshes). One loop iteration = one step of the one-loop two-level traversal of trace_bvh (same visiting order, same
// arithmetic); the per-lane stack lives in LDS as in every other traversing kernel.
#ifndef WF_REFILL_MIN
#define WF_REFILL_MIN 24
#endif
#ifndef WF_REFILL_MIN_B
#define WF_REFILL_MIN_B WF_REFILL_MIN   // ... of the occlusion stage (its rays end at the first hit: measured apart, profiles/r05_c5_stage_b_thresholds.txt)
#endif
#ifndef WF_TRACE_WAVES
#define WF_TRACE_WAVES 4   // waves per SIMD the traversal kernel is compiled for
#endif
#ifndef WF_NODE_STEPS
#define WF_NODE_STEPS 8    // node steps per round of the while-while loop
#endif
#ifndef WF_NODE_MIN
#define WF_NODE_MIN 16     // ... as long as this many lanes still have node work (or nobody waits for the leaf / pop phase)
#endif
#ifndef WF_NODE_MIN_B
#define WF_NODE_MIN_B WF_NODE_MIN
#endif
// Control words 16 + STAGE of segment 0: rays of the stage handed to k_wf_trace_fallback (below); their slots are listed in `fallback`.
#define WF_FB_WORD 16u
// A ray whose reciprocal direction is finite and nonzero in every component: for those the box of a node is implied by the boxes of its
// children (host/gates.hpp: QuadTrees), which is what lets a record hold the grandchildren. Everything else -- a zero or denormal
// direction component (1 / d = inf: 0 * inf and -0 * inf make the reference's slab test erratic), NaN, inf -- is traced by the
// reference's own binary traversal.
TR_DEV bool wf_regular(f3 inv_dir) {
    const float ax = fabsf(inv_dir.x), ay = fabsf(inv_dir.y), az = fabsf(inv_dir.z);
    return (ax > 0.0f) & (ax < TR_INF) & (ay > 0.0f) & (ay < TR_INF) & (az > 0.0f) & (az < TR_INF);
}
template <int STAGE, int ANIM>
__global__ __launch_bounds__(TR_BLOCK, WF_TRACE_WAVES) void k_wf_trace_dyn(const DevScene scv, WfPool pool, const uint32_t* __restrict__ queue,
                                                           uint32_t* __restrict__ qctl, DevStats* __restrict__ stats, uint32_t lds_depth,
                                                           uint32_t* __restrict__ overflow, uint32_t* __restrict__ fallback, uint32_t fused) {
    const DevScene& sc = scv;
    TR_DYN_LDS(uint32_t, s_stack);   // stack_depth x TR_BLOCK entries
    const LdsU stack = TR_LDS_U(s_stack) + threadIdx.x;   // (its own address space: a pop must not become a flat load that may hit either memory)
    // entries past lds_depth live in a per-thread column of `overflow` (HBM): the LDS part is sized for the occupancy the
    // kernel is compiled for, the rarely reached deep levels of the largest meshes must not cost every workgroup its LDS
    const uint32_t ovf_stride = gridDim.x * TR_BLOCK;
    uint32_t* __restrict__ ovf = overflow + (blockIdx.x * TR_BLOCK + threadIdx.x);
#define WF_PUSH(v) do { const uint32_t v_ = (v); if ((uint32_t)sp < lds_depth) stack[sp * TR_BLOCK] = v_; else ovf[(size_t)((uint32_t)sp - lds_depth) * ovf_stride] = v_; ++sp; } while (0)
#define WF_POP(e) do { --sp; if ((uint32_t)sp < lds_depth) e = stack[sp * TR_BLOCK]; else e = ovf[(size_t)((uint32_t)sp - lds_depth) * ovf_stride]; } while (0)
    // a node entry is two words, the entry distance below the descriptor: one test for both where they lie in LDS (the usual case)
#define WF_PUSH_NODE(t_, d_) do { const uint32_t tw_ = (t_), dw_ = (d_);                                                               \
        if ((uint32_t)sp + 2u <= lds_depth) { stack[sp * TR_BLOCK] = tw_; stack[(sp + 1) * TR_BLOCK] = dw_; sp += 2; }                   \
        else { WF_PUSH(tw_); WF_PUSH(dw_); } } while (0)
#define WF_POP_NODE(d_, t_) do {                                                                                                       \
        if ((uint32_t)sp <= lds_depth) { sp -= 2; d_ = stack[(sp + 1) * TR_BLOCK]; t_ = stack[sp * TR_BLOCK]; }                          \
        else { WF_POP(d_); WF_POP(t_); } } while (0)
    // pops node entries until one passes the box test of the moment (stored entry distance < max_t now) -- the lane goes on with it -- or
    // something else is on top (instance entry, end of a mesh: left for the pop phase) or the stack is empty; cheap (LDS), so it runs
    // wherever a lane runs out of node work instead of sending the lane through the pop phase. (Leaving a mesh here as well -- the
    // world ray back into o / d / inv_dir and the direction signs, inside the node loop -- was measured: C5 stand-in 102 -> 87 Msamples/s,
    // profiles/r03_c5_inline_mesh_exit_ab.txt.)
#define WF_POP_NODES() do {                                                                                                          \
        while (sp > 0) {                                                                                                             \
            uint32_t e_, t_;                                                                                                         \
            WF_POP(e_);                                                                                                              \
            if ((e_ & STK_KIND_MASK) != STK_NODE) { ++sp; break; }                                                                   \
            WF_POP(t_);                                                                                                              \
            if (__uint_as_float(t_) < max_t) { cur = e_; cur_count = nd_count(e_); cur_offset = nd_offset(e_); mode = cur_count != 0u ? TM_LEAF : TM_NODE; break; } \
        }                                                                                                                            \
    } while (0)
    const uint32_t lane = threadIdx.x & 63u;
    const bool any_hit = STAGE == 1;
    bool active = false;
    uint32_t slot = 0u, n_rays = 0u, ray_flags = 0u;
    // the queue segment this wave draws from; it moves on (cyclically, to the next segment that still has entries) when the
    // segment is drained, so the cursor atomics of the 4 x tgrid waves are spread over WF_SEGS counters
    uint32_t seg_cnt = 0u;
    uint32_t seg = wf_next_segment(qctl, STAGE, (blockIdx.x * (TR_BLOCK / 64u) + (threadIdx.x >> 6) + WF_SEGS - 1u) & (WF_SEGS - 1u), seg_cnt);
    bool exhausted = seg == WF_SEGS;
#ifdef WF_TRACE_STATS
    uint32_t c_iter = 0u, c_visit = 0u, c_expand = 0u, c_inst = 0u, c_tri = 0u;
#define WF_COUNT(x) (++(x))
#else
#define WF_COUNT(x) ((void)0)
#endif
#ifdef WF_TRACE_CLOCKS   // where a wave's cycles go: 0 refill, 1 node phase, 2 leaf phase, 3 pop phase, 4 result write
    unsigned long long wclk[5] = {0, 0, 0, 0, 0};
    long long wclk_t = clock64();
#define WF_CLK(slot) do { const long long now_ = clock64(); wclk[slot] += (unsigned long long)(now_ - wclk_t); wclk_t = now_; } while (0)
#else
#define WF_CLK(slot) ((void)0)
#endif
    // traversal state (trace_bvh)
    f3 wo = mk(0, 0, 0), wd = mk(0, 0, 0), d = wd;
    f3 winv = wd;   // 1 / wd, kept for the return from a mesh (three IEEE divisions, ~30 instructions, per mesh left; the kernel has the registers since round 5)
    // origin and reciprocal direction of the ray in the space it is traversing, as the register pairs the node step's packed arithmetic
    // reads them from: (o.x, o.y), (o.z, 1 / d.x), (1 / d.y, 1 / d.z)
    f2 oxy = mk2(0.0f, 0.0f), ozix = oxy, iyz = oxy;
#define WF_SET_RAY(o_, inv_) do { oxy = mk2((o_).x, (o_).y); ozix = mk2((o_).z, (inv_).x); iyz = mk2((inv_).y, (inv_).z); } while (0)
    bool in_mesh = false, any = false;
    // the direction signs of the ray in the space it is traversing, as what the node step needs of them: the byte offset of the plane
    // the ray ENTERS through inside a record's six plane rows (0 = the row of minima, 48 = the row of maxima, per axis; the other one
    // is the exit) and 3 x the direction octant (the shift that finds the octant's visiting order in the record)
    uint32_t sel_x = 0u, sel_y = 0u, sel_z = 0u, oct3 = 0u;
#define WF_SIGNS() do { sel_x = d.x < 0.0f ? 48u : 0u; sel_y = d.y < 0.0f ? 48u : 0u; sel_z = d.z < 0.0f ? 48u : 0u;                       \
                        oct3 = 3u * ((d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u)); } while (0)
    float min_t = 0.0f, max_t = 0.0f;
    int sp = 0;
    uint32_t cur = 0u, cur_inst = 0u, tri_base = 0u, cur_offset = 0u, cur_count = 0u;   // cur: descriptor of the node to expand (dev_geom.h: nd_*)
    enum : uint32_t { TM_NODE = 0u, TM_LEAF = 1u, TM_POP = 2u };
    uint32_t mode = TM_NODE;
    uint32_t tree = sc.top_quad_first;   // entry record of the tree the ray is in (index into sc.quads)
    HitRec rec;
    rec.t = 0.0f; rec.inst = 0xffffffffu; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
    // hands the lane's ray to k_wf_trace_fallback (the result is written there)
    // (`fallback` is the buffer of a ray queue that is idle while this stage runs -- B's during A, C's during B, A's during C --: the ray's
    // own record goes there, in the order of the counter, and k_wf_trace_fallback reads nothing else)
#define WF_DEFER() do { const uint32_t k_ = atomicAdd(qctl + WF_FB_WORD + STAGE, 1u);                                                  \
                        wf_put_ray(fallback, k_, slot, wo, wd, ray_flags | ((STAGE == 0 && min_t == 0.0f) ? WF_CAMERA_RAY : 0u)); } while (0)
    for (;;) {
        bool deferred = false;   // the lane's ray goes to k_wf_trace_fallback (ONE hand-over site at the end of the iteration: two cost the kernel 25 spilled VGPRs)
        // ---- refill idle lanes from the queue
        if (!exhausted) {
            const unsigned long long idle = __ballot(!active);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (n_idle >= (STAGE == 1 ? WF_REFILL_MIN_B : WF_REFILL_MIN) || n_idle == (uint32_t)__popcll(__ballot(1))) {
                const uint32_t leader = (uint32_t)__ffsll((long long)idle) - 1u;
                uint32_t base = 0u;
                if (lane == leader) base = atomicAdd(qctl + seg * WF_SEG_STRIDE + 3u + STAGE, n_idle);
                base = __shfl(base, (int)leader);
                if (!active) {
                    const uint32_t q = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    if (q < seg_cnt) {
                        const uint4* __restrict__ rr = reinterpret_cast<const uint4*>(queue + ((size_t)seg * pool.seg_cap + q) * WF_RAY_WORDS);
                        const uint4 r0 = rr[0], r1 = rr[1];   // the ray record (wf_put_ray)
                        slot = r0.x;
                        wo = mk(__uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));
                        wd = mk(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z));
                        ray_flags = r1.w & ~WF_CAMERA_RAY;
                        if (STAGE == 0) { min_t = (r1.w & WF_CAMERA_RAY) ? 0.0f : 0.001f; max_t = TR_INF; }
                        else { min_t = 0.001f; max_t = STAGE == 1 ? 0.999f : TR_INF; }
                        d = wd;
                        const f3 inv_dir = rcp_rn3(d);
                        winv = inv_dir;
                        WF_SET_RAY(wo, inv_dir);
                        WF_SIGNS();
                        tree = sc.top_quad_first; cur = 0u; sp = 0; in_mesh = false; any = false; mode = TM_NODE;
                        rec.t = 0.0f; rec.inst = 0xffffffffu; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
                        ++n_rays;
                        if (wf_regular(inv_dir)) active = true;
                        else deferred = true;
                    }
                }
                if (base + n_idle >= seg_cnt) {   // this segment is drained (by this refill or by somebody else's)
                    seg = wf_next_segment(qctl, STAGE, seg, seg_cnt);
                    exhausted = seg == WF_SEGS;
                }
            }
        }
        WF_CLK(0);
        if (!__any(active || deferred)) { if (exhausted) break; continue; }
        // ---- traversal, while-while form. A lane is in one of three modes:
        //   TM_NODE  EXPANDS the record `cur` refers to (host/gates.hpp: QuadTrees): up to four (box, descriptor) slots -- the children of
        //            a node, with an interior child replaced by ITS two children -- in one 128-byte fetch. All four boxes are tested with
        //            the ray's current max_t; the slots hit are taken in the order in which the reference would reach them (near child of
        //            the node first, inside a child its near child first: bvh.rs:105-119 with the three split axes the record carries);
        //            the first goes on, the others go on the stack as (descriptor, entry distance). A ray enters a tree at record 0.
        //   TM_LEAF  reached a leaf: triangles of a BVH<Triangle> leaf, or the instances of a BVH<Instance> leaf
        //   TM_POP   needs a stack entry that is not a node (instance entry, leaving a mesh) or has none left
        // The reference tests a node's box when it reaches the node, with the max_t of that moment (bvh.rs:89-127). max_t enters
        // the slab test only through `tmin < max_t` (bbox_hit), everything else in it is the same whenever it is evaluated, so the
        // test at pop time is the stored entry distance against the current max_t: no second fetch of the node. The box of a child
        // that was replaced by its children is never tested: for the rays this kernel traverses (wf_regular) it is hit whenever one
        // of them is. Candidates and their order are exactly the reference's.
        // The node phase repeats while enough lanes have node work, so the (much longer) leaf / pop code runs once per
        // several node steps instead of once per step for whichever few lanes happen to need it.
        bool finished = false;
#pragma nounroll
        for (int it = 0; it < WF_NODE_STEPS; ++it) {
            const bool in_node = active && mode == TM_NODE;
            const uint32_t n_node = (uint32_t)__popcll(__ballot(in_node));
            if (n_node == 0u) break;
            if (it > 0 && n_node < (STAGE == 1 ? WF_NODE_MIN_B : WF_NODE_MIN) && __any(active && mode != TM_NODE)) break;
            if (in_node) {
                WF_COUNT(c_iter);
                // the record's planes are fetched as the ray meets them: per axis the plane it enters through and the one it leaves through
                // (bbox.rs:77-85 picks them by the sign of the direction; here the sign picks the ADDRESS, so nothing is selected afterwards)
                // (all records of a scene lie in one buffer of < 4 GB: a uniform base and 32-bit offsets, one add per row)
                const char* __restrict__ qb = reinterpret_cast<const char*>(sc.quads);
                const uint32_t rb = (tree + nd_offset(cur)) << 7;
                const float4 nxp = *reinterpret_cast<const float4*>(qb + (rb + sel_x)), fxp = *reinterpret_cast<const float4*>(qb + (rb + (sel_x ^ 48u)));
                const float4 nyp = *reinterpret_cast<const float4*>(qb + (rb + sel_y) + 16), fyp = *reinterpret_cast<const float4*>(qb + (rb + (sel_y ^ 48u)) + 16);
                const float4 nzp = *reinterpret_cast<const float4*>(qb + (rb + sel_z) + 32), fzp = *reinterpret_cast<const float4*>(qb + (rb + (sel_z ^ 48u)) + 32);
                const float4 dq = *reinterpret_cast<const float4*>(qb + rb + 96);
                const uint32_t order = *reinterpret_cast<const uint32_t*>(qb + rb + 116);
                // fast_intersect (bbox.rs:75-104) of the four slots, two at a time in packed arithmetic: t = (plane - o) * inv_dir per axis
                // and side, then tmin = the largest entry, tmax = the smallest exit. For the rays this kernel traverses (wf_regular) and boxes
                // with min <= max none of these values is a NaN and the reference's early-out comparisons (tmin > tymax || tymin > tmax,
                // then the same against z) say exactly "some entry lies behind some exit", i.e. max3(entries) > min3(exits); the entry of an
                // axis never lies behind its own exit (monotone rounding). An unused slot's planes are all +inf: tmin = +inf or tmax = -inf.
                // (the ray's origin and reciprocal direction live in three register pairs: oxy, ozix, iyz)
                const f2 nxa = pk_mul_hi(pk_sub_lo(mk2(nxp.x, nxp.y), oxy), ozix), nxb = pk_mul_hi(pk_sub_lo(mk2(nxp.z, nxp.w), oxy), ozix);
                const f2 fxa = pk_mul_hi(pk_sub_lo(mk2(fxp.x, fxp.y), oxy), ozix), fxb = pk_mul_hi(pk_sub_lo(mk2(fxp.z, fxp.w), oxy), ozix);
                const f2 nya = pk_mul_lo(pk_sub_hi(mk2(nyp.x, nyp.y), oxy), iyz), nyb = pk_mul_lo(pk_sub_hi(mk2(nyp.z, nyp.w), oxy), iyz);
                const f2 fya = pk_mul_lo(pk_sub_hi(mk2(fyp.x, fyp.y), oxy), iyz), fyb = pk_mul_lo(pk_sub_hi(mk2(fyp.z, fyp.w), oxy), iyz);
                const f2 nza = pk_mul_hi(pk_sub_lo(mk2(nzp.x, nzp.y), ozix), iyz), nzb = pk_mul_hi(pk_sub_lo(mk2(nzp.z, nzp.w), ozix), iyz);
                const f2 fza = pk_mul_hi(pk_sub_lo(mk2(fzp.x, fzp.y), ozix), iyz), fzb = pk_mul_hi(pk_sub_lo(mk2(fzp.z, fzp.w), ozix), iyz);
                float t0 = fmaxf(fmaxf(nxa.x, nya.x), nza.x), t1 = fmaxf(fmaxf(nxa.y, nya.y), nza.y), t2 = fmaxf(fmaxf(nxb.x, nyb.x), nzb.x), t3 = fmaxf(fmaxf(nxb.y, nyb.y), nzb.y);
                const float x0 = fminf(fminf(fxa.x, fya.x), fza.x), x1 = fminf(fminf(fxa.y, fya.y), fza.y), x2 = fminf(fminf(fxb.x, fyb.x), fzb.x), x3 = fminf(fminf(fxb.y, fyb.y), fzb.y);
                // (a slot that is hit has tmin < max_t <= inf, so inf marks a miss)
                if (!((t0 <= x0) & (t0 < max_t) & (x0 > min_t))) t0 = TR_INF;
                if (!((t1 <= x1) & (t1 < max_t) & (x1 > min_t))) t1 = TR_INF;
                if (!((t2 <= x2) & (t2 < max_t) & (x2 > min_t))) t2 = TR_INF;
                if (!((t3 <= x3) & (t3 < max_t) & (x3 > min_t))) t3 = TR_INF;
                WF_COUNT(c_expand);
                // visiting order: near child first by the sign of the split axis (bvh.rs:105-119), on both levels -- the record holds the three
                // decisions for each of the eight direction octants (host/gates.hpp: bit 0 = second child of the node first, bit 1 / 2 = second
                // slot of the A / B group first); occlusion rays (STAGE 1) visit the child on the LIGHT's side first, i.e. every decision
                // inverted: the boolean does not depend on the order (until a candidate is accepted max_t is the original one, so one is
                // accepted iff a valid candidate exists at all), the rays of a light converge there, and what blocks a light tends to sit
                // near it: +1.4 % on the C5 stand-in.
                const uint32_t ord = (order >> oct3) ^ (STAGE == 1 ? 7u : 0u);
                const bool neg_n = (ord & 1u) != 0u, neg_a = (ord & 2u) != 0u, neg_b = (ord & 4u) != 0u;
                const uint32_t d0 = __float_as_uint(dq.x), d1 = __float_as_uint(dq.y), d2 = __float_as_uint(dq.z), d3 = __float_as_uint(dq.w);
                const float ta0 = neg_a ? t1 : t0, ta1 = neg_a ? t0 : t1, tb0 = neg_b ? t3 : t2, tb1 = neg_b ? t2 : t3;
                const uint32_t da0 = neg_a ? d1 : d0, da1 = neg_a ? d0 : d1, db0 = neg_b ? d3 : d2, db1 = neg_b ? d2 : d3;
                const float v0 = neg_n ? tb0 : ta0, v1 = neg_n ? tb1 : ta1, v2 = neg_n ? ta0 : tb0, v3 = neg_n ? ta1 : tb1;
                const uint32_t e0 = neg_n ? db0 : da0, e1 = neg_n ? db1 : da1, e2 = neg_n ? da0 : db0, e3 = neg_n ? da1 : db1;
                const bool g0 = v0 < TR_INF, g1 = v1 < TR_INF, g2 = v2 < TR_INF, g3 = v3 < TR_INF;
                const bool b1_ = g0, b2_ = g0 | g1, b3_ = g0 | g1 | g2;   // something earlier in the order is hit
                if (g3 & b3_) WF_PUSH_NODE(__float_as_uint(v3), e3);
                if (g2 & b2_) WF_PUSH_NODE(__float_as_uint(v2), e2);
                if (g1 & b1_) WF_PUSH_NODE(__float_as_uint(v1), e1);
                if (b3_ | g3) {
                    cur = g0 ? e0 : (g1 ? e1 : (g2 ? e2 : e3));
                    cur_count = nd_count(cur);
                    if (cur_count != 0u) { cur_offset = nd_offset(cur); mode = TM_LEAF; }
                } else {
                    mode = TM_POP;
                    WF_POP_NODES();
                }
            }
        }
        WF_CLK(1);
        if (active && mode == TM_LEAF) {
            if (in_mesh) {   // BVH<Triangle> leaf (<= 16 triangles), tested in order
                for (uint32_t k = 0; k < cur_count; ++k) {
                    float t, bb1, bb2;
                    WF_COUNT(c_tri);
                    if (triangle_test(sc.tri_verts + (tri_base + cur_offset + k), mk(oxy.x, oxy.y, ozix.x), d, min_t, max_t, t, bb1, bb2)) {
                        max_t = t;
                        rec.t = t; rec.inst = cur_inst; rec.prim = tri_base + cur_offset + k; rec.b1 = bb1; rec.b2 = bb2;
                        any = true;
                        if (any_hit) { finished = true; break; }
                    }
                }
            } else {   // BVH<Instance> leaf (<= 4 instances): queue them so that pops come in leaf order
                for (uint32_t k = cur_count; k > 0u; --k) {
                    WF_PUSH(STK_INSTANCE | (cur_offset + k - 1u));
                }
            }
            mode = TM_POP;
            if (!finished) WF_POP_NODES();
        }
        WF_CLK(2);
        if (active && mode == TM_POP && !finished) {
            bool have_node = false;
            while (sp > 0) {
                uint32_t e;
                WF_POP(e);
                uint32_t kind = e & STK_KIND_MASK;
                if (kind == STK_NODE) {   // the box test the reference runs when it reaches the node: the stored entry distance against max_t now
                    uint32_t tb_;
                    WF_POP(tb_);
                    if (!(__uint_as_float(tb_) < max_t)) continue;
                    cur = e; cur_count = nd_count(e); cur_offset = nd_offset(e);
                    have_node = true;
                    break;
                }
                if (kind == STK_EXIT_MESH) {   // back to world space and the top-level tree
                    in_mesh = false;
                    tree = sc.top_quad_first;
                    d = wd;
                    WF_SET_RAY(wo, winv);
                    WF_SIGNS();
                    continue;
                }
                // Instance::intersect (receiver.rs:29-35): world ray -> object ray by `inv`, direction not renormalised. The entry's
                // 64-byte record (host/gates.hpp: WfInst) holds all of it: one fetch, no dependent second one for static instances
                const float4* __restrict__ wr = reinterpret_cast<const float4*>(sc.wf_insts + (e & ~STK_KIND_MASK));
                const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
#ifndef TR_HOST_EMU
                // (all four words of the last quarter are wanted NOW: hipcc otherwise fetches the flags alone, tests the point-emitter bit and
                // only then fetches the other three words -- a second dependent round trip for every instance entry)
                asm volatile("" :: "v"(w3.y), "v"(w3.z), "v"(w3.w));
#endif
                const uint32_t wflags = __float_as_uint(w3.x);
                if (wflags & tray::WI_POINT) continue;   // emitter.rs:120
                const uint32_t i = __float_as_uint(w3.y);
                WF_COUNT(c_inst);
                f3 lo_, ld;
                if (ANIM && (wflags & tray::WI_ANIMATED)) {   // the path's transform of a moving instance, from the per-slot cache
                    float x[TR_XF_WORDS];
                    // (table mode: the path's time index is fetched here, for the few entries that need it -- one more dependent load for them)
                    instance_inv_cached(sc, wflags >> 8, sc.xf_table ? pu(pool, F_KIDX, slot) : slot, x);
                    lo_ = xf_point_affine_w(x + 12, x[24], wo);
                    ld = xf_vector(x + 12, wd);
                } else if (wflags & tray::WI_AFFINE) {
                    const float m[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
                    lo_ = xf_point_affine(m, wo);   // (row 3 = 0 0 0 1: xf_point's w is exactly one)
                    ld = xf_vector(m, wd);
                } else {
                    const TrayInstance* __restrict__ in = sc.instances + i;
                    lo_ = xf_point(in->inv, wo);
                    ld = xf_vector(in->inv, wd);
                }
                const uint32_t gt = wflags & 7u;
                if (gt == TRAY_GEOM_MESH) {
                    const f3 inv_obj = rcp_rn3(ld);
                    if (!wf_regular(inv_obj)) { deferred = true; break; }   // (the whole ray: the reference's traversal decides, as at the refill)
                    WF_PUSH(STK_EXIT_MESH);
                    in_mesh = true;
                    cur_inst = i;
                    tree = __float_as_uint(w3.z);
                    tri_base = __float_as_uint(w3.w);
                    d = ld;
                    WF_SET_RAY(lo_, inv_obj);
                    WF_SIGNS();
                    cur = 0u; cur_count = 0u;
                    have_node = true;
                    break;
                }
                float t;
                bool hit;
                if (gt == TRAY_GEOM_RECT) hit = rect_test(w3.z, w3.w, lo_, ld, min_t, max_t, t);
                else if (gt == TRAY_GEOM_SPHERE) hit = sphere_test(w3.z, lo_, ld, min_t, max_t, t);
                else hit = disk_test(w3.z, w3.w, lo_, ld, min_t, max_t, t);
                if (hit) {
                    max_t = t;
                    rec.t = t; rec.inst = i; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
                    any = true;
                    if (any_hit) { finished = true; break; }
                }
            }
            if (have_node) mode = cur_count != 0u ? TM_LEAF : TM_NODE;
            else finished = true;
        }
        if (deferred) { WF_DEFER(); finished = false; active = false; }
        WF_CLK(3);
        if (finished) {   // write the result to the ray's own slot
            uint32_t flags = ray_flags;   // (the slot's F_FLAGS, brought by the ray: nobody else touches the slot while its ray is traced)
            if (STAGE == 1) {
                flags = any ? (flags | WF_OCCLUDED) : (flags & ~WF_OCCLUDED);
            } else {
                const uint32_t bit = (STAGE == 0 && !(flags & WF_CFLIGHT)) ? WF_HIT_A : WF_HIT_C;   // (a stage C ray that travels with the A rays: WF_FOLD_C)
                flags = any ? (flags | bit) : (flags & ~bit);
            }
            if (STAGE != 1 && any) st_hit(pool, slot, flags, rec);   // one 20-byte piece of the slot's hit record + the flags word
            else if (STAGE == 1 && fused) pu(pool, F_PEND_OCC, slot) = any ? 1u : 0u;   // (fused shading: the flags word has moved on -- the vertex is closed, its next ray queued)
            else pu(pool, F_FLAGS, slot) = flags;
            active = false;
        }
        WF_CLK(4);
    }
#ifdef WF_TRACE_CLOCKS
    if (lane == 0u) for (int k = 0; k < 5; ++k) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].trav[STAGE * 6 + k], wclk[k]);
#endif
    // one counter update per wave
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if (lane == 0u && n_rays) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].rays, (unsigned long long)n_rays);
#ifdef WF_TRACE_STATS
    uint32_t cs[6] = {c_iter, c_visit, c_expand, c_inst, c_tri, 0u};
    for (int k = 0; k < 5; ++k) {
        uint32_t v = cs[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0u && v) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].trav[STAGE * 6 + k], (unsigned long long)v);
    }
    if (lane == 0u && n_rays) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].trav[STAGE * 6 + 5], (unsigned long long)n_rays);
#endif
#undef WF_COUNT
#undef WF_CLK
#undef WF_PUSH
#undef WF_POP
#undef WF_PUSH_NODE
#undef WF_POP_NODE
#undef WF_POP_NODES
#undef WF_DEFER
#undef WF_SIGNS
#undef WF_SET_RAY
}

// The rays k_wf_trace_dyn<STAGE> did not traverse (direction components that are zero, denormal or not finite; about one ray in 1e7 on
// the bundled scenes): one thread per entry of `fallback`, the reference's own traversal over the binary trees (trace_bvh), the result
// written exactly as k_wf_trace_dyn writes it. The ray is the record k_wf_trace_dyn copied there -- origin, direction, slot, flags, camera
// bit -- whatever the pool holds (round 4 rebuilt it from pool fields its producers happened to store: ADVICE round 4).
template <int STAGE, int ANIM>
__global__ __launch_bounds__(TR_BLOCK) void k_wf_trace_fallback(const DevScene scv, WfPool pool, const uint32_t* __restrict__ qctl, const uint32_t* __restrict__ fallback, uint32_t fused) {
    const DevScene& sc = scv;
    TR_DYN_LDS(uint32_t, s_stack);
    const uint32_t n = qctl[WF_FB_WORD + STAGE];
    for (uint32_t k = blockIdx.x * TR_BLOCK + threadIdx.x; k < n; k += gridDim.x * TR_BLOCK) {
        const uint4* __restrict__ rr = reinterpret_cast<const uint4*>(fallback + (size_t)k * WF_RAY_WORDS);
        const uint4 r0 = rr[0], r1 = rr[1];   // the deferred ray's own record (wf_put_ray), flags as its producer left them
        const uint32_t slot = r0.x;
        uint32_t flags = r1.w & ~WF_CAMERA_RAY;
        Ray ray;
        ray.o = mk(__uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));
        ray.d = mk(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z));
        if (STAGE == 0) { ray.min_t = (r1.w & WF_CAMERA_RAY) ? 0.0f : 0.001f; ray.max_t = TR_INF; }
        else { ray.min_t = 0.001f; ray.max_t = STAGE == 1 ? 0.999f : TR_INF; }
        ray.time = ANIM ? pf(pool, F_TIME, slot) : 0.0f;
        ray.col = (ANIM && sc.xf_table) ? pu(pool, F_KIDX, slot) : slot;
        HitRec rec;
        rec.t = 0.0f; rec.inst = 0xffffffffu; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
        const bool any = trace_bvh<ANIM>(sc, s_stack + threadIdx.x, ray, STAGE == 1, rec);
        if (STAGE == 1) {
            flags = any ? (flags | WF_OCCLUDED) : (flags & ~WF_OCCLUDED);
        } else {
            const uint32_t bit = (STAGE == 0 && !(flags & WF_CFLIGHT)) ? WF_HIT_A : WF_HIT_C;
            flags = any ? (flags | bit) : (flags & ~bit);
        }
        if (STAGE != 1 && any) st_hit(pool, slot, flags, rec);
        else if (STAGE == 1 && fused) pu(pool, F_PEND_OCC, slot) = any ? 1u : 0u;
        else pu(pool, F_FLAGS, slot) = flags;
    }
}

// ---- Ray binning before a traversal stage (round 6) ------------------------------------------------------------------------------------
// The producers append a chunk's rays as one run of its segment (grouped by direction octant): a wave of k_wf_trace_dyn that refills from 64
// consecutive entries gets the rays of ONE chunk -- after the first bounce they start anywhere in the scene, so its lanes walk unrelated parts
// of the trees (lanes 0.36, half the wave cycles waiting, every lane's fetch a cache line of its own). Between the producers and the traversal
// every SEGMENT of the queue is therefore counting-sorted by (cell of the ray's origin in the scene's box, direction octant) into the queue
// buffer that is idle during the stage: the same entries, the same segment, the same count, another order -- so the traversal kernel and its
// control words are untouched, and the 64 rays of a refill start in the same part of BVH<Instance> with the same near-child order
// (bvh.rs:105-119). Every number a ray produces is a function of the ray (TRAY-CBRNG keys on pixel and sample, never on the schedule), so the
// per-sample results stay bit-identical; only the order of the film's f32 sums moves, as with any change of schedule.
//   k_wf_bin_hist     one workgroup per WF_BIN_EPB entries of a segment: histogram of the keys in LDS, then one atomic per bin present
//                     into the segment's global histogram
//   k_wf_bin_scatter  the same workgroup shape: exclusive scan of the segment's histogram (LDS), the workgroup's own counts reserve a range
//                     per bin with one atomic each, every entry is copied to base[bin] + reserved + its rank inside the workgroup
// Cost: the queue is read twice and written once (96 B per ray against the ~350 B a stage A ray moves through HBM).
// MEASURED (profiles/r06_c5_ray_binning_ab.txt, C5 stand-in at full detail, 128 spp): frame 64 228.2 -> 215.7 Msamples/s, frame 127 163.3 -> 154.6
// with both stages binned, the same with 8 cells per axis, stage A alone no better: the passes cost their ~0.9 ms per round and the traversal
// gains nothing -- below the first levels of the trees (cache-resident for every wave anyway) the diffuse rays of one cell fan out like any
// others. OFF by default (WF_BIN_DEFAULT); TRAYHIP_WF_BIN=1|2|3 runs it, tests/test_device_emulation.py keeps it correct.
#ifndef WF_BIN_CELL_BITS
#define WF_BIN_CELL_BITS 2   // cells per axis = 2^bits over the box of BVH<Instance>
#endif
#define WF_BINS (8u << (3 * WF_BIN_CELL_BITS))
#define WF_BIN_EPB 4096u     // queue entries per workgroup of the two passes
#ifndef WF_BIN_DEFAULT
#define WF_BIN_DEFAULT 0u    // stages whose rays are binned: bit 0 = A (camera / continuation rays), bit 1 = B (occlusion rays); TRAYHIP_WF_BIN overrides. Measured: -5 ... -6 % on the C5 stand-in (profiles/r06_c5_ray_binning_ab.txt), so off
#endif
struct WfBinGrid { float lo[3], scale[3]; };   // cell = clamp((o - lo) * scale) per axis; scale = cells / extent (0 for a flat or unbounded axis)
// (host) the grid over a box -- the root of the frame's BVH<Instance>
inline WfBinGrid wf_bin_grid(const float* bmin, const float* bmax) {
    WfBinGrid g{};
    for (int a = 0; a < 3; ++a) {
        const float ext = bmax[a] - bmin[a];
        const bool usable = bmin[a] - bmin[a] == 0.0f && ext - ext == 0.0f && ext > 0.0f;   // finite and not flat (a flat or unbounded axis is one cell)
        g.lo[a] = usable ? bmin[a] : 0.0f;
        g.scale[a] = usable ? (float)(1u << WF_BIN_CELL_BITS) / ext : 0.0f;
    }
    return g;
}
TR_DEV uint32_t wf_bin_key(const WfBinGrid& g, f3 o, f3 d) {
    const float m = (float)((1u << WF_BIN_CELL_BITS) - 1u);
    // (fmaxf / fminf return the other operand for a NaN: an origin that is not a number lands in cell 0; the key only orders, it never decides)
    const uint32_t cx = (uint32_t)fminf(fmaxf((o.x - g.lo[0]) * g.scale[0], 0.0f), m);
    const uint32_t cy = (uint32_t)fminf(fmaxf((o.y - g.lo[1]) * g.scale[1], 0.0f), m);
    const uint32_t cz = (uint32_t)fminf(fmaxf((o.z - g.lo[2]) * g.scale[2], 0.0f), m);
    uint32_t cell = 0u;   // Morton order: neighbouring bins are neighbouring cells
#pragma unroll
    for (uint32_t b = 0; b < WF_BIN_CELL_BITS; ++b) cell |= (((cx >> b) & 1u) << (3u * b)) | (((cy >> b) & 1u) << (3u * b + 1u)) | (((cz >> b) & 1u) << (3u * b + 2u));
    return (cell << 3) | wf_octant(d);
}
TR_DEV uint32_t wf_bin_entry_key(const WfBinGrid& g, const uint32_t* __restrict__ queue, size_t pos) {
    const uint4* __restrict__ rr = reinterpret_cast<const uint4*>(queue + pos * WF_RAY_WORDS);
    const uint4 r0 = rr[0], r1 = rr[1];
    return wf_bin_key(g, mk(__uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w)), mk(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z)));
}
// bin_ctl: [WF_SEGS][WF_BINS] histogram, then [WF_SEGS][WF_BINS] cursors; zeroed by the host before the round
template <int STAGE>
__global__ __launch_bounds__(TR_BLOCK) void k_wf_bin_hist(WfPool pool, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ qctl,
                                                          uint32_t* __restrict__ bin_ctl, WfBinGrid grid) {
    __shared__ uint32_t s_h[WF_BINS];
    const uint32_t seg = wf_my_seg(), e0 = blockIdx.x / WF_SEGS * WF_BIN_EPB;
    const uint32_t cnt = qctl[seg * WF_SEG_STRIDE + STAGE];
    if (e0 >= cnt) return;   // (the whole workgroup)
    for (uint32_t b = threadIdx.x; b < WF_BINS; b += TR_BLOCK) s_h[b] = 0u;
    __syncthreads();
    const uint32_t e1 = min(cnt, e0 + WF_BIN_EPB);
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += TR_BLOCK) atomicAdd(&s_h[wf_bin_entry_key(grid, queue, (size_t)seg * pool.seg_cap + e)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < WF_BINS; b += TR_BLOCK) { const uint32_t c = s_h[b]; if (c) atomicAdd(bin_ctl + seg * WF_BINS + b, c); }
}
template <int STAGE>
__global__ __launch_bounds__(TR_BLOCK) void k_wf_bin_scatter(WfPool pool, const uint32_t* __restrict__ queue, uint32_t* __restrict__ sorted,
                                                             const uint32_t* __restrict__ qctl, uint32_t* __restrict__ bin_ctl, WfBinGrid grid) {
    __shared__ uint32_t s_base[WF_BINS], s_cnt[WF_BINS], s_part[TR_BLOCK];
    const uint32_t seg = wf_my_seg(), e0 = blockIdx.x / WF_SEGS * WF_BIN_EPB;
    const uint32_t cnt = qctl[seg * WF_SEG_STRIDE + STAGE];
    if (e0 >= cnt) return;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t PER = WF_BINS / TR_BLOCK;   // bins per thread of the scan
    static_assert(WF_BINS % TR_BLOCK == 0 && PER >= 1, "the scan deals the bins out evenly");
    // exclusive scan of the segment's histogram: per-thread runs, a Hillis-Steele scan of the 256 run totals, then the runs again
    const uint32_t* __restrict__ hist = bin_ctl + seg * WF_BINS;
    uint32_t run = 0u;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) run += hist[tid * PER + k];
    s_part[tid] = run;
    for (uint32_t b = tid; b < WF_BINS; b += TR_BLOCK) s_cnt[b] = 0u;
    __syncthreads();
    for (uint32_t off = 1u; off < TR_BLOCK; off <<= 1) {
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t acc = s_part[tid] - run;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { s_base[tid * PER + k] = acc; acc += hist[tid * PER + k]; }
    // the workgroup's entries: key and rank inside the workgroup (the order inside a bin is whatever the LDS atomics make it: any order serves)
    const uint32_t e1 = min(cnt, e0 + WF_BIN_EPB);
    constexpr uint32_t ROUNDS = WF_BIN_EPB / TR_BLOCK;
    uint32_t key_rank[ROUNDS];
#pragma unroll
    for (uint32_t k = 0; k < ROUNDS; ++k) {
        const uint32_t e = e0 + k * TR_BLOCK + tid;
        key_rank[k] = 0xffffffffu;
        if (e < e1) {
            const uint32_t key = wf_bin_entry_key(grid, queue, (size_t)seg * pool.seg_cap + e);
            key_rank[k] = (key << 16) | atomicAdd(&s_cnt[key], 1u);   // (rank < WF_BIN_EPB <= 65536, key < WF_BINS <= 65535)
        }
    }
    static_assert(WF_BIN_EPB <= 65536u && WF_BINS < 65535u, "key and rank share a word");
    __syncthreads();
    // one atomic per bin present: where this workgroup's entries of the bin go inside the bin's range
    uint32_t* __restrict__ cursor = bin_ctl + WF_SEGS * WF_BINS + seg * WF_BINS;
    for (uint32_t b = tid; b < WF_BINS; b += TR_BLOCK) { const uint32_t c = s_cnt[b]; if (c) s_base[b] += atomicAdd(cursor + b, c); }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < ROUNDS; ++k) {
        if (key_rank[k] == 0xffffffffu) continue;
        const uint32_t e = e0 + k * TR_BLOCK + tid;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(queue + ((size_t)seg * pool.seg_cap + e) * WF_RAY_WORDS);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(sorted + ((size_t)seg * pool.seg_cap + s_base[key_rank[k] >> 16] + (key_rank[k] & 0xffffu)) * WF_RAY_WORDS);
        const uint4 r0 = src[0], r1 = src[1];
        dst[0] = r0; dst[1] = r1;
    }
}

// Stage A shading: vertex_begin for the slots whose ray hit, end of the sample for those that missed.
// Material sort (north_star: "material sort in LDS"): when kind_queues is given, the vertices this workgroup just set up are
// counted per material kind in LDS (one ds_add_rtn per vertex gives its rank inside the workgroup's share), the workgroup
// reserves a range in each kind's global queue with ONE atomic per kind present, and the slot INDICES are written there -- a
// counting sort of 256 jobs, nothing but indices moves. k_wf_query_kind<kind> then shades each queue with full waves of one
// material kind and only that kind's lobe code compiled in, instead of one thread per pool slot running every kind's code.
#ifndef WF_SHADE_WAVES
#define WF_SHADE_WAVES 4   // waves per SIMD the shading stage kernels (k_wf_begin, k_wf_query_kind) are compiled for: they wait on HBM round trips
#endif                     // for two thirds of their wave cycles, so occupancy is what hides them (round 5: query_kind<matte> had drifted to 131 VGPRs = 3 waves)
template <int ANIM>
__global__ __launch_bounds__(TR_BLOCK, WF_SHADE_WAVES) void k_wf_begin(const DevScene scv, WfPool pool, uint32_t n_active, DevStats* __restrict__ stats,
                                                       uint32_t* __restrict__ queue_b, uint32_t* __restrict__ qctl, uint32_t* __restrict__ kind_queues) {
    __shared__ uint32_t s_cnt[8], s_base[8];
    __shared__ uint32_t s_oct_cnt[WF_SORT_KEYS], s_oct_base[WF_SORT_KEYS];
    // the scene's permutation pool in LDS, as in the tile kernel: a shuffled index is two dependent byte reads (dev_integrator.h: lane_perm_entry)
    __shared__ uint4 s_perm[TR_PERM_BYTES / 16];
    s_perm[threadIdx.x] = reinterpret_cast<const uint4*>(scv.perm_pool)[threadIdx.x];   // TR_PERM_BYTES / 16 == TR_BLOCK
    __syncthreads();
    uint32_t b_oct = 0u;   // direction octant of the slot's occlusion ray
    f3 b_o = mk(0.0f, 0.0f, 0.0f), b_d = b_o;   // ... and the ray
    const DevScene& sc = scv;
    const uint32_t tid = threadIdx.x;
    const uint32_t i = blockIdx.x * TR_BLOCK + tid;
    if (kind_queues) {
        if (tid < 8u) s_cnt[tid] = 0u;
        __syncthreads();
    }
    uint32_t kind = WF_MAT_KINDS;   // no vertex from this slot
    bool dark = false;              // the slot's occlusion ray cannot matter: counted, not queued
    bool counted = false;
    uint32_t flags = i < n_active ? pu(pool, F_FLAGS, i) : 0u;
    // Round 6: the slot's hit, ray and throughput records are fetched TOGETHER with its flags word, not after it was looked at -- nearly every slot of a
    // round holds a vertex, and the kernel is a chain of dependent round trips (flags -> records -> instance / triangle -> material: three quarters of its
    // wave cycles wait); this takes one link out of the chain. (A slot past n_active reads slot 0's records and ignores them.)
    const uint32_t ip = i < n_active ? i : 0u;
    HitRec rec;
    ld_hit(pool, ip, rec);
    const uint32_t p_bounce = pu(pool, F_BOUNCE, ip), p_ks = pu(pool, F_KS, ip);
    const f3 p_o = ld3(pool, F_O, ip), p_d = ld3(pool, F_D, ip), p_t = ld3(pool, F_T, ip), p_illum = ld3(pool, F_ILLUM, ip);
    const float p_time = ANIM ? pf(pool, F_TIME, ip) : 0.0f;
    const uint32_t p_kidx = (ANIM && sc.xf_table) ? pu(pool, F_KIDX, ip) : ip;
    if (flags & WF_INVERTEX) {   // (WF_FOLD_C: the vertex is waiting for its stage C ray, which this round's traversal A carried: k_wf_advance closes it)
        flags = 0u;
    } else if ((flags & LF_ALIVE) && !(flags & WF_HIT_A)) {   // camera miss: black sample; continuation miss: path ends (path.rs:112-115)
        pu(pool, F_FLAGS, i) = (flags & ~(LF_ALIVE | WF_INVERTEX)) | WF_FINISHED;
    } else if (flags & LF_ALIVE) {
        Lane ln;
        ln.perm_lds = TR_LDS_B(s_perm);
        ln.flags = flags;
        ln.bounce = p_bounce; ln.ks = p_ks;
        LN_O(ln) = p_o; ln.d = p_d;
        ln.throughput = p_t; ln.illum = p_illum;
        const f3 illum_in = ln.illum;
        // (hit.dg.ng of the camera ray's hit, quirk Q1: read by vertex_begin only on a specular chain -- at bounce 0 it is what vertex_begin writes)
        ln.first_ng = (ln.bounce != 0u && (flags & LF_SPECULAR)) ? ld3(pool, F_NG, i) : mk(0.0f, 0.0f, 0.0f);
        Counters cnt;
        cnt.rays = 0; cnt.vertices = 0;
        ln.time = p_time; ln.col = p_kidx;
        vertex_begin<ANIM>(sc, ln, rec, cnt);
        counted = true;
        // The occlusion ray of a light sample only matters if BSDF::eval of the light direction is not black (mod.rs:127-131 test the
        // occlusion first, then f). Where the surface has no transmission lobe and w_o, w_i do not lie strictly on the same side of
        // the shading normal, eval removes every reflection lobe (w_o.z * w_i.z > 0 fails, bsdf.rs:71-75; the normalisation of the
        // shading-space vectors divides by a positive length, it cannot make signs equal) and returns black whatever the ray
        // finds: such a ray is counted like the reference's and not traced, and the vertex goes on as if it were occluded. In the
        // queue-compacted schedule that removes rays outright (in the tile kernel the wave would trace for its other lanes anyway).
        if (queue_b && (ln.flags & LF_SHADOW) && !ln.bsdf.mat->textured) {
            const DevMaterial* __restrict__ m = ln.bsdf.mat;
            uint32_t types = 0u;
            constexpr uint32_t kMaxLobes = sizeof(m->lobe) / sizeof(m->lobe[0]);   // every lobe a material can have
            for (uint32_t l = 0; l < m->n_lobes && l < kMaxLobes; ++l) types |= m->lobe[l].type;
            const float zo = -dot(ln.d, ln.bsdf.n), zi = dot(ln.wi_l, ln.bsdf.n);
            const bool same_side = (zo > 0.0f && zi > 0.0f) || (zo < 0.0f && zi < 0.0f);
            if (!(types & BX_TRANSMISSION) && !same_side) { ln.flags &= ~LF_SHADOW; dark = true; }
        }
        if (!(ln.flags & LF_ALIVE)) {   // NormalsDebug: the sample ended at its first hit
            pu(pool, F_FLAGS, i) = (ln.flags & ~WF_INVERTEX) | WF_FINISHED;
            st3(pool, F_ILLUM, i, ln.illum);
            flags = 0u;
        } else {
        pu(pool, F_FLAGS, i) = ln.flags | WF_INVERTEX;
        // (illum changes here only where an emitter was hit by a camera or specular ray, path.rs:72-75; `direct` = 0, `t_vertex` = throughput
        // and w_o = -d are not stored: the query kernels have what they are made of)
        if (ln.illum.x != illum_in.x || ln.illum.y != illum_in.y || ln.illum.z != illum_in.z) st3(pool, F_ILLUM, i, ln.illum);
        if (ln.bounce == 0u) st3(pool, F_NG, i, ln.first_ng);
        st_bsdf(sc, pool, i, ln.bsdf);
        pu(pool, F_LINST, i) = ln.light_inst;
        st3(pool, F_LI, i, ln.li); st3(pool, F_WL, i, ln.wi_l); pf(pool, F_PDFL, i) = ln.pdf_l;
        if (ln.flags & LF_SHADOW) { b_oct = wf_octant(ln.aux_d); b_o = ln.bsdf.p; b_d = ln.aux_d; }
        flags = ln.flags;
        kind = ln.bsdf.mat->mat_kind;
        }
    } else flags = 0u;
    {   // one counter update per wave (all lanes of the wave are here: nobody has returned)
        const unsigned long long m = __ballot(counted), dm = __ballot(dark);
        if (m != 0ull && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) {
            atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].vertices, (unsigned long long)__popcll(m));
            if (dm != 0ull) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].rays, (unsigned long long)__popcll(dm));
        }
    }
#ifndef WF_NO_OCTANT_SORT
    if (queue_b) wf_enqueue_ray_by_key(pool, queue_b, qctl, 1u, kind < WF_MAT_KINDS && (flags & LF_SHADOW) != 0u, i, b_o, b_d, flags | WF_INVERTEX, b_oct, s_oct_cnt, s_oct_base);
#else
    if (queue_b) wf_enqueue_ray(pool, queue_b, qctl, 1u, kind < WF_MAT_KINDS && (flags & LF_SHADOW) != 0u, i, b_o, b_d, flags | WF_INVERTEX);
#endif
    if (kind_queues) {
        uint32_t rank = 0u;
        if (kind < WF_MAT_KINDS) rank = atomicAdd(&s_cnt[kind], 1u);
        __syncthreads();
        if (tid < WF_MAT_KINDS && s_cnt[tid]) s_base[tid] = atomicAdd(qctl + wf_my_seg() * WF_SEG_STRIDE + 8u + tid, s_cnt[tid]);
        __syncthreads();
        if (kind < WF_MAT_KINDS) kind_queues[((size_t)kind * WF_SEGS + wf_my_seg()) * pool.seg_cap + s_base[kind] + rank] = i;
    }
}

// ---- Fused shading (round 6): k_wf_sort + k_wf_shade_kind replace k_wf_begin + k_wf_query_kind ----------------------------------------------
// k_wf_begin set a vertex up, wrote its 80-byte record and queued the occlusion ray; after the traversal k_wf_query_kind read the record back (and the ray /
// throughput record again) for the BSDF queries -- because ONE of the three queries, the light half of estimate_direct, is only added if the occlusion ray
// finds nothing (mod.rs:127-131). Nothing else in the vertex depends on that ray: the BSDF half, the continuation and Russian roulette do not, and the light
// half's VALUE does not either -- only whether it counts. So the vertex is shaded in one go, in registers: vertex_begin, the light term computed as if
// unoccluded, the other queries, the end-of-vertex bookkeeping with the light term left out; the term's factors (throughput at the vertex, f * li * |cos| * w /
// pdf) go to the slot's pending words, the occlusion ray is queued, the traversal writes its verdict into F_PEND_OCC, and whoever touches the slot's radiance
// next -- this kernel at the path's next vertex, k_wf_advance for a finished sample or a vertex with a stage C ray -- performs the reference's own
// `illum += throughput * direct` (path.rs:82) with direct = the term or zero. Same operations on the same values in the same order per sample: bit-identical.
// A light term that is black needs no ray at all (the reference traces it and adds nothing): counted, not traced -- round 2's "dark" rays are the special case.
// The material sort moves in front: k_wf_sort reads a slot's flags and the instance of its hit, ends the samples whose ray missed, and sorts the others by
// material kind into the shading queues (counting sort in LDS, as k_wf_begin did).
// the slot's pending light term, applied (flags & WF_PENDING; the caller clears the bit)
TR_DEV f3 wf_apply_pending(const WfPool& pool, uint32_t i, f3 illum) {
    const f3 tv = ld3(pool, F_PEND_TV, i), dl = ld3(pool, F_PEND_DL, i);
    const f3 direct = pu(pool, F_PEND_OCC, i) != 0u ? mk(0.0f, 0.0f, 0.0f) : dl;
    return illum + tv * direct;   // path.rs:82, as vertex_end writes it
}
template <int UNUSED>   // (a template so that it is instantiated in ONE translation unit: kernel_list.h)
__global__ __launch_bounds__(TR_BLOCK) void k_wf_sort(const DevScene scv, WfPool pool, uint32_t n_active, uint32_t* __restrict__ qctl, uint32_t* __restrict__ kind_queues) {
    __shared__ uint32_t s_cnt[8], s_base[8];
    const DevScene& sc = scv;
    const uint32_t tid = threadIdx.x, i = blockIdx.x * TR_BLOCK + tid;
    if (tid < 8u) s_cnt[tid] = 0u;
    __syncthreads();
    uint32_t kind = WF_MAT_KINDS;
    const uint32_t flags = i < n_active ? pu(pool, F_FLAGS, i) : 0u;
    const uint32_t inst = reinterpret_cast<const uint32_t*>(pool.data)[(size_t)((i < n_active ? i : 0u) + pool.first) * WF_HIT_WORDS + 1u];   // (the hit record's instance, fetched with the flags)
    if (flags & WF_INVERTEX) {
        // (the vertex waits for its stage C ray: k_wf_advance closes it)
    } else if ((flags & LF_ALIVE) && !(flags & WF_HIT_A)) {   // camera miss: black sample; continuation miss: path ends (path.rs:112-115)
        pu(pool, F_FLAGS, i) = (flags & ~(LF_ALIVE | WF_INVERTEX)) | WF_FINISHED;
    } else if (flags & LF_ALIVE) {
        kind = sc.materials[sc.instances[inst].material_id].mat_kind;
    }
    uint32_t rank = 0u;
    if (kind < WF_MAT_KINDS) rank = atomicAdd(&s_cnt[kind], 1u);
    __syncthreads();
    if (tid < WF_MAT_KINDS && s_cnt[tid]) s_base[tid] = atomicAdd(qctl + wf_my_seg() * WF_SEG_STRIDE + 8u + tid, s_cnt[tid]);
    __syncthreads();
    if (kind < WF_MAT_KINDS) kind_queues[((size_t)kind * WF_SEGS + wf_my_seg()) * pool.seg_cap + s_base[kind] + rank] = i;
}
template <int ANIM, int MK>
__global__ __launch_bounds__(TR_BLOCK, WF_SHADE_WAVES) void k_wf_shade_kind(const DevScene scv, WfPool pool, const uint32_t* __restrict__ kind_queues, uint32_t* __restrict__ qctl,
                                                                            DevStats* __restrict__ stats, uint32_t* __restrict__ queue_a, uint32_t* __restrict__ queue_b) {
    const DevScene& sc = scv;
    __shared__ uint4 s_perm[TR_PERM_BYTES / 16];
    __shared__ uint32_t s_a_cnt[WF_SORT_KEYS + 1u], s_a_base[WF_SORT_KEYS], s_b_cnt[WF_SORT_KEYS + 1u], s_b_base[WF_SORT_KEYS];
    if (blockIdx.x / WF_SEGS * TR_BLOCK >= qctl[wf_my_seg() * WF_SEG_STRIDE + 8u + MK]) return;   // (the whole workgroup lies past the queue's end)
    s_perm[threadIdx.x] = reinterpret_cast<const uint4*>(sc.perm_pool)[threadIdx.x];
    if (threadIdx.x <= WF_SORT_KEYS) { s_a_cnt[threadIdx.x] = 0u; s_b_cnt[threadIdx.x] = 0u; }
    __syncthreads();
    uint32_t i;
    if (!wf_my_entry(pool, kind_queues + (size_t)MK * WF_SEGS * pool.seg_cap, qctl, 8u + MK, i)) return;
    constexpr int FEAT = feat_of_material(MK);
    constexpr uint32_t KM = km_of_material(MK);
    const uint32_t flags = pu(pool, F_FLAGS, i);
    HitRec rec;
    ld_hit(pool, i, rec);
    Lane ln;
    ln.perm_lds = TR_LDS_B(s_perm);
    ln.flags = flags & ~WF_PENDING;
    ln.bounce = pu(pool, F_BOUNCE, i); ln.ks = pu(pool, F_KS, i);
    LN_O(ln) = ld3(pool, F_O, i); ln.d = ld3(pool, F_D, i);
    ln.throughput = ld3(pool, F_T, i); ln.illum = ld3(pool, F_ILLUM, i);
    ln.time = ANIM ? pf(pool, F_TIME, i) : 0.0f; ln.col = (ANIM && sc.xf_table) ? pu(pool, F_KIDX, i) : i;
    if (flags & WF_PENDING) ln.illum = wf_apply_pending(pool, i, ln.illum);   // the previous vertex's light term, now that its occlusion ray is traced
    ln.first_ng = (ln.bounce != 0u && (flags & LF_SPECULAR)) ? ld3(pool, F_NG, i) : mk(0.0f, 0.0f, 0.0f);
    Counters cnt;
    cnt.rays = 0; cnt.vertices = 0;
    vertex_begin<ANIM>(sc, ln, rec, cnt);
    bool goes_on = false, shadow = false, skipped = false, mis_miss = false;
    f3 ro = mk(0.0f, 0.0f, 0.0f), rd = ro, bo = ro, bd = ro;
    uint32_t rf = 0u, bf = 0u;
    if (!(ln.flags & LF_ALIVE)) {   // NormalsDebug: the sample ended at its first hit
        pu(pool, F_FLAGS, i) = (ln.flags & ~WF_INVERTEX) | WF_FINISHED;
        st3(pool, F_ILLUM, i, ln.illum);
    } else {
        if (ln.bounce == 0u) st3(pool, F_NG, i, ln.first_ng);
        bo = ln.bsdf.p; bd = ln.aux_d;                    // the occlusion segment (vertex_begin), before the queries reuse aux_d
        const bool had_shadow = (ln.flags & LF_SHADOW) != 0u;
        vertex_queries<ANIM, FEAT, KM>(sc, ln, false);     // the light half as if unoccluded: ln.direct is the term, or stays zero
        mis_ray_filter<ANIM>(sc, ln);
        mis_miss = (ln.flags & LF_MIS_MISS) != 0u;
        ln.flags &= ~(LF_MIS_MISS | LF_MIS_UNTESTED);
        const f3 dl = ln.direct, tv = ln.t_vertex;
        shadow = had_shadow && !is_black(dl);               // a black term: the reference traces the ray and adds nothing
        skipped = had_shadow && !shadow;
        st3(pool, F_T, i, ln.throughput);
        if (!(ln.flags & LF_LAST)) { st3(pool, F_O, i, LN_O(ln)); st3(pool, F_D, i, ln.d); }
        if (ln.flags & LF_MIS) {   // the vertex ends in k_wf_advance, after the BSDF-sampled light ray was traced (WF_FOLD_C): what it needs, field by field
            pu(pool, F_FLAGS, i) = ln.flags | WF_INVERTEX | (shadow ? (uint32_t)WF_PENDING : 0u);
            st3(pool, F_ILLUM, i, ln.illum);
            st3(pool, F_DIRECT, i, dl); st3(pool, F_TV, i, tv);
            st3(pool, F_AUX, i, ln.aux_d); st3(pool, F_MISF, i, ln.mis_f); st3(pool, F_LI, i, ln.li);
            st3(pool, F_P, i, ln.bsdf.p); pu(pool, F_LINST, i) = ln.light_inst;
        } else {
            ln.direct = mk(0.0f, 0.0f, 0.0f);   // (the term is applied when its ray has been traced)
            HitRec none;
            none.t = 0.0f; none.inst = 0xffffffffu; none.prim = 0u; none.b1 = 0.0f; none.b2 = 0.0f;
            const bool cont = vertex_end<ANIM>(sc, ln, false, none);
            st3(pool, F_ILLUM, i, ln.illum);
            pu(pool, F_BOUNCE, i) = ln.bounce;
            uint32_t f2 = (ln.flags & ~(WF_INVERTEX | WF_HIT_A | WF_HIT_C | WF_OCCLUDED | LF_SHADOW | LF_MIS | LF_LAST)) | (shadow ? (uint32_t)WF_PENDING : 0u);
            if (!cont) f2 = (f2 & ~LF_ALIVE) | WF_FINISHED;
            pu(pool, F_FLAGS, i) = f2;
            if (shadow) { st3(pool, F_PEND_TV, i, tv); st3(pool, F_PEND_DL, i, dl); }
            if (cont) { goes_on = true; ro = LN_O(ln); rd = ln.d; rf = f2; }
        }
        bf = ln.flags;
    }
    {   // one counter update per wave: vertices; occlusion rays whose term is black and BSDF-sampled light rays proven to miss: counted like the reference's, not traced
        const unsigned long long m = __ballot(1), sk = __ballot(skipped), mm = __ballot(mis_miss);
        if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) {
            atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].vertices, (unsigned long long)__popcll(m));
            if ((sk | mm) != 0ull) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].rays, (unsigned long long)(__popcll(sk) + __popcll(mm)));
        }
    }
    wf_append_live(pool, queue_b, qctl, 1u, shadow, i, bo, bd, bf | WF_INVERTEX, s_b_cnt, s_b_base);
    wf_append_live(pool, queue_a, qctl, 0u, goes_on, i, ro, rd, rf, s_a_cnt, s_a_base);
}

// Stage B shading of pool slot i: the BSDF queries of the vertex (light half, BSDF half, continuation)
// Returns true when the path goes on from this vertex WITHOUT a stage C ray: (ray_o, ray_d, ray_flags) is then the ray towards the next vertex and
// the slot's flags word as stored -- the caller appends it to the NEXT round's queue A (round 6: k_wf_advance used to re-read origin, direction and
// bounce of every continuing slot from the pool to do that, 64 bytes fetched per slot and round for a ray this kernel holds in registers).
template <int ANIM, int FEAT, uint32_t KM>
TR_DEV bool wf_query_slot(const DevScene& sc, const WfPool& pool, uint32_t i, uint32_t flags, uint32_t* __restrict__ queue_c, uint32_t* __restrict__ qctl,
                           DevStats* __restrict__ stats, LdsB perm_lds, f3& ray_o, f3& ray_d, uint32_t& ray_flags) {
    bool goes_on = false;
    Lane ln;
    ln.perm_lds = perm_lds;
    ln.flags = flags;
    ln.bounce = pu(pool, F_BOUNCE, i); ln.ks = pu(pool, F_KS, i);
    ln.throughput = ld3(pool, F_T, i);
    ld_bsdf(sc, pool, i, ln.bsdf);
    ln.light_inst = pu(pool, F_LINST, i);
    ln.aux_d = mk(0.0f, 0.0f, 0.0f); ln.mis_f = mk(0.0f, 0.0f, 0.0f);   // (before wi_l is loaded: the two share storage)
    ln.li = ld3(pool, F_LI, i); ln.wi_l = ld3(pool, F_WL, i); ln.pdf_l = pf(pool, F_PDFL, i);
    ln.direct = mk(0.0f, 0.0f, 0.0f);   // (as vertex_begin left it)
    ln.t_vertex = ln.throughput;        // (likewise: the throughput the vertex was reached with, before the PATH query updates it)
    ln.d = ld3(pool, F_D, i);           // (the ray that reached the vertex: -d is the outgoing direction until the PATH query replaces d)
    ln.time = ANIM ? pf(pool, F_TIME, i) : 0.0f; ln.col = (ANIM && sc.xf_table) ? pu(pool, F_KIDX, i) : i;
    vertex_queries<ANIM, FEAT, KM>(sc, ln, (flags & WF_OCCLUDED) != 0u);
    mis_ray_filter<ANIM>(sc, ln);
    {   // BSDF-sampled light rays proven to miss the light: counted like the reference's, never queued
        const unsigned long long mm = __ballot((ln.flags & LF_MIS_MISS) != 0u);
        if (mm != 0ull && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)mm) - 1u && stats)
            atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].rays, (unsigned long long)__popcll(mm));
        ln.flags &= ~(LF_MIS_MISS | LF_MIS_UNTESTED);
    }
    st3(pool, F_T, i, ln.throughput);
    if (!(ln.flags & LF_LAST)) { st3(pool, F_O, i, LN_O(ln)); st3(pool, F_D, i, ln.d); }
    if (ln.flags & LF_MIS) {   // the vertex ends in k_wf_advance, after stage C has traced the BSDF-sampled light ray
        pu(pool, F_FLAGS, i) = ln.flags;
        st3(pool, F_DIRECT, i, ln.direct); st3(pool, F_TV, i, ln.t_vertex);
        st3(pool, F_AUX, i, ln.aux_d); st3(pool, F_MISF, i, ln.mis_f); st3(pool, F_LI, i, ln.li);
    } else {   // no stage C ray (the usual case): vertex_end here, while the vertex is in registers
        ln.illum = ld3(pool, F_ILLUM, i);
        HitRec none;
        none.t = 0.0f; none.inst = 0xffffffffu; none.prim = 0u; none.b1 = 0.0f; none.b2 = 0.0f;
        const bool cont = vertex_end<ANIM>(sc, ln, false, none);
        st3(pool, F_ILLUM, i, ln.illum);
        pu(pool, F_BOUNCE, i) = ln.bounce;
        uint32_t f2 = ln.flags & ~(WF_INVERTEX | WF_HIT_A | WF_HIT_C | WF_OCCLUDED | LF_SHADOW | LF_MIS | LF_LAST);
        if (!cont) f2 = (f2 & ~LF_ALIVE) | WF_FINISHED;
        pu(pool, F_FLAGS, i) = f2;
        if (cont) { goes_on = true; ray_o = LN_O(ln); ray_d = ln.d; ray_flags = f2; }
    }
    if (queue_c) wf_enqueue_ray(pool, queue_c, qctl, 2u, (ln.flags & LF_MIS) != 0u, i, ln.bsdf.p, ln.aux_d, ln.flags);
    return goes_on;
}

// one thread per pool slot, every material kind's code: scenes with textured materials (their lobes exist per hit only, so there is no table to sort by)
// queue_a: the NEXT round's stage A queue -- the rays of the paths that go on from here, appended by the whole workgroup in direction-octant order
// (wf_enqueue_ray_by_key; its count word survives until the next round's traversal: the host clears the control words between trace A and k_wf_begin)
template <int ANIM, int FEAT>
__global__ __launch_bounds__(TR_BLOCK) void k_wf_query(
    const DevScene scv, WfPool pool, uint32_t n_active, uint32_t* __restrict__ queue_c, uint32_t* __restrict__ qctl, DevStats* __restrict__ stats,
    uint32_t* __restrict__ queue_a) {
    const DevScene& sc = scv;
    __shared__ uint4 s_perm[TR_PERM_BYTES / 16];   // the permutation pool in LDS (as in k_wf_begin)
    __shared__ uint32_t s_oct_cnt[WF_SORT_KEYS + 1u], s_oct_base[WF_SORT_KEYS];
    s_perm[threadIdx.x] = reinterpret_cast<const uint4*>(sc.perm_pool)[threadIdx.x];
    if (threadIdx.x <= WF_SORT_KEYS) s_oct_cnt[threadIdx.x] = 0u;   // (the octant counts and the ticket of wf_append_next_a_live: while every thread is still here)
    __syncthreads();
    const uint32_t i = blockIdx.x * TR_BLOCK + threadIdx.x;
    if (i >= n_active) return;
    const uint32_t flags = pu(pool, F_FLAGS, i);
    if ((flags & (LF_ALIVE | WF_INVERTEX)) != (LF_ALIVE | WF_INVERTEX) || (flags & (LF_MIS | WF_CFLIGHT)) != 0u) return;   // (a vertex whose queries ran in an earlier round waits for its stage C ray: WF_FOLD_C)
    f3 ro = mk(0.0f, 0.0f, 0.0f), rd = ro;
    uint32_t rf = 0u;
    const bool goes_on = wf_query_slot<ANIM, FEAT, KM_ALL>(sc, pool, i, flags, queue_c, qctl, stats, TR_LDS_B(s_perm), ro, rd, rf);
    wf_append_next_a_live(pool, queue_a, qctl, goes_on, i, ro, rd, rf, s_oct_cnt, s_oct_base);
}

// kind-pure shading: one thread per entry of material kind MK's queue (filled by k_wf_begin's counting sort); only the lobes that
// kind lowers to are compiled in (dev_bsdf.h: km_of_material), so the kernels are small (matte 2 lobes' code instead of 9) and
// every lane of a wave runs the same material's code
template <int ANIM, int MK>
__global__ __launch_bounds__(TR_BLOCK, WF_SHADE_WAVES) void k_wf_query_kind(const DevScene scv, WfPool pool, const uint32_t* __restrict__ kind_queues,
                                                            uint32_t* __restrict__ queue_c, uint32_t* __restrict__ qctl, DevStats* __restrict__ stats,
                                                            uint32_t* __restrict__ queue_a) {
    const DevScene& sc = scv;
    __shared__ uint4 s_perm[TR_PERM_BYTES / 16];   // the permutation pool in LDS (as in k_wf_begin)
    __shared__ uint32_t s_oct_cnt[WF_SORT_KEYS + 1u], s_oct_base[WF_SORT_KEYS];
    if (blockIdx.x / WF_SEGS * TR_BLOCK >= qctl[wf_my_seg() * WF_SEG_STRIDE + 8u + MK]) return;   // (the whole workgroup lies past the queue's end)
    s_perm[threadIdx.x] = reinterpret_cast<const uint4*>(sc.perm_pool)[threadIdx.x];
    if (threadIdx.x <= WF_SORT_KEYS) s_oct_cnt[threadIdx.x] = 0u;   // (the octant counts and the ticket of wf_append_next_a_live: while every thread is still here)
    __syncthreads();
    uint32_t i;
    if (!wf_my_entry(pool, kind_queues + (size_t)MK * WF_SEGS * pool.seg_cap, qctl, 8u + MK, i)) return;
    f3 ro = mk(0.0f, 0.0f, 0.0f), rd = ro;
    uint32_t rf = 0u;
    const bool goes_on = wf_query_slot<ANIM, feat_of_material(MK), km_of_material(MK)>(sc, pool, i, pu(pool, F_FLAGS, i), queue_c, qctl, stats, TR_LDS_B(s_perm), ro, rd, rf);
    wf_append_next_a_live(pool, queue_a, qctl, goes_on, i, ro, rd, rf, s_oct_cnt, s_oct_base);
}

// New camera sample for pool slot i of a chunk that works on tile `tile_idx` (multithreaded.rs:90-96)
template <int ANIM>
TR_DEV void wf_regenerate(const DevScene& sc, const WfPool& pool, uint32_t i, uint32_t tile_idx, const uint2* __restrict__ tiles, uint32_t chunk,
                          uint32_t chunk_stride, uint32_t spp, uint32_t kf, DevStats* __restrict__ stats, f3& ray_o, f3& ray_d) {
    // F_SNEXT: the (pixel, sample) pair k_wf_advance handed to this slot: pixel = pair % 64 in Region order, sample = pair / 64 (the slice's
    // first sample included: tile_idx is the TILE here, the caller has shifted the slice bits of the work item out)
    const uint32_t pair = pu(pool, F_SNEXT, i), pix = pair & 63u, s_next = pair >> 6;
    const uint2 tile = tiles[(tile_idx / chunk) * chunk_stride * chunk + (tile_idx % chunk)];
    const uint32_t px = tile.x * 8u + (pix & 7u), py = tile.y * 8u + (pix >> 3);
    const uint32_t kp = key_pixel(kf, py * sc.width + px);
    float sx, sy, t;
    pixel_sample(kp, s_next, spp, px, py, sx, sy, t);
    const Ray cam = camera_ray<ANIM>(sc, sx, sy, t);
    if (ANIM) { pf(pool, F_TIME, i) = cam.time; pu(pool, F_KIDX, i) = xf_time_index(t); xf_cache_fill(sc, cam.time, i); }
    pu(pool, F_BOUNCE, i) = 0u;
    pu(pool, F_KS, i) = key_sample(kp, s_next);
    pf(pool, F_SX, i) = sx; pf(pool, F_SY, i) = sy;
    st3(pool, F_O, i, cam.o); st3(pool, F_D, i, cam.d);
    ray_o = cam.o; ray_d = cam.d;
    st3(pool, F_T, i, mk(1.0f, 1.0f, 1.0f)); st3(pool, F_ILLUM, i, mk(0.0f, 0.0f, 0.0f));
    const unsigned long long m = __ballot(1);
    if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(&stats[blockIdx.x & (WF_STAT_SLOTS - 1)].samples, (unsigned long long)__popcll(m));
}

// Compacted regeneration: one thread per entry of queue R (slots whose sample finished and whose tile has samples left)
template <int ANIM>
__global__ __launch_bounds__(TR_BLOCK) void k_wf_regen(const DevScene scv, WfPool pool, const WfChunk* __restrict__ chunks, const uint2* __restrict__ tiles,
                                                       uint32_t chunk, uint32_t chunk_stride, uint32_t spp, uint32_t kf, DevStats* __restrict__ stats,
                                                       const uint32_t* __restrict__ queue_r, uint32_t* __restrict__ queue_a, uint32_t* __restrict__ qctl,
                                                       uint32_t slice_shift) {
    const DevScene& sc = scv;
    uint32_t i;
    if (!wf_my_entry(pool, queue_r, qctl, 6u, i)) return;
    f3 ray_o, ray_d;
    wf_regenerate<ANIM>(sc, pool, i, chunks[i / TR_BLOCK].tile >> slice_shift, tiles, chunk, chunk_stride, spp, kf, stats, ray_o, ray_d);
    pu(pool, F_FLAGS, i) = LF_ALIVE;
    wf_enqueue_ray(pool, queue_a, qctl, 0u, true, i, ray_o, ray_d, LF_ALIVE | WF_CAMERA_RAY);
}

// Round head, one workgroup per chunk: vertex_end of the previous round, film splat of finished samples,
// tile completion / switch, path regeneration.
template <int ANIM>
__global__ __launch_bounds__(TR_BLOCK) void k_wf_advance(const DevScene scv, WfPool pool, WfChunk* __restrict__ chunks,
                                                         float* __restrict__ bins, const uint2* __restrict__ tiles, uint32_t tile_count,
                                                         uint32_t chunk, uint32_t chunk_stride, uint32_t spp, uint32_t kf,
                                                         float* __restrict__ rgbw, uint32_t* __restrict__ tile_counter,
                                                         uint32_t* __restrict__ tiles_done, DevStats* __restrict__ stats,
                                                         uint32_t* __restrict__ queue_a, uint32_t* __restrict__ queue_r,
                                                         uint32_t* __restrict__ qctl, uint32_t slice_shift) {
    // A work item is a SLICE of a tile: samples [slice, slice + 1) * (spp >> slice_shift) of its 64 pixels (item = tile << slice_shift | slice;
    // tile_count counts items). The film is a sum and every sample is keyed by pixel and index, so the slices of a tile are independent:
    // launch_wavefront cuts tiles when the pool has more chunks than the launch has tiles -- the schedule's rate grows with the slots in flight
    // (C5 stand-in 78 / 106 / 132 Msamples/s at 2 / 4 / 8 M slots), and one chunk per tile capped them at 4 per pixel.
    const uint32_t s_per = spp >> slice_shift, n_pairs = 64u * s_per;
    __shared__ float s_win[4 * WIN_PLANE];
    __shared__ float s_table[TRAY_FILTER_TABLE_SIZE * TRAY_FILTER_TABLE_SIZE];
    __shared__ float s_tx[TRAY_FILTER_TABLE_SIZE], s_ty[TRAY_FILTER_TABLE_SIZE];
    __shared__ uint32_t s_tile, s_done, s_fin, s_pair;
    __shared__ uint32_t s_oct_cnt[WF_SORT_KEYS], s_oct_base[WF_SORT_KEYS];
    __shared__ float s_bins[ROWBIN_SIZE];   // this round's contribution to the chunk's row bins
    const DevScene& sc = scv;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t c = blockIdx.x;
    const uint32_t i = c * TR_BLOCK + tid;
    float* __restrict__ my_bins = bins + (size_t)c * ROWBIN_SIZE;
    if (tid == 0) { s_tile = chunks[c].tile; s_done = chunks[c].done; s_pair = chunks[c].next_pair; s_fin = 0u; }
    s_table[tid] = sc.filter_table[tid];
    if (tid < TRAY_FILTER_TABLE_SIZE) { s_tx[tid] = sc.filter_x[tid]; s_ty[tid] = sc.filter_y[tid]; }
    for (uint32_t k = tid; k < ROWBIN_SIZE; k += TR_BLOCK) s_bins[k] = 0.0f;
    uint32_t flags = pu(pool, F_FLAGS, i);   // issued before the barrier: this kernel is a chain of dependent loads (95 % of its wave cycles wait)
    __syncthreads();
    uint32_t tile_idx = s_tile;
    if (tile_idx == WF_TILE_IDLE) return;
    const uint32_t flags_in = flags;
    bool c_ray = false;   // the slot's stage C ray joins this round's queue A (WF_FOLD_C)
    bool closed_here = false;   // this kernel ran the slot's vertex_end (the vertex had a stage C ray): if the path goes on, its ray is queued here
    const bool film_rows = sc.film_rows != 0u;
    if (tile_idx != WF_TILE_NEED) {
        const uint32_t tile_of = tile_idx >> slice_shift;
        const uint2 tile = tiles[(tile_of / chunk) * chunk_stride * chunk + (tile_of % chunk)];
        const int x0 = (int)tile.x * 8, y0 = (int)tile.y * 8;
        // ---- stage C shading of the previous round
        if (WF_FOLD_C && (flags & (LF_ALIVE | WF_INVERTEX | LF_MIS | WF_CFLIGHT)) == (LF_ALIVE | WF_INVERTEX | LF_MIS)) {
            // the query kernels left a BSDF-sampled light ray (origin F_P, direction F_AUX): it goes into THIS round's queue A below
            c_ray = true;
            flags |= WF_CFLIGHT;
        } else if ((flags & (LF_ALIVE | WF_INVERTEX)) == (LF_ALIVE | WF_INVERTEX)) {
            Lane ln;
            ln.flags = flags;
            HitRec rec;
            rec.t = 0.0f; rec.inst = 0xffffffffu; rec.prim = 0u; rec.b1 = 0.0f; rec.b2 = 0.0f;
            ln.bounce = pu(pool, F_BOUNCE, i);
            ln.illum = ld3(pool, F_ILLUM, i);
            ln.direct = ld3(pool, F_DIRECT, i);
            if ((flags & WF_PENDING) && pu(pool, F_PEND_OCC, i) != 0u) ln.direct = mk(0.0f, 0.0f, 0.0f);   // (fused shading: the stored light term was speculative, its ray found something)
            ln.t_vertex = ld3(pool, F_TV, i);
            ln.light_inst = pu(pool, F_LINST, i);
            if (flags & LF_MIS) {
                ln.bsdf.p = ld3(pool, F_P, i); ln.aux_d = ld3(pool, F_AUX, i); ln.mis_f = ld3(pool, F_MISF, i); ln.li = ld3(pool, F_LI, i);
                ld_hit(pool, i, rec);
            }
            ln.time = ANIM ? pf(pool, F_TIME, i) : 0.0f; ln.col = (ANIM && sc.xf_table) ? pu(pool, F_KIDX, i) : i;
            const bool cont = vertex_end<ANIM>(sc, ln, (flags & WF_HIT_C) != 0u, rec);
            st3(pool, F_ILLUM, i, ln.illum);
            pu(pool, F_BOUNCE, i) = ln.bounce;
            flags = ln.flags & ~(WF_INVERTEX | WF_HIT_A | WF_HIT_C | WF_OCCLUDED | LF_SHADOW | LF_MIS | LF_LAST | WF_CFLIGHT | WF_PENDING);
            if (!cont) flags = (flags & ~LF_ALIVE) | WF_FINISHED;
            closed_here = true;
        }
        // ---- RenderTarget::write of the samples that finished (here or in k_wf_begin)
        if (flags & WF_FINISHED) {
            f3 il = ld3(pool, F_ILLUM, i);
            if (flags & WF_PENDING) { il = wf_apply_pending(pool, i, il); flags &= ~WF_PENDING; }   // (fused shading: the last vertex's light term)
            const float sx = pf(pool, F_SX, i), sy = pf(pool, F_SY, i);
            const f3 col = mk(clampf(il.x, 0.0f, 1.0f), clampf(il.y, 0.0f, 1.0f), clampf(il.z, 0.0f, 1.0f));   // quirk Q3
            if (film_rows) film_splat_rows_global<true>(sc, s_bins, s_tx, rgbw, s_table, x0, y0, (int)((pu(pool, F_SNEXT, i) & 63u) >> 3), sx, sy, col);
            else film_splat_global(sc, rgbw, s_table, x0, y0, sx, sy, col);
            flags &= ~WF_FINISHED;
            atomicAdd(&s_fin, 1u);
        }
        __syncthreads();
        if (film_rows && s_fin != 0u) {
            // the chunk owns its bins, so this is a plain sum -- issued as no-return atomics: a read-modify-write loop is a chain of
            // 17 dependent HBM round trips per thread in a kernel that does little else, the atomic is fire-and-forget
            for (uint32_t k = tid; k < ROWBIN_SIZE; k += TR_BLOCK) {
                const float v = s_bins[k];
                if (v != 0.0f) wg_add(my_bins + k, v);
            }
        }
        // ---- tile complete: spread the row bins over the window, flush it, take the next tile
        const uint32_t done = s_done + s_fin;
        __syncthreads();   // everybody has read s_done / s_fin before thread 0 rewrites them; the bins are up to date
        if (done == n_pairs) {
            if (film_rows) {
                for (uint32_t k = tid; k < 4 * WIN_PLANE; k += TR_BLOCK) s_win[k] = 0.0f;
                __syncthreads();
                film_resolve_rows(sc, my_bins, s_ty, s_win, y0, tid);
                __syncthreads();
                const int wx0 = x0 - sc.fpw, wy0 = y0 - sc.fph;
                const int ww = 8 + 2 * sc.fpw + 1, wh = 8 + 2 * sc.fph + 1;
                for (int k = (int)tid; k < ww * wh; k += TR_BLOCK) {
                    int wy = k / ww, wx = k - wy * ww;
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
                for (uint32_t k = tid; k < ROWBIN_SIZE; k += TR_BLOCK) my_bins[k] = 0.0f;
            }
            if (tid == 0) { atomicAdd(tiles_done, 1u); s_tile = WF_TILE_NEED; s_done = 0u; s_pair = 0u; }
            tile_idx = WF_TILE_NEED;
        } else if (tid == 0) {
            s_done = done;
        }
    }
    __syncthreads();
    if (tile_idx == WF_TILE_NEED) {
        if (tid == 0) {
            uint32_t t = atomicAdd(tile_counter, 1u);
            s_tile = t < tile_count ? t : WF_TILE_IDLE;
        }
        __syncthreads();
        tile_idx = s_tile;
        flags = 0u;
    }
    // ---- path regeneration (multithreaded.rs:90-96), deferred to k_wf_regen so that the camera rays and the per-path transforms of
    // moving scenes are computed by full waves. The (pixel, sample) pairs of the chunk's tile are handed out dynamically, as in the
    // tile kernel: an idle slot takes the next pair of the chunk's counter whatever pixel it belongs to, so no slot sits out the end
    // of a tile because its own pixel's samples are used up while other pixels' are not.
    const bool idle = tile_idx != WF_TILE_IDLE && !(flags & LF_ALIVE);
    bool wants_sample = false;
    {
        const unsigned long long im = __ballot(idle);
        if (im != 0ull) {
            const uint32_t leader = (uint32_t)__ffsll((long long)im) - 1u;
            uint32_t base = 0u;
            if (lane == leader) base = atomicAdd(&s_pair, (uint32_t)__popcll(im));
            base = __shfl(base, (int)leader);
            const uint32_t pair = base + (uint32_t)__popcll(im & ((1ull << lane) - 1ull));
            // (stored with the slice's first sample in the sample bits: wf_regenerate and the film's row bins read pixel and sample from it)
            if (idle && pair < n_pairs) { pu(pool, F_SNEXT, i) = pair + (((tile_idx & ((1u << slice_shift) - 1u)) * s_per) << 6); wants_sample = true; }
        }
    }
    wf_enqueue(pool, queue_r, qctl, 6u, wants_sample, i);
    if (flags != flags_in) pu(pool, F_FLAGS, i) = flags;   // (a slot that simply goes on is not touched beyond the read of this word)
    {   // Round 6: the ray of a path that goes on was put into this round's queue A by the shading kernel that sampled it (k_wf_query_kind, last
        // round: it has origin and direction in registers -- this kernel used to fetch the slot's 64-byte ray record for them, per slot and
        // round). What is queued HERE: the stage C rays (WF_FOLD_C) and the rays of the few paths whose vertex this kernel just closed.
        const bool cont_ray = c_ray || (closed_here && (flags & LF_ALIVE) != 0u);
        f3 ro = mk(0.0f, 0.0f, 0.0f), rd = ro;
        if (c_ray) { ro = ld3(pool, F_P, i); rd = ld3(pool, F_AUX, i); }   // Ray::segment(p, w_i, 0.001, inf) (mod.rs:154): never a camera ray
        else if (cont_ray) { ro = ld3(pool, F_O, i); rd = ld3(pool, F_D, i); }   // (never a camera ray either: bounce >= 1 after a vertex_end; k_wf_regen queues those)
        wf_append_next_a(pool, queue_a, qctl, cont_ray, i, ro, rd, flags, s_oct_cnt, s_oct_base);
    }
    __syncthreads();   // every wave has taken its pairs
    if (tid == 0) { chunks[c].tile = s_tile; chunks[c].done = s_done; chunks[c].next_pair = s_pair < n_pairs ? s_pair : n_pairs; }
}

}  // namespace tr
