// Device-side C ABI of libtrayhip.so: device scenes (upload, frame update), the launches of the tile kernel / the wavefront schedule / the
// samplers, the transform table, the multi-device path (one stream per device + one RCCL sum-reduce), the debug entry points.
// The kernels themselves are kernels.hip + the dev_*.h / wavefront.h headers; their template instantiations are compiled in the objects of
// kernel_group.hip (kernel_list.h) -- this translation unit only DECLARES them, so a change here recompiles no kernel.
#include "kernels.hip"
#include <cmath>
#define TR_INST_EXTERN   // explicit instantiation declarations of every kernel the launch sites below name
#include "kernel_list.h"
#undef TR_INST_EXTERN


#ifndef WF_PIPES_MAX
#define WF_PIPES_MAX 4   // views of the wavefront schedule (WfView below) the buffers are sized for
#endif
struct TrayDevBuf { const char* key; void* ptr; size_t bytes; };
// What tray_scene_update_frame requires of its argument: the SAME scene at another frame. The device buffers a frame update takes
// over without a copy (meshes, trees, triangles, MERL tables, textures, permutation pool, filter tables) are recognised by name and size
// only, so the sizes and parameters they depend on are recorded at creation and compared before anything is moved.
struct TraySceneIdentity {
    uint32_t n_instances, n_meshes, n_mesh_nodes, n_tris, n_materials, n_merl, n_textures, n_tex_frames, max_depth, min_depth, integrator, width, height;
    uint64_t n_merl_floats, n_tex_bytes;
    float filter[4];
    int32_t filter_px[2];
    uint32_t filter_hash;
    bool operator==(const TraySceneIdentity& o) const { return std::memcmp(this, &o, sizeof *this) == 0; }
};
static TraySceneIdentity scene_identity(const TrayFlatScene* f) {
    TraySceneIdentity id;
    std::memset(&id, 0, sizeof id);   // (padding bytes take part in the comparison)
    id.n_instances = f->n_instances; id.n_meshes = f->n_meshes; id.n_mesh_nodes = f->n_mesh_nodes; id.n_tris = f->n_tris; id.n_materials = f->n_materials;
    id.n_merl = f->n_merl; id.n_textures = f->n_textures; id.n_tex_frames = f->n_tex_frames; id.max_depth = f->max_depth; id.min_depth = f->min_depth;
    id.integrator = f->integrator; id.width = f->film.width; id.height = f->film.height;
    id.n_merl_floats = f->n_merl_floats; id.n_tex_bytes = f->n_tex_bytes;
    id.filter[0] = f->film.filter_w; id.filter[1] = f->film.filter_h; id.filter[2] = f->film.inv_w; id.filter[3] = f->film.inv_h;
    id.filter_px[0] = f->film.filter_pixel_w; id.filter_px[1] = f->film.filter_pixel_h;
    uint32_t h = 2166136261u;   // FNV-1a over the filter table's bits
    for (int k = 0; k < TRAY_FILTER_TABLE_SIZE * TRAY_FILTER_TABLE_SIZE; ++k) { uint32_t w; std::memcpy(&w, &f->film.table[k], 4); h = (h ^ w) * 16777619u; }
    id.filter_hash = h;
    return id;
}
struct TrayDeviceScene {
    int device = 0;
    DevScene dev{};
    std::vector<void*> allocs;
    std::vector<uint32_t> mesh_depths;   // deepest node of every BVH<Triangle> (scene_build: traversal stack size)
    std::vector<TrayDevBuf> bufs;        // the named uploads among `allocs`: what tray_scene_update_frame can carry over to the next frame
    TraySceneIdentity identity{};
    TrayDeviceScene* donor = nullptr;    // while a frame update builds the new state: the previous frame's scene, whose buffers may be taken
    size_t xf_cache_bytes = 0;
    bool broken = false;                 // a frame update failed half way: only tray_scene_destroy is valid
    uint2* d_tiles = nullptr;      // full Morton queue
    uint32_t n_tiles = 0;
    uint32_t* d_counter = nullptr;
    DevStats* d_stats = nullptr;
    uint32_t* d_retraced = nullptr;   // rays the flat instance loop re-traced through BVH<Instance> (dev_geom.h: trace)
    TrayInstance* d_instances = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_valid = false;
    bool empty_launch = false;           // the last render call had no tiles to render (reported as a launch of zero samples)
    uint32_t launches = 0;
    int n_blocks = 0;
    uint32_t n_materials = 0;
    // wavefront mode (lazily allocated)
    WfPool pool{nullptr, 0};
    bool wf_ready = false;               // every buffer below exists (a failed allocation leaves this false for good)
    WfChunk* d_chunks = nullptr;
    float* d_bins = nullptr;
    uint32_t* d_wf_counters = nullptr;   // [0] tile counter, [1] tiles done
    uint32_t* h_done = nullptr;          // pinned host mirror of tiles done
    uint32_t n_chunks = 0;
    uint32_t stack_bytes = 0;   // dynamic LDS of every kernel that traverses: stack depth x TR_BLOCK x 4
    bool wavefront = false;   // TRAYHIP_MODE=wave selects the stage-kernel schedule (wavefront.h)
    bool animated = false;    // something moves while the shutter is open: the <ANIM = true> kernels run
    uint32_t deferred_n_moving = 0;
    int feat = FEAT_ALL;              // lobe kinds of the scene's materials that need the large kernels (dev_bsdf.h)
    uint32_t* d_queues = nullptr;     // wavefront schedule: ray queues A, B, C (n_slots each) + WF_QCTL_WORDS counters
    uint32_t n_blocks_trace = 0;      // persistent grid of k_wf_trace_dyn
    std::vector<TrayMesh> paired_meshes;   // the meshes with node_offset / node_count in device order (what the `meshes` buffer holds)
    std::vector<uint32_t> quad_first;      // per mesh: entry record of its tree in the `mesh_quads` buffer (host/gates.hpp: QuadTrees)
    size_t n_mesh_quads = 0;               // records of the BVH<Triangle>s at the head of the `quads` buffer
    uint32_t quad_mesh_pend = 0;           // most node entries a traversal of one BVH<Triangle>'s quad records can have pending
    uint32_t quad_stack_words = 0;         // stack words per lane the wavefront traversal needs for this frame's trees
    uint32_t* d_fallback = nullptr;        // slots of the rays k_wf_trace_dyn hands to k_wf_trace_fallback (one word per pool slot)
    bool narrow_trees = true;         // every node's offset fits a descriptor (host/gates.hpp): the wavefront traversal keeps nodes as descriptors
    bool ordered_boxes = true;        // every BVH box has min <= max (host/gates.hpp: QuadTrees::ordered)
    uint32_t trace_lds_depth = 0, trace_lds_bytes = 0;   // LDS part of the dynamic-fetch kernel's stacks; deeper entries go to d_stack_overflow
    uint32_t* d_stack_overflow = nullptr;
    size_t ovf_entries = 0;              // ... per view of the schedule (WF_PIPES_MAX of them)
    DevScene launch_dev{};               // what the kernels of the current render call get: `dev`, or `dev` with the transform table in the cache's place
    bool camera_animated = false;
    float* d_xf_table = nullptr;         // the frame's transform table (dev_geom.h: xf_time_index), built by the first launch that wants it
    uint32_t xf_table_stride = 0;
    uint32_t xf_table_cap = 0;           // records per time index the buffer has room for (>= xf_table_stride: the next frames of a sequence may move more instances)
    uint32_t xf_movable = 0;             // instances (+ camera) whose transform has a level of several keyframes: what ANY frame of the sequence can move
    bool xf_table_built = false;         // ... for THIS frame (a frame update takes the buffer over and builds anew)
    hipEvent_t xf_table_ev = nullptr;    // recorded behind k_xf_table_build: a later launch of the frame on ANOTHER stream waits for it
    hipStream_t xf_table_stream = nullptr;   // the stream the build was put on
    int xf_table_req = -1;               // tray_scene_set_transform_table: -1 = by the launch's sample count, 0 = never, 1 = always
    bool last_used_table = false;
    std::vector<void*> wf_allocs;        // the wavefront buffers among `allocs` (tray_scene_set_wavefront frees them to change the pool's size)
    uint32_t wf_req_slots = 0, wf_req_views = 0, wf_req_slices = 0;   // tray_scene_set_wavefront: 0 = the library's own rule
    bool wf_shrunk = false;              // the pool came out smaller than asked for (allocation failed, halved): frame updates keep it
    uint32_t last_views = 0, last_slices = 0;   // shape of the last wavefront launch (tray_last_schedule)
    bool last_was_wavefront = false;
    hipStream_t wf_streams[WF_PIPES_MAX] = {nullptr, nullptr, nullptr, nullptr};   // streams of views 1.. (view 0 runs on the caller's), created on first use
    hipEvent_t wf_fork = nullptr, wf_join[WF_PIPES_MAX] = {nullptr, nullptr, nullptr, nullptr};
    bool light_filter = false;        // a sphere light or specular lobes: the tile kernel with mis_ray_filter (dev_integrator.h) compiled in
    bool wf_sort = true;              // material sort of the shading stage (k_wf_begin's LDS counting sort -> k_wf_query_kind); off for textured scenes
    uint32_t* d_kind_queues = nullptr;   // WF_MAT_KINDS x n_slots slot indices
    uint32_t* d_bin_ctl = nullptr;       // ray binning before the traversal stages (wavefront.h: k_wf_bin_hist): per view and stage, histogram + cursors of every segment
    WfBinGrid bin_grid{};                // the cells of the frame's BVH<Instance> box
    bool wf_fused = true;                // fused shading between the traversals (wavefront.h: k_wf_shade_kind); TRAYHIP_WF_FUSED=0|1 overrides
    uint32_t wf_bin_stages = 0u;         // bit 0: stage A rays are binned, bit 1: stage B rays (TRAYHIP_WF_BIN overrides)
    uint32_t mat_kinds_present = 0;   // bit per TRAY_MAT_* kind among the scene's materials
    // tray_scene_set_sampler: which Sampler the render calls stand for, and the per-pixel state of k_sampler_pass / k_sampler_decide
    bool deforming = false;           // the scene holds an AnimatedMesh: every render runs k_sampler_pass<3> (dev_geom.h: ANIM = 3), the debug kernels their <3> forms
    uint32_t sampler_kind = TRAY_SAMPLER_LOW_DISCREPANCY, smp_min = 1, smp_max = 1;
    void* d_smp = nullptr;            // [state u32 | running average f32 | luminances f32 x cap] per pixel of a batch of tiles
    size_t smp_bytes = 0;
};

static thread_local int g_device = 0;

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));                          \
            return TRAY_E_DEVICE;                                                                         \
        }                                                                                                 \
    } while (0)

static void forget_alloc(TrayDeviceScene* s, void* p) {
    s->allocs.erase(std::remove(s->allocs.begin(), s->allocs.end(), p), s->allocs.end());
}
// Device copy of a host array under a name. During a frame update (s->donor set) a buffer of the same name and size is taken from
// the previous frame's scene instead of allocated: `unchanged` arrays (meshes, MERL tables, textures, the tile queue: the caller
// passes the same scene at another frame, tray_scene_update_frame) keep their content, the others are overwritten.
template <class T>
static int upload(TrayDeviceScene* s, const char* key, bool unchanged, const T* host, size_t n, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    void* d = nullptr;
    bool reused = false;
    if (TrayDeviceScene* don = s->donor)
        for (size_t i = 0; i < don->bufs.size(); ++i)
            if (std::strcmp(don->bufs[i].key, key) == 0 && don->bufs[i].bytes == bytes) {
                d = don->bufs[i].ptr;
                forget_alloc(don, d);
                don->bufs.erase(don->bufs.begin() + (long)i);
                reused = true;
                break;
            }
    if (!d) HIP_CHECK(hipMalloc(&d, bytes));
    s->allocs.push_back(d);
    s->bufs.push_back(TrayDevBuf{key, d, bytes});
    if (!(reused && unchanged)) {
        if (n) HIP_CHECK(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
        else HIP_CHECK(hipMemset(d, 0, bytes));
    }
    *out = static_cast<const T*>(d);
    return TRAY_OK;
}


#ifndef WF_SLOTS
// Path pool slots the schedule asks for (capped by the film: a chunk of 256 per tile slice, 16 slices per tile -- 132.7 M at 1080p -- and by a third of the
// free memory; 0.47 KB per slot with queues and bins: 62 GB for a 1080p film on a 288 GB part). Every stage kernel ends with the tail of its slowest rays,
// and fewer, larger rounds pay it less often: 78 / 106 / 132 / 145 / 152 Msamples/s at 2 / 4 / 8 / 16 / 32 M slots (round 4); round 6, frame 64 / frame 127
// of the C5 stand-in at 512 spp: 237.6 / 169.7 at 32 M (4 slices per tile), 250 - 253 / 174 - 175 at 64 M (8), 256.6 / 177.7 at 128 M (16) -- as long as the
// pool's chunks and the work items are EQUAL in number (96 M slots under 16 slices: 242.0 / 170.5, a quarter of the chunks takes a second item while the others
// idle): launch_wavefront cuts tiles until the items fill the chunks and no further (profiles/r06_c5_pool_64m.txt, r06_c5_pool_128m.txt).
#define WF_SLOTS (128u << 20)
#endif
#define WF_POLL 16
#ifndef WF_MAX_SLICES
#define WF_MAX_SLICES 16u  // work items a tile's samples are cut into at most (k_wf_advance): the pool may hold that many chunks per tile
#endif
// one round of the wavefront schedule: advance -> regen -> trace A -> begin -> trace B -> query -> trace C (compacted ray queues, persistent
// traversal with dynamic fetch, kind-pure shading over the material sort's queues; scenes with textured materials, whose lobes
// exist per hit only, shade unsorted in the one instantiation that lowers them)
// A VIEW of the wavefront buffers: a range of the pool's chunks with queues, control words and overflow columns of its own. The schedule
// runs WF_PIPES views, each on its own stream: every stage kernel ends with the tail of its slowest rays (trace C is nearly all
// tail: a few thousand rays, ~600 us for the longest chain of dependent fetches), and while one view's kernel drains, the
// workgroups it frees run the other view's kernels. The views share the tile counter, so the split does not unbalance them.
struct WfView {
    DevScene dev;       // the scene with xf_cache moved to the view's first slot
    WfPool pool;        // data moved to the view's first slot, n_slots = the pool's stride, seg_cap of the view's chunks
    WfChunk* chunks;
    float* bins;
    uint32_t *qa, *qb, *qc, *qr, *qctl, *kq, *overflow, *fallback, *bin_ctl;
    uint32_t n_chunks;
    hipStream_t stream;
};
template <int ANIM>
static void wf_round(TrayDeviceScene* s, const WfView& v, const uint2* tiles, uint32_t tile_count, uint32_t chunk, uint32_t chunk_stride, uint32_t spp,
                     uint32_t kf, float* rgbw_dev, uint32_t slice_shift) {
    const dim3 grid(v.n_chunks), block(TR_BLOCK);
    const dim3 qgrid((v.n_chunks + WF_SEGS - 1u) / WF_SEGS * WF_SEGS);   // one-thread-per-entry kernels: block b reads segment b % WF_SEGS
    const dim3 tgrid(std::min<uint32_t>(s->n_blocks_trace, v.n_chunks));
    const uint32_t n_active = v.n_chunks * TR_BLOCK;
    hipStream_t stream = v.stream;
    hipLaunchKernelGGL(k_wf_advance<ANIM>, grid, block, 0, stream, v.dev, v.pool, v.chunks, v.bins, tiles, tile_count, chunk, chunk_stride,
                       spp, kf, rgbw_dev, s->d_wf_counters, s->d_wf_counters + 1, s->d_stats, v.qa, v.qr, v.qctl, slice_shift);
    hipLaunchKernelGGL(k_wf_regen<ANIM>, qgrid, block, 0, stream, v.dev, v.pool, v.chunks, tiles, chunk, chunk_stride, spp, kf, s->d_stats, v.qr, v.qa, v.qctl, slice_shift);
    // (each traversal is followed by the few-thread kernel that traces the rays it handed over to the reference's binary traversal:
    // direction components that are zero / denormal / not finite -- normally none, the kernel reads one word and exits. The deferred rays'
    // records go to the buffer of a ray queue that is idle during the stage: B's during A, C's during B, B's during C)
    const dim3 fgrid(8);
    // ray binning (wavefront.h: k_wf_bin_hist / k_wf_bin_scatter): the stage's queue, every segment sorted by (origin cell, direction octant) into the
    // ray queue that is idle during the stage -- C's for stage A (WF_FOLD_C: nobody fills it), A's for stage B (consumed by then) --, which is what
    // the traversal then draws from; the fallback records go where they went (B's buffer during A, C's during B: the sorted copy is consumed by then)
    const dim3 bgrid((v.pool.seg_cap + WF_BIN_EPB - 1u) / WF_BIN_EPB * WF_SEGS);
    const uint32_t* trace_a = v.qa;
    if (WF_FOLD_C && v.bin_ctl && (s->wf_bin_stages & 1u)) {
        hipLaunchKernelGGL(k_wf_bin_hist<0>, bgrid, block, 0, stream, v.pool, v.qa, v.qctl, v.bin_ctl, s->bin_grid);
        hipLaunchKernelGGL(k_wf_bin_scatter<0>, bgrid, block, 0, stream, v.pool, v.qa, v.qc, v.qctl, v.bin_ctl, s->bin_grid);
        trace_a = v.qc;
    }
    hipLaunchKernelGGL((k_wf_trace_dyn<0, ANIM>), tgrid, block, s->trace_lds_bytes, stream, v.dev, v.pool, trace_a, v.qctl, s->d_stats, s->trace_lds_depth, v.overflow, v.qb, 0u);
    hipLaunchKernelGGL((k_wf_trace_fallback<0, ANIM>), fgrid, block, s->stack_bytes, stream, v.dev, v.pool, v.qctl, v.qb, 0u);
    // the control words of every queue are cleared HERE, between the traversal of stage A and the first shading kernel: queue A, its cursors, the
    // regeneration queue and stage A's fallback counter are consumed, stage B's and the material kinds' are not produced yet -- and the count of
    // queue A must survive from the query kernels below (which append the NEXT round's continuation rays) to the next round's traversal
    (void)hipMemsetAsync(v.qctl, 0, WF_QCTL_WORDS * sizeof(uint32_t), stream);
    // Fused shading (wavefront.h: k_wf_sort + k_wf_shade_kind; scenes whose materials are sorted by kind, i.e. without textured ones): the vertex is shaded in ONE
    // kernel between the two traversals, its light term pending until the occlusion ray is traced; TRAYHIP_WF_FUSED=0: the two-kernel form of rounds 2-5
    const uint32_t fused = (v.kq && s->wf_fused && WF_FOLD_C) ? 1u : 0u;
    if (fused) {
        hipLaunchKernelGGL(k_wf_sort<0>, grid, block, 0, stream, v.dev, v.pool, n_active, v.qctl, v.kq);
#define WF_SHADE_KIND(K) if (s->mat_kinds_present & (1u << K)) hipLaunchKernelGGL((k_wf_shade_kind<ANIM, K>), qgrid, block, 0, stream, v.dev, v.pool, v.kq, v.qctl, s->d_stats, v.qa, v.qb)
        WF_SHADE_KIND(TRAY_MAT_MATTE); WF_SHADE_KIND(TRAY_MAT_PLASTIC); WF_SHADE_KIND(TRAY_MAT_METAL); WF_SHADE_KIND(TRAY_MAT_GLASS);
        WF_SHADE_KIND(TRAY_MAT_ROUGH_GLASS); WF_SHADE_KIND(TRAY_MAT_SPECULAR_METAL); WF_SHADE_KIND(TRAY_MAT_MERL);
#undef WF_SHADE_KIND
        hipLaunchKernelGGL((k_wf_trace_dyn<1, ANIM>), tgrid, block, s->trace_lds_bytes, stream, v.dev, v.pool, v.qb, v.qctl, s->d_stats, s->trace_lds_depth, v.overflow, v.qc, 1u);
        hipLaunchKernelGGL((k_wf_trace_fallback<1, ANIM>), fgrid, block, s->stack_bytes, stream, v.dev, v.pool, v.qctl, v.qc, 1u);
        return;
    }
    hipLaunchKernelGGL(k_wf_begin<ANIM>, grid, block, 0, stream, v.dev, v.pool, n_active, s->d_stats, v.qb, v.qctl, v.kq);
    const uint32_t* trace_b = v.qb;
    if (WF_FOLD_C && v.bin_ctl && (s->wf_bin_stages & 2u)) {
        uint32_t* const ctl_b = v.bin_ctl + 2u * WF_SEGS * WF_BINS;
        hipLaunchKernelGGL(k_wf_bin_hist<1>, bgrid, block, 0, stream, v.pool, v.qb, v.qctl, ctl_b, s->bin_grid);
        hipLaunchKernelGGL(k_wf_bin_scatter<1>, bgrid, block, 0, stream, v.pool, v.qb, v.qa, v.qctl, ctl_b, s->bin_grid);
        trace_b = v.qa;
    }
    hipLaunchKernelGGL((k_wf_trace_dyn<1, ANIM>), tgrid, block, s->trace_lds_bytes, stream, v.dev, v.pool, trace_b, v.qctl, s->d_stats, s->trace_lds_depth, v.overflow, v.qc, 0u);
    hipLaunchKernelGGL((k_wf_trace_fallback<1, ANIM>), fgrid, block, s->stack_bytes, stream, v.dev, v.pool, v.qctl, v.qc, 0u);
    uint32_t* const qc = WF_FOLD_C ? nullptr : v.qc;   // (WF_FOLD_C: stage C rays travel with the next round's stage A rays, no queue and no launch of their own)
    if (v.kq) {   // kind-pure shading over the sorted queues: one launch per material kind the scene contains
#define WF_QUERY_KIND(K) if (s->mat_kinds_present & (1u << K)) hipLaunchKernelGGL((k_wf_query_kind<ANIM, K>), qgrid, block, 0, stream, v.dev, v.pool, v.kq, qc, v.qctl, s->d_stats, v.qa)
        WF_QUERY_KIND(TRAY_MAT_MATTE); WF_QUERY_KIND(TRAY_MAT_PLASTIC); WF_QUERY_KIND(TRAY_MAT_METAL); WF_QUERY_KIND(TRAY_MAT_GLASS);
        WF_QUERY_KIND(TRAY_MAT_ROUGH_GLASS); WF_QUERY_KIND(TRAY_MAT_SPECULAR_METAL); WF_QUERY_KIND(TRAY_MAT_MERL);
#undef WF_QUERY_KIND
    } else hipLaunchKernelGGL((k_wf_query<ANIM, FEAT_ALL | FEAT_TEX>), grid, block, 0, stream, v.dev, v.pool, n_active, qc, v.qctl, s->d_stats, v.qa);
    if (WF_FOLD_C) return;
    // (builds without WF_FOLD_C: the deferred rays of stage C go to B's buffer -- A's holds the next round's continuation rays by now)
    hipLaunchKernelGGL((k_wf_trace_dyn<2, ANIM>), tgrid, block, s->trace_lds_bytes, stream, v.dev, v.pool, v.qc, v.qctl, s->d_stats, s->trace_lds_depth, v.overflow, v.qb, 0u);
    hipLaunchKernelGGL((k_wf_trace_fallback<2, ANIM>), fgrid, block, s->stack_bytes, stream, v.dev, v.pool, v.qctl, v.qb, 0u);
}

// Path pool slots of the wavefront schedule: never more than the film has pixels x 4 x WF_MAX_SLICES (a chunk of 256 per tile slice), and for
// moving scenes never more than the per-path transform cache (112 B per slot and instance that moves within the frame) can hold within two
// fifths of the device's free memory; a few hundred thousand slots already fill the chip, but every stage kernel ends with the tail of its
// slowest rays and fewer, larger rounds pay it less often
// reclaimable: bytes a frame update's donor still holds that the new frame either takes over or frees (its pool and transform cache):
// they count as free, or the budget -- and with it the pool size -- would depend on which frame came first
// bytes of the schedule's buffers per pool slot (pool fields, three ray queues + regeneration queue, kind queues, fallback word, row bins per chunk)
static size_t wf_bytes_per_slot() {
    return (size_t)F_COUNT * sizeof(float) + (3 * WF_RAY_WORDS + 1) * sizeof(uint32_t) + WF_MAT_KINDS * sizeof(uint32_t) +
           ((size_t)ROWBIN_SIZE * sizeof(float) + sizeof(WfChunk)) / TR_BLOCK;
}
// what the caller (tray_scene_set_wavefront), the environment or the default ask for, before memory is looked at
static uint32_t wf_slot_wish(const TrayDeviceScene* s) {
    uint32_t n_slots = WF_SLOTS;
    if (s->wf_req_slots) n_slots = std::max<uint32_t>(s->wf_req_slots, 64u * TR_BLOCK) / TR_BLOCK * TR_BLOCK;   // tray_scene_set_wavefront
    if (const char* e = getenv("TRAYHIP_WF_SLOTS")) n_slots = (uint32_t)std::max(256l, atol(e)) / TR_BLOCK * TR_BLOCK;
    const uint64_t by_tiles = (uint64_t)std::max<uint32_t>(s->n_tiles, 1u) * TR_BLOCK * WF_MAX_SLICES;   // (a tile's samples can be cut into that many work items: launch_wavefront)
    return (uint32_t)std::min<uint64_t>(n_slots, by_tiles);
}
// with_cache: the launch evaluates transforms per path (xf_cache_ensure will want 112 B per slot and moving instance); a launch that reads the
// frame's transform table instead allocates no cache, and its pool is bounded by the pool's own bytes alone
static uint32_t wf_slot_count(const TrayDeviceScene* s, size_t reclaimable = 0, bool with_cache = true) {
    uint64_t slots = wf_slot_wish(s);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)16 << 30; }
    // the pool with its queues and bins (~0.47 KB per slot: 15 GB at 32 M) stays within a third of what is free now -- static scenes too
    // (a host that keeps several device scenes gets pools that fit beside each other; launch_wavefront halves on top if hipMalloc refuses)
    {
        const uint64_t fit = (uint64_t)(free_b + reclaimable) / 3u / wf_bytes_per_slot() / TR_BLOCK * TR_BLOCK;
        slots = std::min<uint64_t>(slots, std::max<uint64_t>(fit, (uint64_t)64 * TR_BLOCK));
    }
    if (with_cache && s->animated && s->deferred_n_moving > 0) {
        uint64_t budget = (free_b + reclaimable) / 5 * 2;   // (two fifths of the free memory: 112 B per slot and instance that moves within the frame -- the C5 stand-in has 2 .. 11 of them, 7 .. 39 GB at 32 M slots)
        if (const char* e = getenv("TRAYHIP_XF_CACHE_BYTES")) budget = (uint64_t)std::max(0ll, atoll(e));
        const uint64_t per_slot = (uint64_t)s->deferred_n_moving * TR_XF_REC * sizeof(float);
        const uint64_t fit = budget / per_slot / TR_BLOCK * TR_BLOCK;
        slots = std::min<uint64_t>(slots, std::max<uint64_t>(fit, (uint64_t)64 * TR_BLOCK));   // (at least 64 chunks: below that the schedule cannot fill the chip)
    }
    // (round 6) the schedule is fastest with ONE work item per chunk (profiles/r06_c5_pool_128m.txt: 96 M slots under 16 slices per tile lose to 64 M under 8), so a
    // pool that memory cut below the film's 16 slices per tile is cut further to the next tiles x 256 x 2^k below it -- never to less than one chunk per tile
    {
        const uint64_t per_tile_level = (uint64_t)std::max<uint32_t>(s->n_tiles, 1u) * TR_BLOCK;
        if (slots > per_tile_level && !s->wf_req_slots && !getenv("TRAYHIP_WF_SLOTS")) {
            uint64_t fit = per_tile_level;
            while (fit * 2u <= slots && fit * 2u <= per_tile_level * WF_MAX_SLICES) fit *= 2u;
            slots = fit;
        }
    }
    return (uint32_t)slots;
}

extern "C" {

int tray_device_count(int* n) {
    if (!n) { set_error("tray_device_count: null argument"); return TRAY_E_INVALID; }
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; set_error(std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e)); return TRAY_E_DEVICE; }
    return TRAY_OK;
}

int tray_init(int device) {
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { set_error("tray_init: no such HIP device " + std::to_string(device)); return TRAY_E_INVALID; }
    HIP_CHECK(hipSetDevice(device));
    g_device = device;
    return TRAY_OK;
}

void tray_scene_destroy(TrayDeviceScene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->d_smp) (void)hipFree(s->d_smp);
    if (s->h_done) (void)hipHostFree(s->h_done);
    for (int k = 0; k < WF_PIPES_MAX; ++k) {
        if (s->wf_streams[k]) (void)hipStreamDestroy(s->wf_streams[k]);
        if (s->wf_join[k]) (void)hipEventDestroy(s->wf_join[k]);
    }
    if (s->wf_fork) (void)hipEventDestroy(s->wf_fork);
    if (s->xf_table_ev) (void)hipEventDestroy(s->xf_table_ev);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

} // extern "C" (scene_build is internal)

// the k_path_tiles instantiation launch_tiles runs for this scene (same selection as its PATH_TILES_F): what the occupancy is asked of
static const void* tile_kernel(const TrayDeviceScene* s) {
#define TK(A, F) (s->light_filter ? reinterpret_cast<const void*>(k_path_tiles<A, F, TRAY_INTEGRATOR_PATH, true>) : reinterpret_cast<const void*>(k_path_tiles<A, F, TRAY_INTEGRATOR_PATH, false>))
#define TK_F(A) (s->feat == FEAT_NONE ? TK(A, FEAT_NONE) : s->feat == FEAT_MERL ? TK(A, FEAT_MERL) : s->feat == FEAT_SPEC ? TK(A, FEAT_SPEC) \
                 : s->feat == (FEAT_MERL | FEAT_SPEC) ? TK(A, FEAT_MERL | FEAT_SPEC) : s->feat == (FEAT_ALL | FEAT_TEX) ? TK(A, FEAT_ALL | FEAT_TEX) : TK(A, FEAT_ALL))
    if (s->dev.integrator == TRAY_INTEGRATOR_WHITTED)
        return s->animated ? reinterpret_cast<const void*>(k_path_tiles<1, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_WHITTED>)
                           : reinterpret_cast<const void*>(k_path_tiles<0, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_WHITTED>);
    return s->animated ? TK_F(1) : TK_F(0);
#undef TK_F
#undef TK
}

static int scene_build(const TrayFlatScene* f, TrayDeviceScene* donor, TrayDeviceScene** out) {
    if (!f || !out) { set_error("tray_scene_create: null argument"); return TRAY_E_INVALID; }
    *out = nullptr;
    if (f->abi_version != TRAY_ABI_VERSION) { set_error("tray_scene_create: ABI version mismatch"); return TRAY_E_INVALID; }
    if (f->n_lights == 0) { set_error("At least one light is required"); return TRAY_E_INVALID; }   // multithreaded.rs:39
    // (a frame update that keeps the device's trees never reads the new scene's BVH<Triangle> nodes: they are walked below only if it does not)
    if (const std::string bad = tray::validate_flat_scene(f, donor == nullptr); !bad.empty()) { set_error("tray_scene_create: inconsistent scene: " + bad); return TRAY_E_INVALID; }
    if (f->film.width % 8 != 0 || f->film.height % 8 != 0 || f->film.width == 0 || f->film.height == 0) {
        set_error("Image dimensions not evenly divided by blocks of (8, 8)");
        return TRAY_E_INVALID;
    }
    if (f->film.filter_pixel_w > 4 || f->film.filter_pixel_h > 4 || f->film.filter_pixel_w < 0 || f->film.filter_pixel_h < 0) {
        set_error("reconstruction filters wider than 2 px are not supported by the LDS film window");
        return TRAY_E_UNSUPPORTED;
    }
    if (f->film.filter_w > 2.0f || f->film.filter_h > 2.0f) {
        set_error("reconstruction filters wider than 2.0 are not supported by the LDS film window");
        return TRAY_E_UNSUPPORTED;
    }
    if (f->integrator > TRAY_INTEGRATOR_WHITTED) { set_error("unknown integrator"); return TRAY_E_INVALID; }
    if (f->integrator == TRAY_INTEGRATOR_WHITTED && f->max_depth > WH_MAX_DEPTH) { set_error("whitted recursion depth > 16 is not supported"); return TRAY_E_UNSUPPORTED; }
    if (f->integrator != TRAY_INTEGRATOR_WHITTED && f->max_depth > 15) { set_error("pathtracer max_depth > 15 is not supported"); return TRAY_E_UNSUPPORTED; }
    auto stack_ok = [&](uint32_t first, uint32_t count, bool moving) {   // spline stacks the device evaluates per ray
        if ((uint64_t)first + count > f->n_xf_levels) return false;
        for (uint32_t l = 0; moving && l < count; ++l) {
            const TrayXformLevel& lv = f->xf_levels[first + l];
            if (lv.kf_count < 2) continue;
            if (lv.degree > 3 || lv.knot_count != lv.kf_count + lv.degree + 1 || (uint64_t)lv.kf_first + lv.kf_count > f->n_keyframes ||
                (uint64_t)lv.knot_first + lv.knot_count > f->n_knots) return false;
        }
        return true;
    };
    if (!stack_ok(f->camera.xf_first, f->camera.xf_count, f->camera.animated != 0)) {
        set_error("camera keyframes: the device evaluates B-splines of degree <= 3 with consistent knot vectors"); return TRAY_E_UNSUPPORTED;
    }
    bool deforming = false;
    bool moving = f->camera.animated != 0;
    for (uint32_t i = 0; i < f->n_instances; ++i) {
        const TrayInstance& in = f->instances[i];
        if (!stack_ok(in.xf_first, in.xf_count, in.animated != 0)) {
            set_error("instance keyframes: the device evaluates B-splines of degree <= 3 with consistent knot vectors"); return TRAY_E_UNSUPPORTED;
        }
        if (in.emis_count && (uint64_t)in.emis_first + in.emis_count > f->n_color_keys) { set_error("instance references missing colour keys"); return TRAY_E_INVALID; }
        moving = moving || in.animated != 0 || in.emis_count >= 2;
        if (in.kind != TRAY_INST_POINT_EMITTER && in.material_id >= f->n_materials) { set_error("instance references a missing material"); return TRAY_E_INVALID; }
        if ((in.geom_type == TRAY_GEOM_MESH || in.geom_type == TRAY_GEOM_ANIMATED_MESH) && in.mesh_id >= f->n_meshes) { set_error("instance references a missing mesh"); return TRAY_E_INVALID; }
        if (in.geom_type == TRAY_GEOM_ANIMATED_MESH) { deforming = true; moving = true; }
    }
    TrayDeviceScene* s = new TrayDeviceScene();
    s->deforming = deforming;
    s->identity = scene_identity(f);
    s->device = donor ? donor->device : g_device;
    s->donor = donor;
    if (hipSetDevice(s->device) != hipSuccess) { delete s; set_error("hipSetDevice failed (is a GPU present?)"); return TRAY_E_DEVICE; }
    DevScene& d = s->dev;
    int rc = TRAY_OK;
    const TrayInstance* d_inst = nullptr;
#define UP_(field, hostptr, count, unchanged)                                         \
    if (rc == TRAY_OK) {                                                              \
        std::remove_cv_t<std::remove_pointer_t<decltype(hostptr)>> const* _p = nullptr; \
        rc = upload(s, #field, unchanged, hostptr, (size_t)(count), &_p);             \
        d.field = _p;                                                                 \
    }
#define UP(field, hostptr, count) UP_(field, hostptr, count, false)    /* may differ from frame to frame */
#define UPS(field, hostptr, count) UP_(field, hostptr, count, true)    /* part of the scene, the same at every frame */
    if (rc == TRAY_OK) rc = upload(s, "instances", false, f->instances, f->n_instances, &d_inst);
    d.instances = d_inst;
    s->d_instances = const_cast<TrayInstance*>(d_inst);
    // the trees in device order (host/gates.hpp: sibling pairs); the BVH<Triangle>s are part of the scene, a frame update keeps the donor's
    tray::PairedTrees paired;
    size_t n_paired = 0;   // nodes of the BVH<Triangle>s in device order: one more per tree
    for (uint32_t m = 0; m < f->n_meshes; ++m) n_paired += f->meshes[m].node_count ? f->meshes[m].node_count + 1u : 0u;
    bool keep_trees = false;
    const size_t top_quad_cap = 2u * (size_t)f->n_instances + 2u;   // records a BVH<Instance> of this scene can need (one per interior node + the entry)
    if (donor && donor->quad_first.size() == f->n_meshes) {
        bool have_pairs = false, have_quads = false;
        for (const TrayDevBuf& b : donor->bufs) {
            have_pairs = have_pairs || (std::strcmp(b.key, "mesh_nodes") == 0 && b.bytes == std::max<size_t>(n_paired, 1) * sizeof(TrayBvhNode));
            have_quads = have_quads || (std::strcmp(b.key, "quads") == 0 && b.bytes == (donor->n_mesh_quads + top_quad_cap) * sizeof(tray::QuadNode));
        }
        keep_trees = have_pairs && have_quads;
    }
    if (rc == TRAY_OK && donor && !keep_trees) {
        if (const std::string bad = tray::validate_mesh_trees(f); !bad.empty()) { rc = TRAY_E_INVALID; set_error("tray_scene_update_frame: inconsistent scene: " + bad); }
    }
    if (rc == TRAY_OK && !tray::pair_trees(f, paired, keep_trees)) { rc = TRAY_E_INVALID; set_error("BVH arrays do not describe trees"); }
    s->narrow_trees = keep_trees ? donor->narrow_trees : paired.narrow;
    s->paired_meshes = keep_trees ? donor->paired_meshes : paired.meshes;
    // the same trees as 128-byte records of two levels each, for the wavefront traversal (host/gates.hpp: QuadTrees)
    tray::QuadTrees quads;
    if (rc == TRAY_OK) {
        uint32_t bfs_levels = TRAY_QUAD_BFS_LEVELS;
        if (const char* e = getenv("TRAYHIP_QUAD_BFS")) bfs_levels = (uint32_t)std::max(0, atoi(e));
        tray::quad_trees(f, quads, keep_trees, bfs_levels);
        s->narrow_trees = s->narrow_trees && quads.narrow;
        s->ordered_boxes = (keep_trees ? donor->ordered_boxes : true) && quads.ordered;
        s->quad_first = keep_trees ? donor->quad_first : quads.mesh_first;
        s->n_mesh_quads = keep_trees ? donor->n_mesh_quads : quads.mesh.size();
        s->quad_mesh_pend = keep_trees ? donor->quad_mesh_pend : quads.mesh_pend;
        // per lane: node entries are two words (descriptor, entry distance); the instances of a BVH<Instance> leaf (<= 31) and the
        // exit-mesh sentinel one each; rounded up so that consecutive frames of a sequence keep the pool's overflow columns
        s->quad_stack_words = (2u * (quads.top_pend + s->quad_mesh_pend) + 32u + 2u + 15u) / 16u * 16u;
    }
    {   // the wavefront traversal's instance records (host/gates.hpp): per frame, as the instances are
        std::vector<tray::WfInst> recs;
        tray::wf_inst_records(f, s->quad_first, recs);
        UP(wf_insts, recs.data(), recs.size())
    }
    UP(top_nodes, paired.top.data(), paired.top.size())
    if (rc == TRAY_OK && (quads.top.size() > top_quad_cap || (s->n_mesh_quads + top_quad_cap) * sizeof(tray::QuadNode) >= ((size_t)1 << 32))) {
        rc = TRAY_E_UNSUPPORTED; set_error("the quad records of the scene's trees do not fit 32-bit offsets");
    }
    if (rc == TRAY_OK) {   // one buffer: the BVH<Triangle>s (kept across frames), then this frame's BVH<Instance>
        const tray::QuadNode* dq = nullptr;
        std::vector<tray::QuadNode> all;
        if (!keep_trees) {
            all = quads.mesh;
            all.insert(all.end(), quads.top.begin(), quads.top.end());
            all.resize(s->n_mesh_quads + top_quad_cap, tray::quad_empty_record());
        }
        rc = upload(s, "quads", keep_trees, keep_trees ? static_cast<const tray::QuadNode*>(nullptr) : all.data(), s->n_mesh_quads + top_quad_cap, &dq);
        if (rc == TRAY_OK && keep_trees && hipMemcpy(const_cast<tray::QuadNode*>(dq) + s->n_mesh_quads, quads.top.data(), quads.top.size() * sizeof(tray::QuadNode), hipMemcpyHostToDevice) != hipSuccess) {
            rc = TRAY_E_DEVICE; set_error("hipMemcpy of the BVH<Instance> records failed");
        }
        d.quads = reinterpret_cast<const float4*>(dq);
        d.top_quad_first = (uint32_t)s->n_mesh_quads;
    }
    {   // ray binning before the traversal stages (wavefront.h): the cells are those of the frame's BVH<Instance> root box
        s->bin_grid = f->n_top_nodes ? wf_bin_grid(f->top_nodes[0].bmin, f->top_nodes[0].bmax) : WfBinGrid{};
        s->wf_bin_stages = WF_BIN_DEFAULT;
        if (const char* e = getenv("TRAYHIP_WF_BIN")) s->wf_bin_stages = (uint32_t)std::max(0, atoi(e)) & 3u;
        s->wf_fused = WF_FUSED_DEFAULT != 0;
        if (const char* e = getenv("TRAYHIP_WF_FUSED")) s->wf_fused = atoi(e) != 0;
    }
    UP(top_order, f->top_order, f->n_top_order)
    UPS(meshes, keep_trees ? f->meshes : paired.meshes.data(), f->n_meshes)          // (kept: the donor's copies are not written)
    UPS(mesh_nodes, keep_trees ? f->mesh_nodes : paired.mesh.data(), n_paired)
    UPS(tri_verts, f->tri_verts, f->n_tris)
    UPS(tri_attrs, f->tri_attrs, f->n_tris)
    UPS(mesh_keys, f->mesh_keys, f->n_mesh_keys)      // (AnimatedMesh: keyframe counts and times; read by the ANIM = 3 kernels only)
    UPS(key_times, f->key_times, f->n_key_times)
    std::vector<DevMaterial> mats(f->n_materials);
    for (uint32_t i = 0; i < f->n_materials; ++i) {
        if (f->materials[i].kind == TRAY_MAT_MERL && f->materials[i].table >= f->n_merl) { rc = TRAY_E_INVALID; set_error("material references a missing MERL table"); }
        else mats[i] = lower_material(f->materials[i], f->merl_tables);
    }
    {   // smallest kernel feature set that covers the materials
        int feat = FEAT_NONE;
        for (const DevMaterial& dm : mats)
            for (uint32_t l = 0; l < dm.n_lobes && l < 2u; ++l) {
                const uint32_t k = dm.lobe[l].kind;
                if (k == LB_MERL) feat |= FEAT_MERL;
                if (k == LB_MF_TRANS) feat |= FEAT_MF_TRANS;
                if (k == LB_SPEC_REFL_DIEL || k == LB_SPEC_REFL_COND || k == LB_SPEC_TRANS || k == LB_TS_COND) feat |= FEAT_SPEC;
            }
        s->feat = (feat & FEAT_MF_TRANS) ? FEAT_ALL : feat;
        if (getenv("TRAYHIP_FEAT_ALL")) s->feat = FEAT_ALL;
        for (const DevMaterial& dm : mats)   // lobes of textured materials are only known per hit; GGX lives in the same instantiation
            if (dm.textured || dm.microfacet == TRAY_MF_GGX) s->feat = FEAT_ALL | FEAT_TEX;
        s->light_filter = (s->feat & FEAT_SPEC) != 0;
        for (uint32_t l = 0; l < f->n_lights; ++l)
            if (f->instances[f->lights[l]].kind != TRAY_INST_POINT_EMITTER && f->instances[f->lights[l]].geom_type == TRAY_GEOM_SPHERE) s->light_filter = true;
        if (getenv("TRAYHIP_NO_LIGHT_FILTER")) s->light_filter = false;
    }
    for (const DevMaterial& dm : mats) s->mat_kinds_present |= 1u << dm.mat_kind;
    if (f->n_textures) {
        UPS(textures, f->textures, f->n_textures)
        UPS(tex_frames, f->tex_frames, f->n_tex_frames)
        UPS(tex_data, f->tex_data, f->n_tex_bytes)
    }
    UP(materials, mats.data(), f->n_materials)
    UPS(merl_data, f->merl_data, f->n_merl_floats)
    UP(lights, f->lights, f->n_lights)
    UPS(filter_table, &f->film.table[0], TRAY_FILTER_TABLE_SIZE * TRAY_FILTER_TABLE_SIZE)
    UPS(filter_x, &f->film.table_x[0], TRAY_FILTER_TABLE_SIZE)
    UPS(filter_y, &f->film.table_y[0], TRAY_FILTER_TABLE_SIZE)
    UP(xf_levels, f->xf_levels, f->n_xf_levels)
    UP(keyframes, f->keyframes, f->n_keyframes)
    UP(knots, f->knots, f->n_knots)
    UP(color_keys, f->color_keys, f->n_color_keys)
    {   // the shuffles of the per-path LD arrays (path.rs:55-60: arrays of max_depth + 1 samples) come from a pool built once per scene
        std::vector<uint8_t> pool(TR_PERM_BYTES);
        perm_pool_build(f->max_depth + 1u, pool.data());
        UPS(perm_pool, pool.data(), pool.size())
    }
#undef UP
#undef UPS
#undef UP_
    for (uint32_t t = 0; t < f->n_textures; ++t) moving = moving || f->textures[t].n_frames >= 2u;   // animated_image: sampled at ray.time, which only the ANIM kernels carry
    s->animated = moving;
    if (rc == TRAY_OK) {   // the flat instance loop's records and gates (host/gates.hpp; dev_geom.h: trace_flat, mesh_leaf_coop)
        std::vector<tray::FlatLeaf> leaves;
        std::vector<tray::FlatInst> insts;
        std::vector<uint8_t> tri_leaf;
        tray::flat_loop_gates(f, paired, TR_COOP_MAX_TRIS, leaves, insts, tri_leaf);
        const tray::FlatLeaf* d_leaves = nullptr;
        const tray::FlatInst* d_insts = nullptr;
        const uint8_t* d_tri_leaf = nullptr;
        rc = upload(s, "flat_leaves", false, leaves.data(), leaves.size(), &d_leaves);
        if (rc == TRAY_OK) rc = upload(s, "flat_insts", false, insts.data(), insts.size(), &d_insts);
        if (rc == TRAY_OK) rc = upload(s, "tri_leaf", true, tri_leaf.data(), tri_leaf.size(), &d_tri_leaf);
        d.flat_leaves = d_leaves; d.flat_insts = d_insts; d.n_flat_leaves = (uint32_t)leaves.size(); d.tri_leaf = d_tri_leaf;
    }
    if (rc != TRAY_OK) { tray_scene_destroy(s); return rc; }
    s->n_materials = f->n_materials;
    d.n_instances = f->n_instances; d.n_lights = f->n_lights; d.min_depth = f->min_depth; d.max_depth = f->max_depth;
    d.width = f->film.width; d.height = f->film.height; d.frame = f->frame; d.integrator = f->integrator;
    {   // row-binned film needs: separable table, filter_h == 2 (class = eighth of a pixel), consistent factors
        bool ok = f->film.separable != 0 && f->film.filter_h == 2.0f && f->film.inv_h == 0.5f && f->film.filter_pixel_h == 4;
        for (int y = 0; ok && y < TRAY_FILTER_TABLE_SIZE; ++y)
            for (int x = 0; x < TRAY_FILTER_TABLE_SIZE; ++x)
                if (f->film.table[y * TRAY_FILTER_TABLE_SIZE + x] != f->film.table_x[x] * f->film.table_y[y]) { ok = false; break; }
        d.film_rows = (ok && !getenv("TRAYHIP_DIRECT_FILM")) ? 1u : 0u;
        // schedule: the tile megakernel for scenes its flat instance loop covers, the wavefront stage kernels (compacted ray
        // queues, persistent traversal with dynamic fetch) for scenes that go through BVH<Instance>; TRAYHIP_MODE overrides
        s->wavefront = f->n_instances > TR_FLAT_MAX;
        if (const char* m = getenv("TRAYHIP_MODE")) s->wavefront = std::string(m) == "wave";
        // a tree the quad-record traversal cannot take (a box with min > max or NaN -- BBox::new() is +inf / -inf in the reference --, a mesh beyond the
        // 23-bit descriptors) renders through the tile kernel's binary traversal, which is the reference's, instead of failing at render time
        if (s->wavefront && (!s->narrow_trees || !s->ordered_boxes) && !getenv("TRAYHIP_MODE")) s->wavefront = false;
        if (f->integrator == TRAY_INTEGRATOR_WHITTED) s->wavefront = false;   // the recursion runs inside the tile kernel only (dev_whitted.h)
        if (s->deforming) s->wavefront = false;   // (k_sampler_pass<3> renders these scenes: launch_tiles)
    }
    d.filter_w = f->film.filter_w; d.filter_h = f->film.filter_h; d.inv_w = f->film.inv_w; d.inv_h = f->film.inv_h;
    d.fpw = f->film.filter_pixel_w; d.fph = f->film.filter_pixel_h;
    {
        const TrayCamera* d_cam = nullptr;
        rc = upload(s, "camera", false, &f->camera, 1, &d_cam);
        d.camera_p = d_cam;
        s->camera_animated = f->camera.animated != 0;
        if (rc != TRAY_OK) { tray_scene_destroy(s); return rc; }
    }
    // Morton tile queue (BlockQueue::new)
    uint32_t n_tiles = 0;
    rc = tray_block_queue(d.width, d.height, 0, 0, nullptr, 0, &n_tiles);
    std::vector<uint32_t> xy(2 * (size_t)n_tiles);
    if (rc == TRAY_OK) rc = tray_block_queue(d.width, d.height, 0, 0, xy.data(), n_tiles, &n_tiles);
    const uint2* d_tiles = nullptr;
    if (rc == TRAY_OK) rc = upload(s, "tiles", true, reinterpret_cast<const uint2*>(xy.data()), n_tiles, &d_tiles);
    s->d_tiles = const_cast<uint2*>(d_tiles);
    s->n_tiles = n_tiles;
    const uint32_t* d_counter = nullptr;
    const DevStats* d_stats = nullptr;
    uint32_t zero = 0;
    std::vector<DevStats> zs(WF_STAT_SLOTS);
    std::memset(zs.data(), 0, zs.size() * sizeof(DevStats));
    if (rc == TRAY_OK) rc = upload(s, "counter", false, &zero, 1, &d_counter);
    if (rc == TRAY_OK) rc = upload(s, "stats", false, zs.data(), zs.size(), &d_stats);
    const uint32_t* d_retraced = nullptr;
    if (rc == TRAY_OK) rc = upload(s, "retraced", false, &zero, 1, &d_retraced);
    s->d_counter = const_cast<uint32_t*>(d_counter);
    s->d_stats = const_cast<DevStats*>(d_stats);
    s->d_retraced = const_cast<uint32_t*>(d_retraced);
    d.retraced = s->d_retraced;
    if (rc != TRAY_OK) { tray_scene_destroy(s); return rc; }
    if (donor && donor->ev0 && donor->ev1) { s->ev0 = donor->ev0; s->ev1 = donor->ev1; donor->ev0 = nullptr; donor->ev1 = nullptr; }
    else if (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) {
        tray_scene_destroy(s); set_error("hipEventCreate failed"); return TRAY_E_DEVICE;
    }
    {   // traversal stack depth: deepest node of any BVH<Triangle>; the one-loop two-level traversal (more than
        // TR_FLAT_MAX instances) also keeps the top-level path, the instances of a leaf and a sentinel
        auto depth_of = [](const TrayBvhNode* nodes, uint32_t n) {
            uint32_t best = 0;
            std::vector<std::pair<uint32_t, uint32_t>> st;   // node, depth
            if (n) st.push_back({0u, 1u});
            while (!st.empty()) {
                auto [idx, dep] = st.back();
                st.pop_back();
                best = std::max(best, dep);
                if (idx < n && nodes[idx].count == 0) { st.push_back({idx + 1, dep + 1}); st.push_back({nodes[idx].offset, dep + 1}); }
            }
            return best;
        };
        // (a frame update that keeps the device's trees keeps their depths: the walk over the 6.2 M nodes of the tr15 stand-in's meshes was 80 ms per frame)
        std::vector<uint32_t>& mesh_depths = s->mesh_depths;
        if (keep_trees && donor->mesh_depths.size() == f->n_meshes) mesh_depths = donor->mesh_depths;
        else {
            mesh_depths.assign(f->n_meshes, 0u);
            for (uint32_t m = 0; m < f->n_meshes; ++m) mesh_depths[m] = depth_of(f->mesh_nodes + f->meshes[m].node_offset, f->meshes[m].node_count);
        }
        uint32_t mesh_depth = 0;
        for (uint32_t m = 0; m < f->n_meshes; ++m) mesh_depth = std::max(mesh_depth, mesh_depths[m]);
        uint32_t depth = mesh_depth + 1;   // per-lane BVH<Triangle> traversal: one pending far child per level
        {   // (scenes the flat instance loop serves need it too: rays with tied candidates are re-traced through BVH<Instance>)
            // two-level traversal: exact worst case over the instances. While instance j of a BVH<Instance> leaf at depth d is
            // traversed the stack holds the pending far children of the top-level path (d - 1), the leaf's later instances,
            // the exit-mesh sentinel and the pending far children inside the mesh (its depth - 1).
            uint32_t worst = 0;
            std::vector<std::pair<uint32_t, uint32_t>> st;
            if (f->n_top_nodes) st.push_back({0u, 1u});
            while (!st.empty()) {
                auto [idx, dep] = st.back();
                st.pop_back();
                if (idx >= f->n_top_nodes) continue;
                const TrayBvhNode& nd = f->top_nodes[idx];
                if (nd.count == 0) { st.push_back({idx + 1, dep + 1}); st.push_back({nd.offset, dep + 1}); continue; }
                for (uint32_t j = 0; j < nd.count; ++j) {
                    uint32_t need = (dep - 1) + (nd.count - 1 - j);
                    if (nd.offset + j < f->n_top_order) {
                        const TrayInstance& in = f->instances[f->top_order[nd.offset + j]];
                        if ((in.geom_type == TRAY_GEOM_MESH || in.geom_type == TRAY_GEOM_ANIMATED_MESH) && in.mesh_id < f->n_meshes) need += 1 + (mesh_depths[in.mesh_id] > 0 ? mesh_depths[in.mesh_id] - 1 : 0);
                    }
                    worst = std::max(worst, need);
                }
                worst = std::max(worst, dep - 1 + nd.count);   // right after the leaf queued its instances
            }
            depth = std::max(depth, worst + 1);   // + 1 spare entry
        }
        depth = std::max(depth, 4u);
        if (depth > 96) { tray_scene_destroy(s); set_error("BVH too deep for the LDS traversal stack (" + std::to_string(depth) + " levels)"); return TRAY_E_UNSUPPORTED; }
        s->stack_bytes = depth * TR_BLOCK * (uint32_t)sizeof(uint32_t);
        if (getenv("TRAYHIP_STATS")) fprintf(stderr, "[trayhip] traversal stack: %u entries per lane (deepest BVH<Triangle> %u)\n", depth, mesh_depth);
        // cooperative test of small meshes (dev_geom.h: mesh_leaf_coop) in the flat instance loop: per-wave LDS behind the stacks
        bool single_leaf = false;
        for (uint32_t m = 0; m < f->n_meshes; ++m) single_leaf = single_leaf || f->meshes[m].tri_count <= TR_COOP_MAX_TRIS;
        if (single_leaf && !s->wavefront && f->n_instances <= TR_FLAT_MAX && !getenv("TRAYHIP_NO_COOP")) {   // (moving scenes too: the flat loop serves them since round 4)
            s->dev.coop_offset = depth * TR_BLOCK;
            s->stack_bytes += (TR_BLOCK / 64) * TR_COOP_WORDS * (uint32_t)sizeof(float);
        }
        {   // the tile kernel's film window: over the stacks for the row-binned film, behind everything else otherwise (k_path_tiles)
            const uint32_t win_bytes = 4u * WIN_PLANE * (uint32_t)sizeof(float);
            if (d.film_rows) { s->dev.win_offset = 0u; s->stack_bytes = std::max(s->stack_bytes, win_bytes); }
            else { s->dev.win_offset = s->stack_bytes / (uint32_t)sizeof(uint32_t); s->stack_bytes += win_bytes; }
        }
        if (s->stack_bytes > 32u * 1024u) {   // past the default dynamic-LDS window: raise the per-kernel limit (160 KB LDS per CU)
            const int bytes = (int)s->stack_bytes;
            const void* const traversing[] = {   // every kernel that is launched with s->stack_bytes (or its LDS part)
                reinterpret_cast<const void*>(k_path_tiles<0, FEAT_NONE>), reinterpret_cast<const void*>(k_path_tiles<0, FEAT_MERL>),
                reinterpret_cast<const void*>(k_path_tiles<0, FEAT_SPEC>), reinterpret_cast<const void*>(k_path_tiles<0, FEAT_MERL | FEAT_SPEC>),
                reinterpret_cast<const void*>(k_path_tiles<0, FEAT_ALL>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_ALL>),
                reinterpret_cast<const void*>(k_path_tiles<0, FEAT_ALL | FEAT_TEX>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_ALL | FEAT_TEX>),
                reinterpret_cast<const void*>(k_path_tiles<1, FEAT_NONE>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_MERL>),
                reinterpret_cast<const void*>(k_path_tiles<1, FEAT_SPEC>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_MERL | FEAT_SPEC>),
                reinterpret_cast<const void*>(k_path_tiles<0, FEAT_NONE, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<0, FEAT_MERL, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<0, FEAT_SPEC, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<0, FEAT_MERL | FEAT_SPEC, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<0, FEAT_ALL, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<0, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_PATH, true>),
                reinterpret_cast<const void*>(k_path_tiles<1, FEAT_NONE, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_MERL, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_SPEC, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_MERL | FEAT_SPEC, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_ALL, TRAY_INTEGRATOR_PATH, true>), reinterpret_cast<const void*>(k_path_tiles<1, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_PATH, true>),
                reinterpret_cast<const void*>(k_path_tiles<0, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_WHITTED>),
                reinterpret_cast<const void*>(k_path_tiles<1, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_WHITTED>),
                reinterpret_cast<const void*>(k_wf_trace_dyn<0, 0>), reinterpret_cast<const void*>(k_wf_trace_dyn<0, 1>),
                reinterpret_cast<const void*>(k_wf_trace_dyn<1, 0>), reinterpret_cast<const void*>(k_wf_trace_dyn<1, 1>),
                reinterpret_cast<const void*>(k_wf_trace_dyn<2, 0>), reinterpret_cast<const void*>(k_wf_trace_dyn<2, 1>),
                reinterpret_cast<const void*>(k_wf_trace_fallback<0, 0>), reinterpret_cast<const void*>(k_wf_trace_fallback<0, 1>),
                reinterpret_cast<const void*>(k_wf_trace_fallback<1, 0>), reinterpret_cast<const void*>(k_wf_trace_fallback<1, 1>),
                reinterpret_cast<const void*>(k_wf_trace_fallback<2, 0>), reinterpret_cast<const void*>(k_wf_trace_fallback<2, 1>),
                reinterpret_cast<const void*>(k_debug_intersect<0>), reinterpret_cast<const void*>(k_debug_intersect<2>),
                reinterpret_cast<const void*>(k_debug_sample_radiance<0>), reinterpret_cast<const void*>(k_debug_sample_radiance<2>),
                reinterpret_cast<const void*>(k_sampler_pass<0>), reinterpret_cast<const void*>(k_sampler_pass<2>), reinterpret_cast<const void*>(k_sampler_pass<3>),
                reinterpret_cast<const void*>(k_sampler_pass<0, FEAT_NONE>), reinterpret_cast<const void*>(k_sampler_pass<2, FEAT_NONE>), reinterpret_cast<const void*>(k_sampler_pass<3, FEAT_NONE>),
                reinterpret_cast<const void*>(k_debug_intersect<3>), reinterpret_cast<const void*>(k_debug_sample_radiance<3>)};
            for (const void* k : traversing) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            (void)hipGetLastError();
        }
    }
    int per_cu = 0, cus = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, s->device) == hipSuccess) cus = prop.multiProcessorCount;
    hipError_t occ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tile_kernel(s), TR_BLOCK, s->stack_bytes);   // of the instantiation launch_tiles will run
    if (occ != hipSuccess || per_cu < 1) per_cu = 1;
    s->deferred_n_moving = 0;
    for (uint32_t i = 0; i < f->n_instances; ++i) if (f->instances[i].animated) s->deferred_n_moving++;
    {   // how many records per time index a later frame of this scene can need (xf_table_prepare sizes the table's buffer once)
        auto movable = [&](uint32_t first, uint32_t count) {
            for (uint32_t l = first; l < first + count && l < f->n_xf_levels; ++l) if (f->xf_levels[l].kf_count > 1u) return true;
            return false;
        };
        s->xf_movable = movable(f->camera.xf_first, f->camera.xf_count) ? 1u : 0u;
        for (uint32_t i = 0; i < f->n_instances; ++i) if (movable(f->instances[i].xf_first, f->instances[i].xf_count)) s->xf_movable++;
    }
    if (cus < 1) cus = 256;
    s->n_blocks = cus * per_cu;
    if (getenv("TRAYHIP_STATS")) fprintf(stderr, "[trayhip] tile kernel: %d workgroups per CU (dynamic LDS %u B)\n", per_cu, s->stack_bytes);
    if (s->deferred_n_moving > 64) {
        tray_scene_destroy(s); set_error("more than 64 instances move within one frame: the per-path transform cache does not cover that"); return TRAY_E_UNSUPPORTED;
    }
    if (s->animated && s->deferred_n_moving > 0) {   // per-path transform cache (dev_geom.h)
        std::vector<uint32_t> ids(s->deferred_n_moving, 0u);
        for (uint32_t i = 0; i < f->n_instances; ++i)
            if (f->instances[i].animated && f->instances[i].moving_slot < ids.size()) ids[f->instances[i].moving_slot] = i;
        const uint32_t* d_ids = nullptr;
        if (upload(s, "moving_ids", false, ids.data(), ids.size(), &d_ids) != TRAY_OK) { tray_scene_destroy(s); return TRAY_E_NOMEM; }
        // one column per pool slot / per thread; a frame update keeps the previous frame's pool size (the budget of wf_slot_count is a
        // share of the memory that was free BEFORE the pool existed)
        const bool keep_lanes = donor && donor->dev.xf_cache_lanes != 0u && donor->wavefront == s->wavefront && (s->wavefront || donor->n_blocks == s->n_blocks);
        const size_t reclaimable = donor ? donor->xf_cache_bytes + (donor->pool.data ? (size_t)F_COUNT * donor->pool.n_slots * sizeof(float) : 0) : 0;
        uint32_t lanes = keep_lanes ? donor->dev.xf_cache_lanes : (s->wavefront ? wf_slot_count(s, reclaimable) : (uint32_t)s->n_blocks * TR_BLOCK);
        const uint32_t n_moving_for_msg = s->deferred_n_moving;
        void* cache = nullptr;
        size_t cache_bytes = (size_t)s->deferred_n_moving * (s->wavefront ? TR_XF_REC : TR_XF_WORDS) * lanes * sizeof(float);   // (records per path / rows of columns: dev_anim.h)
        // (the wavefront schedule's pool follows the cache: if the budget of wf_slot_count -- a share of what hipMemGetInfo calls free -- cannot be
        // had in one piece, halve the pool rather than fail; the tile kernel's cache has one column per resident thread and cannot shrink)
        // Round 5: the wavefront schedule's cache (112 B per pool slot and moving instance: 38.5 GB for the tr15 stand-in's 11 at 33 M slots) is
        // allocated by the first launch that needs it (xf_cache_ensure): launches of many samples index the frame's transform table instead
        // (xf_table_prepare) and never touch it. A frame update still takes the previous frame's cache over if there is one.
        const bool lazy = s->wavefront;
        if (keep_lanes && donor->dev.xf_cache && donor->xf_cache_bytes >= cache_bytes) {   // (every path fills its columns before it reads them)
            cache = donor->dev.xf_cache;
            s->xf_cache_bytes = donor->xf_cache_bytes;
            forget_alloc(donor, cache);
            donor->dev.xf_cache = nullptr; donor->xf_cache_bytes = 0;
        } else if (lazy) {
            s->xf_cache_bytes = 0;
        } else if (hipMalloc(&cache, cache_bytes) == hipSuccess) {
            s->xf_cache_bytes = cache_bytes;
        } else {
            tray_scene_destroy(s);
            set_error("hipMalloc of the per-path transform cache failed: " + std::to_string(cache_bytes >> 20) + " MiB for " + std::to_string(n_moving_for_msg) +
                      " moving instances x " + std::to_string(lanes) + " paths (TRAYHIP_WF_SLOTS / TRAYHIP_XF_CACHE_BYTES bound it)");
            return TRAY_E_NOMEM;
        }
        if (cache) s->allocs.push_back(cache);
        s->dev.xf_cache = static_cast<float*>(cache);
        s->dev.moving_ids = d_ids;
        s->dev.n_moving = s->deferred_n_moving;
        s->dev.xf_stride = s->deferred_n_moving;
        s->dev.xf_cache_lanes = lanes;
        s->dev.xf_aos = s->wavefront ? 1u : 0u;
    }
    s->wf_req_slots = donor ? donor->wf_req_slots : 0u; s->wf_req_views = donor ? donor->wf_req_views : 0u; s->wf_req_slices = donor ? donor->wf_req_slices : 0u;
    s->xf_table_req = donor ? donor->xf_table_req : -1;
    if (donor && donor->d_xf_table) {
        // the previous frame's transform table serves as the buffer of this frame's, built anew by the first launch (xf_table_prepare checks that this
        // frame's records fit: freeing and allocating 20 GB costs 4 s -- profiles/r06_d_frame_overheads.txt --, so the buffer is sized once for a sequence)
        forget_alloc(donor, donor->d_xf_table); s->allocs.push_back(donor->d_xf_table);
        s->d_xf_table = donor->d_xf_table; s->xf_table_cap = donor->xf_table_cap; s->xf_table_stride = 0u; s->xf_table_built = false;
        donor->d_xf_table = nullptr; donor->xf_table_cap = 0u;
    }
    if (donor && donor->wf_ready && s->wavefront && donor->stack_bytes == s->stack_bytes && donor->quad_stack_words == s->quad_stack_words && donor->animated == s->animated &&
        (donor->pool.n_slots <= ((s->animated && s->dev.xf_cache_lanes) ? s->dev.xf_cache_lanes : wf_slot_wish(s)) ||   // (a pool that came out smaller than wished -- memory -- stays as it is)
         (donor->last_used_table && s->d_xf_table))) {   // (a pool sized for launches that read the frame's transform table -- no per-path cache bounds it --: this frame's launches read theirs, xf_table_prepare)
        // the wavefront schedule's pool, queues, chunk records and row bins (2.2 GB at 8 M slots) serve the next frame as they are:
        // launch_wavefront re-initialises the chunk records and the control words of every launch, the bins are zero between tiles
        for (void* p : {(void*)donor->pool.data, (void*)donor->d_chunks, (void*)donor->d_bins, (void*)donor->d_wf_counters, (void*)donor->d_queues,
                        (void*)donor->d_kind_queues, (void*)donor->d_stack_overflow, (void*)donor->d_fallback, (void*)donor->d_bin_ctl})
            if (p) { forget_alloc(donor, p); s->allocs.push_back(p); s->wf_allocs.push_back(p); }
        donor->wf_allocs.clear(); s->wf_shrunk = donor->wf_shrunk;
        s->pool = donor->pool; s->d_chunks = donor->d_chunks; s->d_bins = donor->d_bins; s->d_wf_counters = donor->d_wf_counters;
        s->d_queues = donor->d_queues; s->d_kind_queues = donor->d_kind_queues; s->d_stack_overflow = donor->d_stack_overflow; s->d_fallback = donor->d_fallback;
        s->d_bin_ctl = donor->d_bin_ctl;
        s->h_done = donor->h_done; donor->h_done = nullptr;
        s->n_chunks = donor->n_chunks; s->n_blocks_trace = donor->n_blocks_trace; s->trace_lds_depth = donor->trace_lds_depth;
        s->trace_lds_bytes = donor->trace_lds_bytes; s->wf_sort = donor->wf_sort; s->ovf_entries = donor->ovf_entries;
        s->wf_ready = true;
        donor->wf_ready = false; donor->pool.data = nullptr;
    }
    if (donor && getenv("TRAYHIP_STATS"))
        fprintf(stderr, "[trayhip] frame update: pool %s (donor ready %d, wavefront %d, stack %u / %u B, quad stack %u / %u, animated %d / %d, donor slots %u, cache lanes %u, donor used the table %d, table buffer %s)\n",
                s->wf_ready ? "kept" : "not kept", donor->wf_ready || s->wf_ready ? 1 : 0, s->wavefront ? 1 : 0, donor->stack_bytes, s->stack_bytes, donor->quad_stack_words, s->quad_stack_words,
                donor->animated ? 1 : 0, s->animated ? 1 : 0, s->pool.n_slots ? s->pool.n_slots : donor->pool.n_slots, s->dev.xf_cache_lanes, donor->last_used_table ? 1 : 0, s->d_xf_table ? "kept" : "none");
    s->donor = nullptr;
    *out = s;
    return TRAY_OK;
}

extern "C" {

int tray_scene_create(const TrayFlatScene* f, TrayDeviceScene** out) { return scene_build(f, nullptr, out); }

// Scene::update_frame (scene.rs:152-176) for the device copy: the frame loop of main.rs:91-106 keeps the scene and rebuilds the
// instance transforms and BVH<Instance> per frame. `f` is the SAME scene flattened at another frame: meshes, MERL tables, textures,
// materials' tables and the tile queue are kept on the device (buffers whose size is unchanged are taken over without a copy),
// instances, BVH<Instance>, flat-loop records, camera, spline tables, colour keys and the moving set are uploaded anew, and the
// wavefront pool / queues and the per-path transform cache are carried over. Nothing else is assumed: every decision of
// tray_scene_create (schedule, kernel instantiation, stack depth, occupancy) is taken again for the new frame.
int tray_scene_update_frame(TrayDeviceScene* s, const TrayFlatScene* f) {
    if (!s || !f) { set_error("tray_scene_update_frame: null argument"); return TRAY_E_INVALID; }
    if (f->film.width != s->dev.width || f->film.height != s->dev.height) { set_error("tray_scene_update_frame: the film size changed: create a new device scene"); return TRAY_E_INVALID; }
    // nothing has moved yet: a scene that is not the one this copy was created from is refused and the handle stays usable
    if (!(scene_identity(f) == s->identity)) {
        set_error("tray_scene_update_frame: not the scene this device copy was created from (mesh / triangle / material / table counts, depths, integrator or filter differ)");
        return TRAY_E_INVALID;
    }
    HIP_CHECK(hipSetDevice(s->device));
    HIP_CHECK(hipDeviceSynchronize());   // (launches of the previous frame read the buffers that are about to be overwritten)
    TrayDeviceScene* n = nullptr;
    const int before = g_device;
    g_device = s->device;
    const size_t owned_before = s->allocs.size();
    const int rc = scene_build(f, s, &n);
    g_device = before;
    if (rc != TRAY_OK) {   // if a buffer has moved to the half-built (and now destroyed) new state the handle can only be destroyed; a refusal before that leaves it as it was
        s->donor = nullptr;
        if (s->allocs.size() != owned_before) s->broken = true;
        return rc;
    }
    n->sampler_kind = s->sampler_kind; n->smp_min = s->smp_min; n->smp_max = s->smp_max;   // (tray_scene_set_sampler belongs to the handle)
    std::swap(n->d_smp, s->d_smp); std::swap(n->smp_bytes, s->smp_bytes);
    std::swap(*s, *n);        // the handle keeps its identity; n now owns what the new frame did not take over
    tray_scene_destroy(n);
    return TRAY_OK;
}

static int launch_tiles(TrayDeviceScene* s, uint32_t tile_start, uint32_t tile_count, uint32_t chunk, uint32_t chunk_stride,
                        uint32_t spp, uint64_t seed, float* rgbw_dev, void* stream_);

// every buffer of the wavefront schedule for a pool of n_slots; on failure nothing stays allocated and TRAY_E_NOMEM is returned
static void wf_free(TrayDeviceScene* s) {
    for (void* p : s->wf_allocs) { forget_alloc(s, p); (void)hipFree(p); }
    s->wf_allocs.clear();
    if (s->h_done) { (void)hipHostFree(s->h_done); s->h_done = nullptr; }
    s->pool.data = nullptr; s->pool.n_slots = 0; s->d_chunks = nullptr; s->d_bins = nullptr; s->d_wf_counters = nullptr; s->d_queues = nullptr;
    s->d_kind_queues = nullptr; s->d_stack_overflow = nullptr; s->d_fallback = nullptr; s->d_bin_ctl = nullptr; s->n_chunks = 0; s->wf_ready = false;
}
static int wf_alloc(TrayDeviceScene* s, uint32_t n_slots) {
    auto grab = [&](size_t bytes, void** out) -> bool {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("hipMalloc of " + std::to_string(bytes >> 20) + " MiB failed: " + hipGetErrorString(e));
            wf_free(s);
            return false;
        }
        s->allocs.push_back(p); s->wf_allocs.push_back(p);
        *out = p;
        return true;
    };
    void* p = nullptr;
    s->n_chunks = n_slots / TR_BLOCK;
    if (!grab((size_t)F_COUNT * n_slots * sizeof(float), &p)) return TRAY_E_NOMEM;
    s->pool.data = static_cast<float*>(p); s->pool.n_slots = n_slots; s->pool.seg_cap = wf_seg_cap(s->n_chunks);
    const size_t q_cap = (size_t)WF_SEGS * s->pool.seg_cap;   // entries of one queue: WF_SEGS segments (wavefront.h)
    const uint32_t n_chunks = s->n_chunks;                    // (wf_free resets the scene's fields on a failed grab)
    if (!grab((size_t)n_chunks * sizeof(WfChunk), &p)) return TRAY_E_NOMEM;
    s->d_chunks = static_cast<WfChunk*>(p);
    if (!grab((size_t)n_chunks * ROWBIN_SIZE * sizeof(float), &p)) return TRAY_E_NOMEM;
    s->d_bins = static_cast<float*>(p);
    if (hipMemset(s->d_bins, 0, (size_t)n_chunks * ROWBIN_SIZE * sizeof(float)) != hipSuccess) { set_error("hipMemset of the row bins failed"); wf_free(s); return TRAY_E_DEVICE; }
    if (!grab(2 * sizeof(uint32_t), &p)) return TRAY_E_NOMEM;
    s->d_wf_counters = static_cast<uint32_t*>(p);
    // ray queues A, B, C, regeneration queue and the control words of their segments, for up to WF_PIPES_MAX views (a view's segments
    // are sized for its own chunks: WF_SEGS * TR_BLOCK entries of rounding per view and queue)
    const size_t q_slack = (size_t)WF_PIPES_MAX * WF_SEGS * TR_BLOCK;
    // (an entry of a ray queue is the ray: WF_RAY_WORDS words; the regeneration queue holds slot indices)
    if (!grab(((3 * WF_RAY_WORDS + 1) * (q_cap + q_slack) + (size_t)WF_PIPES_MAX * WF_QCTL_WORDS) * sizeof(uint32_t), &p)) return TRAY_E_NOMEM;
    s->d_queues = static_cast<uint32_t*>(p);
    // ray binning: per view and stage (A, B) the histogram and the cursors of every segment's bins
    if (!grab((size_t)WF_PIPES_MAX * 2u * 2u * WF_SEGS * WF_BINS * sizeof(uint32_t), &p)) return TRAY_E_NOMEM;
    s->d_bin_ctl = static_cast<uint32_t*>(p);
    s->wf_sort = !(s->feat & FEAT_TEX);   // the kind-pure kernels read lobes from the material table; textured materials have theirs per hit
    if (s->wf_sort) {   // shading queues of the material sort (slot indices), one per material kind
        if (!grab((size_t)WF_MAT_KINDS * (q_cap + q_slack) * sizeof(uint32_t), &p)) return TRAY_E_NOMEM;
        s->d_kind_queues = static_cast<uint32_t*>(p);
    }
    {
        int per_cu = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        // LDS stack entries per lane such that WF_TRACE_WAVES workgroups (4 waves each = one wave per SIMD) fit in the CU's 160 KB
        // (a node on this kernel's stack is two words: descriptor and entry distance; up to three per expanded record)
        const uint32_t full_depth = std::max(s->quad_stack_words, 8u);
        uint32_t lds_depth = std::min<uint32_t>(full_depth, (160u * 1024u / WF_TRACE_WAVES) / (TR_BLOCK * (uint32_t)sizeof(uint32_t)));
        if (const char* e = getenv("TRAYHIP_WF_LDS_DEPTH")) lds_depth = std::min<uint32_t>(full_depth, (uint32_t)std::max(1, atoi(e)));
        s->trace_lds_depth = lds_depth;
        s->trace_lds_bytes = lds_depth * TR_BLOCK * (uint32_t)sizeof(uint32_t);
        hipError_t oe = s->animated ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_trace_dyn<0, 1>, TR_BLOCK, s->trace_lds_bytes)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_trace_dyn<0, 0>, TR_BLOCK, s->trace_lds_bytes);
        if (oe != hipSuccess || per_cu < 1) per_cu = 1;
        s->n_blocks_trace = (uint32_t)(cus * per_cu);
        const size_t ovf_entries = (size_t)(full_depth - lds_depth + 1u) * s->n_blocks_trace * TR_BLOCK;   // per view: their traversal kernels overlap
        s->ovf_entries = ovf_entries;
        if (!grab((size_t)WF_PIPES_MAX * ovf_entries * sizeof(uint32_t), &p)) return TRAY_E_NOMEM;
        s->d_stack_overflow = static_cast<uint32_t*>(p);
        if (getenv("TRAYHIP_STATS")) fprintf(stderr, "[trayhip] dynamic-fetch traversal: %u of %u stack entries in LDS, %d workgroups per CU\n", lds_depth, full_depth, per_cu);
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&s->h_done), sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("hipHostMalloc failed"); wf_free(s); return TRAY_E_NOMEM; }
    s->wf_ready = true;
    return TRAY_OK;
}

// Wavefront schedule: rounds of six stage kernels over the path pool until every tile is done.
// The host only polls a "tiles done" word every WF_POLL rounds; kernels of finished chunks exit at once.
static int launch_wavefront(TrayDeviceScene* s, uint32_t tile_start, uint32_t tile_count, uint32_t chunk, uint32_t chunk_stride,
                            uint32_t spp, uint32_t kf, float* rgbw_dev, hipStream_t stream) {
    if (!s->narrow_trees) { set_error("wavefront schedule: a BVH of more than 8 388 607 nodes or triangles (the traversal keeps a node as a 32-bit descriptor)"); return TRAY_E_UNSUPPORTED; }
    if (!s->ordered_boxes) { set_error("wavefront schedule: a BVH box with min > max (or NaN) on some axis; TRAYHIP_MODE=mega renders such a scene with the tile kernel"); return TRAY_E_UNSUPPORTED; }
    if (!s->wf_ready) {
        // the pool and its queues: as many slots as wf_slot_count grants; if hipMalloc still refuses (fragmentation, another scene's pool
        // allocated since), everything allocated so far is freed and half the slots are tried, down to 64 chunks -- then TRAY_E_NOMEM,
        // with the handle left as it was
        // (the per-path transform cache was budgeted at creation; a launch that reads the frame's table never allocates it: xf_table_prepare ran before this)
        uint32_t n_slots = s->last_used_table ? wf_slot_count(s, 0, false)
                         : (s->animated && s->dev.xf_cache_lanes) ? std::min(s->dev.xf_cache_lanes, wf_slot_wish(s)) : wf_slot_count(s);
        for (;;) {
            const int rc = wf_alloc(s, n_slots);
            if (rc == TRAY_OK) break;
            if (rc != TRAY_E_NOMEM) return rc;
            if (n_slots / 2u < 64u * TR_BLOCK) {
                set_error("the wavefront schedule's buffers could not be allocated even for " + std::to_string(n_slots) + " pool slots (" +
                          std::to_string((n_slots * wf_bytes_per_slot()) >> 20) + " MiB): " + tray_last_error());
                return TRAY_E_NOMEM;
            }
            n_slots = n_slots / 2u / TR_BLOCK * TR_BLOCK;
            s->wf_shrunk = true;
        }
    }
    // tiles are cut into slices of their samples while the pool has at least as many chunks as the launch then has work items
    // (k_wf_advance; a slice costs its own film resolve: at 8 M slots and 32 400 tiles halving them measured 124 against 132 Msamples/s); a
    // slice keeps at least 16 samples per pixel (TRAYHIP_WF_SLICES overrides: 1, 2, 4)
    uint32_t slice_shift = 0u;
    while ((1u << (slice_shift + 1u)) <= WF_MAX_SLICES && ((uint64_t)tile_count << (slice_shift + 1u)) <= s->n_chunks && (spp >> (slice_shift + 1u)) >= 16u) ++slice_shift;   // (cut while the items still fit the chunks: one item per chunk is the optimum)
    if (s->wf_req_slices) { slice_shift = 0u; while ((2u << slice_shift) <= s->wf_req_slices && (2u << slice_shift) <= WF_MAX_SLICES && (spp >> (slice_shift + 1u)) >= 1u) ++slice_shift; }   // tray_scene_set_wavefront
    if (const char* e = getenv("TRAYHIP_WF_SLICES")) { slice_shift = 0u; while ((2u << slice_shift) <= (uint32_t)std::max(1, atoi(e)) && (2u << slice_shift) <= WF_MAX_SLICES && (spp >> (slice_shift + 1u)) >= 1u) ++slice_shift; }
    tile_count <<= slice_shift;   // from here on: work items
    const uint32_t n_chunks = std::min(s->n_chunks, tile_count);
    HIP_CHECK(hipMemsetAsync(s->d_wf_counters, 0, 2 * sizeof(uint32_t), stream));
        {   // chunks start in WF_TILE_NEED with done = 0
        std::vector<WfChunk> init(n_chunks, WfChunk{WF_TILE_NEED, 0u});
        HIP_CHECK(hipMemcpyAsync(s->d_chunks, init.data(), (size_t)n_chunks * sizeof(WfChunk), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));   // `init` is pageable host memory
    }
    HIP_CHECK(hipEventRecord(s->ev0, stream));
    const uint2* tiles = s->d_tiles + tile_start;
    uint32_t launches = 0;
    // the views: equal shares of the chunks in use. Measured on the C5 stand-in at full detail (1 / 2 / 3 / 4 views): see DESIGN.md section 4
    // (with rounds of 24 M slots and more one view is ahead: C5 stand-in 155.8 against 152.9 Msamples/s at 32 M; at 16 M two views 146.8 against 144.3)
    uint32_t n_views = n_chunks >= (24u << 20) / TR_BLOCK ? 1u : 2u;
    if (s->wf_req_views) n_views = std::min<uint32_t>(WF_PIPES_MAX, s->wf_req_views);   // tray_scene_set_wavefront
    if (const char* e = getenv("TRAYHIP_WF_PIPES")) n_views = (uint32_t)std::max(1, std::min(WF_PIPES_MAX, atoi(e)));
    n_views = std::max(1u, std::min(n_views, n_chunks / WF_SEGS));   // (a view of a few chunks would only add launches)
    s->last_views = n_views; s->last_slices = 1u << slice_shift; s->last_was_wavefront = true;
    WfView views[WF_PIPES_MAX];
    {
        const size_t q_total = (size_t)WF_SEGS * s->pool.seg_cap + (size_t)WF_PIPES_MAX * WF_SEGS * TR_BLOCK;   // entries of one queue kind over all views
        uint32_t* const qbase[4] = {s->d_queues, s->d_queues + WF_RAY_WORDS * q_total, s->d_queues + 2 * WF_RAY_WORDS * q_total, s->d_queues + 3 * WF_RAY_WORDS * q_total};
        uint32_t* const qctl_base = qbase[3] + q_total;
        size_t q_off = 0;
        uint32_t c0 = 0;
        for (uint32_t k = 0; k < n_views; ++k) {
            const uint32_t c1 = (uint32_t)((uint64_t)n_chunks * (k + 1u) / n_views);
            WfView& v = views[k];
            v.n_chunks = c1 - c0;
            v.dev = s->launch_dev;
            if (v.dev.xf_cache && !v.dev.xf_table) v.dev.xf_cache += (size_t)c0 * TR_BLOCK * v.dev.n_moving * TR_XF_REC;   // [slot][moving instance][TR_XF_REC] (the table is indexed by time, not by slot)
            v.pool = s->pool;
            v.pool.first = c0 * TR_BLOCK;   // (the hit records are slot-major, the other fields field-major: the accessors add the view's first slot)
            v.pool.seg_cap = wf_seg_cap(v.n_chunks);
            v.chunks = s->d_chunks + c0;
            v.bins = s->d_bins + (size_t)c0 * ROWBIN_SIZE;
            v.qa = qbase[0] + WF_RAY_WORDS * q_off; v.qb = qbase[1] + WF_RAY_WORDS * q_off; v.qc = qbase[2] + WF_RAY_WORDS * q_off; v.qr = qbase[3] + q_off;
            v.qctl = qctl_base + (size_t)k * WF_QCTL_WORDS;
            v.kq = s->wf_sort ? s->d_kind_queues + (size_t)WF_MAT_KINDS * q_off : nullptr;
            v.overflow = s->d_stack_overflow + (size_t)k * s->ovf_entries;
            v.bin_ctl = s->d_bin_ctl ? s->d_bin_ctl + (size_t)k * 2u * 2u * WF_SEGS * WF_BINS : nullptr;
            v.stream = stream;
            if (k > 0) {
                if (!s->wf_streams[k]) HIP_CHECK(hipStreamCreateWithFlags(&s->wf_streams[k], hipStreamNonBlocking));
                if (!s->wf_join[k]) HIP_CHECK(hipEventCreateWithFlags(&s->wf_join[k], hipEventDisableTiming));
                v.stream = s->wf_streams[k];
            }
            q_off += (size_t)WF_SEGS * v.pool.seg_cap;
            c0 = c1;
        }
        if (n_views > 1u) {   // the other views start after what this call has put on the caller's stream so far
            if (!s->wf_fork) HIP_CHECK(hipEventCreateWithFlags(&s->wf_fork, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(s->wf_fork, stream));
            for (uint32_t k = 1; k < n_views; ++k) HIP_CHECK(hipStreamWaitEvent(views[k].stream, s->wf_fork, 0));
        }
    }
    // every chunk needs at most (spp/4 rounded up) samples x (max_depth + 2) rounds per tile, plus one round per tile switch
    const uint64_t tiles_per_chunk = (tile_count + n_chunks - 1) / n_chunks;
    const uint64_t max_rounds = tiles_per_chunk * (((uint64_t)(spp >> slice_shift) + 3) / 4 * ((WF_FOLD_C ? 2u : 1u) * s->dev.max_depth + 3) + 4) + 2 * WF_POLL;   // (WF_FOLD_C: a vertex with a stage C ray takes two rounds)
    bool done = false;
    for (uint32_t k = 0; k < n_views; ++k) HIP_CHECK(hipMemsetAsync(views[k].qctl, 0, WF_QCTL_WORDS * sizeof(uint32_t), views[k].stream));   // (later: inside wf_round)
    for (uint32_t round = 0; !done; ++round) {
        for (uint32_t k = 0; k < n_views; ++k) {
            const WfView& v = views[k];
            if (v.bin_ctl && s->wf_bin_stages) HIP_CHECK(hipMemsetAsync(v.bin_ctl, 0, (size_t)2u * 2u * WF_SEGS * WF_BINS * sizeof(uint32_t), v.stream));
            if (s->animated) wf_round<1>(s, v, tiles, tile_count, chunk, chunk_stride, spp, kf, rgbw_dev, slice_shift);
            else wf_round<0>(s, v, tiles, tile_count, chunk, chunk_stride, spp, kf, rgbw_dev, slice_shift);
            launches += (WF_FOLD_C ? 8 : 10) + ((s->wf_bin_stages & 1u) ? 2 : 0) + ((s->wf_bin_stages & 2u) ? 2 : 0);   // (nominal: a kind-pure launch per material kind present comes on top in both forms)
        }
        if (round % WF_POLL == WF_POLL - 1) {
            HIP_CHECK(hipGetLastError());
            for (uint32_t k = 1; k < n_views; ++k) HIP_CHECK(hipStreamSynchronize(views[k].stream));
            HIP_CHECK(hipMemcpyAsync(s->h_done, s->d_wf_counters + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            done = *s->h_done >= tile_count;
        }
        if (round > max_rounds) { set_error("wavefront schedule did not terminate"); return TRAY_E_DEVICE; }
    }
    for (uint32_t k = 1; k < n_views; ++k) {   // (the polls above have synchronised them; this keeps the caller's stream ordered after them in any case)
        HIP_CHECK(hipEventRecord(s->wf_join[k], views[k].stream));
        HIP_CHECK(hipStreamWaitEvent(stream, s->wf_join[k], 0));
    }
    HIP_CHECK(hipEventRecord(s->ev1, stream));
    s->timing_valid = true;
    s->launches = launches;
    return TRAY_OK;
}

int tray_scene_set_sampler(TrayDeviceScene* s, uint32_t kind, uint32_t min_spp, uint32_t max_spp) {
    if (!s) { set_error("tray_scene_set_sampler: null argument"); return TRAY_E_INVALID; }
    if (kind > TRAY_SAMPLER_ADAPTIVE) { set_error("tray_scene_set_sampler: unknown sampler kind"); return TRAY_E_INVALID; }
    if (kind == TRAY_SAMPLER_ADAPTIVE) {
        if (min_spp > (1u << 16) || max_spp > (1u << 16)) { set_error("tray_scene_set_sampler: Adaptive takes at most 65 536 samples per pixel here"); return TRAY_E_UNSUPPORTED; }
        const uint32_t lo = tray_round_spp(min_spp), hi = tray_round_spp(max_spp);   // adaptive.rs:36-47
        if (hi < lo) { set_error("tray_scene_set_sampler: max_spp < min_spp (Adaptive::new would underflow, adaptive.rs:48)"); return TRAY_E_INVALID; }
        s->smp_min = lo; s->smp_max = hi;
    } else { s->smp_min = s->smp_max = 1u; }
    s->sampler_kind = kind;
    return TRAY_OK;
}

int tray_render_tiles_device(TrayDeviceScene* s, uint32_t tile_start, uint32_t tile_count, uint32_t spp, uint64_t seed,
                             float* rgbw_dev, void* stream_) {
    if (!s || !rgbw_dev) { set_error("tray_render_tiles_device: null argument"); return TRAY_E_INVALID; }
    if (tile_count == 0) { tile_start = 0; tile_count = s->n_tiles; }          // BlockQueue::new ignores `start` when count == 0 (block_queue.rs:39-41), like tray_block_queue
    if (tile_start > s->n_tiles) tile_start = s->n_tiles;                       // skip(start).take(count)
    if (tile_count > s->n_tiles - tile_start) tile_count = s->n_tiles - tile_start;
    return launch_tiles(s, tile_start, tile_count, tile_count ? tile_count : 1, 1, spp, seed, rgbw_dev, stream_);
}

int tray_render_shard_device(TrayDeviceScene* s, uint32_t shard, uint32_t n_shards, uint32_t chunk_tiles, uint32_t spp, uint64_t seed,
                             float* rgbw_dev, void* stream_) {
    if (!s || !rgbw_dev) { set_error("tray_render_shard_device: null argument"); return TRAY_E_INVALID; }
    if (n_shards == 0 || shard >= n_shards || chunk_tiles == 0) { set_error("tray_render_shard_device: bad shard / chunk arguments"); return TRAY_E_INVALID; }
    // chunks c = shard, shard + n_shards, ... of chunk_tiles tiles each; the last chunk may be short
    uint32_t n_chunks = (s->n_tiles + chunk_tiles - 1) / chunk_tiles;
    uint32_t my_chunks = shard < n_chunks ? (n_chunks - shard + n_shards - 1) / n_shards : 0;
    if (my_chunks == 0) return launch_tiles(s, 0, 0, 1, 1, spp, seed, rgbw_dev, stream_);
    uint32_t last_chunk = shard + (my_chunks - 1) * n_shards;
    uint32_t tail = s->n_tiles - last_chunk * chunk_tiles;   // tiles in my last chunk
    if (tail > chunk_tiles) tail = chunk_tiles;
    uint32_t work = (my_chunks - 1) * chunk_tiles + tail;
    return launch_tiles(s, shard * chunk_tiles, work, chunk_tiles, n_shards, spp, seed, rgbw_dev, stream_);
}

// thread_work with sampler::Uniform / sampler::Adaptive (include/trayhip.h: tray_scene_set_sampler): rounds of k_sampler_pass (+
// k_sampler_decide) over batches of tiles, all on `stream`, no host synchronisation -- a pixel that is finished sits out the later rounds.
static int launch_sampler(TrayDeviceScene* s, uint32_t tile_start, uint32_t tile_count, uint32_t chunk, uint32_t chunk_stride,
                          uint32_t spp, uint32_t kf, float* rgbw_dev, hipStream_t stream) {
    SamplerPass sp{};
    sp.kind = s->sampler_kind; sp.min_spp = s->smp_min; sp.max_spp = s->smp_max;
    uint32_t rounds = 1;
    if (sp.kind == TRAY_SAMPLER_LOW_DISCREPANCY) {   // (scenes with an AnimatedMesh: LowDiscrepancy::get_samples hands out all spp samples of a pixel at once, ld.rs:33-52)
        sp.min_spp = sp.max_spp = spp; sp.step = 1u; sp.lum_cap = 0u;
    } else if (sp.kind == TRAY_SAMPLER_ADAPTIVE) {
        sp.step = tray_adaptive_step(sp.min_spp, sp.max_spp);
        while (sp.min_spp + (rounds - 1u) * sp.step < sp.max_spp) ++rounds;     // get_samples until samples_taken >= max_spp (adaptive.rs:136)
        sp.lum_cap = sp.min_spp + (rounds - 1u) * sp.step;
    } else { sp.min_spp = sp.max_spp = 1u; sp.step = 1u; sp.lum_cap = 0u; }
    // tiles per batch: the per-pixel state within 256 MB and the largest round within 2^28 threads
    const size_t px_bytes = 8u + 4u * (size_t)sp.lum_cap;
    const uint32_t widest = std::max(sp.min_spp, sp.step);
    uint32_t batch = (uint32_t)std::min<size_t>({(size_t)tile_count, ((size_t)256 << 20) / (64u * px_bytes), ((size_t)1 << 28) / (64u * (size_t)widest + TR_BLOCK)});
    if (batch == 0u) { set_error("Adaptive sampler: min_spp / max_spp too large for one tile's state"); return TRAY_E_UNSUPPORTED; }
    const size_t need = (size_t)batch * 64u * px_bytes;
    if (need > s->smp_bytes) {
        if (s->d_smp) { HIP_CHECK(hipStreamSynchronize(stream)); (void)hipFree(s->d_smp); s->d_smp = nullptr; s->smp_bytes = 0; }
        HIP_CHECK(hipMalloc(&s->d_smp, need));
        s->smp_bytes = need;
    }
    uint32_t* const px_state = static_cast<uint32_t*>(s->d_smp);
    float* const px_avg = reinterpret_cast<float*>(px_state + (size_t)batch * 64u);
    float* const px_lum = px_avg + (size_t)batch * 64u;
    HIP_CHECK(hipEventRecord(s->ev0, stream));
    uint32_t launches = 0;
    for (uint32_t item0 = 0; item0 < tile_count; item0 += batch) {
        const uint32_t n_items = std::min(batch, tile_count - item0), n_px = n_items * 64u;
        if (sp.kind == TRAY_SAMPLER_ADAPTIVE) HIP_CHECK(hipMemsetAsync(s->d_smp, 0, (size_t)batch * 64u * 8u, stream));   // states and averages
        for (uint32_t j = 0; j < rounds; ++j) {
            sp.pass = j;
            sp.count = sp.kind == TRAY_SAMPLER_ADAPTIVE ? (j == 0u ? sp.min_spp : sp.step) : sp.min_spp;   // (Uniform: 1, LowDiscrepancy: spp)
            sp.taken = sp.kind == TRAY_SAMPLER_ADAPTIVE ? sp.min_spp + j * sp.step : 0u;
            sp.before = j == 0u ? 0u : sp.min_spp + (j - 1u) * sp.step;
            // (k_sampler_pass: a workgroup owns a group of consecutive tiles -- enough of them for ~4096 (pixel, sample) pairs of the round, 16 at most: a
            // 32 x 32 pixel square of the Z-order queue -- and hands the pairs to its lanes as their paths end)
            const uint32_t per_tile = 64u * sp.count;
            uint32_t group = std::max(1u, std::min<uint32_t>(SP_GROUP_MAX, 4096u / per_tile));
            if (const char* ge = getenv("TRAYHIP_SAMPLER_GROUP")) group = (uint32_t)std::max(1, std::min(SP_GROUP_MAX, atoi(ge)));   // (measurement)
            const dim3 grid((n_items + group - 1u) / group), block(TR_BLOCK);
#define SAMPLER_PASS(A, F) hipLaunchKernelGGL((k_sampler_pass<A, F>), grid, block, s->stack_bytes, stream, s->dev, s->d_tiles + tile_start, item0, n_items, chunk, chunk_stride, kf, sp, px_state, px_lum, rgbw_dev, s->d_stats, group)
            const bool lean = s->feat == FEAT_NONE && s->dev.integrator != TRAY_INTEGRATOR_WHITTED;   // (no optional lobe, no texture: the small instantiation)
            if (s->deforming) { if (lean) SAMPLER_PASS(3, FEAT_NONE); else SAMPLER_PASS(3, FEAT_ALL | FEAT_TEX); }
            else if (s->animated) { if (lean) SAMPLER_PASS(2, FEAT_NONE); else SAMPLER_PASS(2, FEAT_ALL | FEAT_TEX); }
            else { if (lean) SAMPLER_PASS(0, FEAT_NONE); else SAMPLER_PASS(0, FEAT_ALL | FEAT_TEX); }
#undef SAMPLER_PASS
            ++launches;
            if (sp.kind == TRAY_SAMPLER_ADAPTIVE) {
                hipLaunchKernelGGL(k_sampler_decide, dim3((n_px + TR_BLOCK - 1) / TR_BLOCK), block, 0, stream, n_px, sp, px_state, px_avg, px_lum);
                ++launches;
            }
        }
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(s->ev1, stream));
    s->timing_valid = true;
    s->launches = launches;
    return TRAY_OK;
}

// The transform table of a moving scene's frame (dev_geom.h: xf_time_index; k_xf_table_build): wanted when the launch needs every time index
// about twice over -- from 3e7 camera samples on (a 1080p frame at 16 spp; a GPU's eighth of a 512-spp frame is 1.3e8; building the table costs
// what 2^24 = 1.7e7 camera samples' evaluations cost, ~1.4 ms per moving instance, but it also takes 1.9 GB per instance) --, or as
// tray_scene_set_transform_table says. Built once per frame on the launch stream by the first launch that wants it. If the allocation fails
// the launch evaluates per path. Sets s->launch_dev.
#ifndef XF_TABLE_MIN_SAMPLES
#define XF_TABLE_MIN_SAMPLES 30000000ull
#endif
// the wavefront schedule's per-path transform cache, on first use (one record of TR_XF_WORDS floats per pool slot and moving instance): as many
// columns as scene_build budgeted; halved while hipMalloc refuses and the pool does not exist yet, never below the pool's slots once it does
static int xf_cache_ensure(TrayDeviceScene* s) {
    if (!s->wavefront || !s->animated || s->dev.n_moving == 0u || s->dev.xf_cache) return TRAY_OK;
    // (a pool that was sized for launches reading the frame's table -- no cache bounded it -- can be larger than any cache the device holds: then the
    // pool goes, the cache takes its budgeted size and launch_wavefront allocates a pool that fits it)
    if (s->wf_ready && s->dev.xf_cache_lanes && s->pool.n_slots > s->dev.xf_cache_lanes) {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)s->dev.n_moving * TR_XF_REC * s->pool.n_slots * sizeof(float);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        if (need > free_b / 5u * 2u) { HIP_CHECK(hipDeviceSynchronize()); wf_free(s); }
    }
    uint32_t lanes = std::max<uint32_t>(s->dev.xf_cache_lanes, s->wf_ready ? s->pool.n_slots : 0u);
    const uint32_t floor_lanes = s->wf_ready ? s->pool.n_slots : 64u * TR_BLOCK;
    for (;;) {
        const size_t bytes = (size_t)s->dev.n_moving * TR_XF_REC * lanes * sizeof(float);
        void* p = nullptr;
        if (hipMalloc(&p, bytes) == hipSuccess) {
            s->allocs.push_back(p);
            s->dev.xf_cache = static_cast<float*>(p); s->dev.xf_cache_lanes = lanes; s->xf_cache_bytes = bytes;
            return TRAY_OK;
        }
        (void)hipGetLastError();
        if (lanes <= floor_lanes) {
            set_error("hipMalloc of the per-path transform cache failed: " + std::to_string(bytes >> 20) + " MiB for " + std::to_string(s->dev.n_moving) +
                      " moving instances x " + std::to_string(lanes) + " paths (tray_scene_set_wavefront / TRAYHIP_XF_CACHE_BYTES bound it)");
            return TRAY_E_NOMEM;
        }
        lanes = std::max(floor_lanes, lanes / 2u / TR_BLOCK * TR_BLOCK);
    }
}
static int xf_table_prepare(TrayDeviceScene* s, uint64_t samples, hipStream_t stream) {
    s->launch_dev = s->dev;
    s->last_used_table = false;
    const uint32_t stride = s->dev.n_moving + (s->camera_animated ? 1u : 0u);
    if (!s->animated || stride == 0u) return TRAY_OK;
    int want = s->xf_table_req;
    if (const char* e = getenv("TRAYHIP_XF_TABLE")) want = atoi(e) != 0 ? 1 : 0;
    // (a table that exists for this frame serves every launch; so does the buffer of the previous frame's: a sequence that rendered through the table goes on
    // doing so -- its pool may be larger than a per-path cache could cover)
    if (want < 0) want = (samples >= XF_TABLE_MIN_SAMPLES || s->xf_table_built || s->d_xf_table) ? 1 : 0;
    if (!want) {
        const int rc = xf_cache_ensure(s);
        if (rc == TRAY_OK) { s->launch_dev = s->dev; return TRAY_OK; }
        if (rc != TRAY_E_NOMEM) return rc;   // (no room for the cache: the table is smaller from a few million slots on)
    }
    if (s->d_xf_table && s->xf_table_cap < stride) {   // (the buffer taken over from the previous frame is too small for this frame's records)
        forget_alloc(s, s->d_xf_table); (void)hipFree(s->d_xf_table); s->d_xf_table = nullptr; s->xf_table_cap = 0u; s->xf_table_built = false;
    }
    if (s->d_xf_table && s->xf_table_stride != stride) { s->xf_table_stride = stride; s->xf_table_built = false; }
    if (!s->d_xf_table) {
        // room for what any frame of the sequence can move (instances whose transforms have several keyframes, tray_scene_create), if that much is to be had
        uint32_t cap = std::max(stride, s->xf_movable);
        size_t bytes = ((size_t)1 << 24) * cap * TR_XF_REC * sizeof(float);   // 2.1 GB per moving instance
        void* p = nullptr;
        // by the library's own rule (nobody asked for the table) it takes at most a third of what is free: on a device shared with other
        // allocators 22.5 GB for eleven movers must not be what starves them; then, or if hipMalloc refuses: per-path evaluation
        bool room = true;
        const bool own_rule = s->xf_table_req < 0 && !getenv("TRAYHIP_XF_TABLE");
        for (;;) {   // (the sequence's headroom first, then this frame's records alone)
            room = true;
            if (own_rule) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) room = bytes <= free_b / 3u; else (void)hipGetLastError();
            }
            if (room && hipMalloc(&p, bytes) == hipSuccess) break;
            (void)hipGetLastError();
            p = nullptr;
            if (cap == stride) break;
            cap = stride; bytes = ((size_t)1 << 24) * cap * TR_XF_REC * sizeof(float);
        }
        if (!p) {
            const int rc = xf_cache_ensure(s);
            s->launch_dev = s->dev;
            return rc;
        }
        s->allocs.push_back(p);
        s->d_xf_table = static_cast<float*>(p);
        s->xf_table_stride = stride;
        s->xf_table_cap = cap;
        s->xf_table_built = false;
    }
    if (!s->xf_table_built) {
        DevScene d = s->dev;   // (the build reads moving_ids, instances, spline tables, camera: none of them depends on the cache fields)
        const uint32_t n_index = 1u << 24;
        hipLaunchKernelGGL(k_xf_table_build, dim3(n_index / TR_BLOCK, stride), dim3(TR_BLOCK), 0, stream, d, s->d_xf_table, stride, 0u, n_index);
        HIP_CHECK(hipGetLastError());
        // the build is ordered before this launch by the stream; the frame's later launches may come on other streams (tray_render_tiles_device /
        // tray_render_shard_device take the caller's): they wait for this event instead of reading a table that is still being written
        if (!s->xf_table_ev) HIP_CHECK(hipEventCreateWithFlags(&s->xf_table_ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(s->xf_table_ev, stream));
        s->xf_table_stream = stream;
        s->xf_table_built = true;
    } else if (stream != s->xf_table_stream && s->xf_table_ev) {
        HIP_CHECK(hipStreamWaitEvent(stream, s->xf_table_ev, 0));
    }
    s->launch_dev.xf_tab = s->d_xf_table;
    s->launch_dev.xf_tab_stride = s->xf_table_stride;
    if (s->wavefront) {   // the stage kernels index the table by the path's time index; the tile kernel keeps its cache columns and fills them from it
        s->launch_dev.xf_cache = s->d_xf_table;
        s->launch_dev.xf_table = 1u;
        s->launch_dev.xf_aos = 1u;
        s->launch_dev.xf_stride = s->xf_table_stride;
    }
    s->last_used_table = true;
    return TRAY_OK;
}

static int launch_tiles(TrayDeviceScene* s, uint32_t tile_start, uint32_t tile_count, uint32_t chunk, uint32_t chunk_stride,
                        uint32_t spp, uint64_t seed, float* rgbw_dev, void* stream_) {
    if (s->sampler_kind == TRAY_SAMPLER_LOW_DISCREPANCY && (spp == 0 || (spp & (spp - 1)) != 0)) {
        set_error("spp must be a power of two (LowDiscrepancy sampler, ld.rs:22-25); use tray_round_spp"); return TRAY_E_INVALID;
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (s->broken) { set_error("this device scene is unusable: a tray_scene_update_frame on it failed"); return TRAY_E_INVALID; }
    HIP_CHECK(hipSetDevice(s->device));
    s->timing_valid = false;
    s->empty_launch = false;
    // (an empty queue is a valid work item -- a GPU's shard of a small frame dealt to many GPUs: tray_last_timing then reports zeros
    // instead of "no launch recorded", which made tray_multi_timing fail for 48 tiles on 8 devices: found by tests/test_multi_stub.py)
    if (tile_count == 0) { std::fprintf(stderr, "Warning: This block queue is empty!\n"); s->empty_launch = true; return TRAY_OK; }   // block_queue.rs:42-44
    HIP_CHECK(hipMemsetAsync(s->d_counter, 0, sizeof(uint32_t), stream));
    HIP_CHECK(hipMemsetAsync(s->d_stats, 0, WF_STAT_SLOTS * sizeof(DevStats), stream));
    HIP_CHECK(hipMemsetAsync(s->d_retraced, 0, sizeof(uint32_t), stream));
    // key_frame on the host (same mixing as the device function)
    auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
    uint32_t kf = mix((uint32_t)seed + 0x9E3779B9u);
    kf = mix(kf ^ (uint32_t)(seed >> 32));
    kf = mix(kf + s->dev.frame);
    if (s->sampler_kind != TRAY_SAMPLER_LOW_DISCREPANCY || s->deforming) return launch_sampler(s, tile_start, tile_count, chunk, chunk_stride, spp, kf, rgbw_dev, stream);
    s->last_was_wavefront = false;
    {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = xf_table_prepare(s, (uint64_t)tile_count * 64u * spp, stream);
        if (getenv("TRAYHIP_STATS")) fprintf(stderr, "[trayhip] transform table / cache prepared in %.1f ms (table %s, %u of %u records per index in use)\n",
                                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), s->last_used_table ? "on" : "off", s->xf_table_stride, s->xf_table_cap);
        if (rc != TRAY_OK) return rc;
    }
    if (s->wavefront) {
        const auto t0 = std::chrono::steady_clock::now();
        const bool had_pool = s->wf_ready;
        const int rc = launch_wavefront(s, tile_start, tile_count, chunk, chunk_stride, spp, kf, rgbw_dev, stream);
        if (getenv("TRAYHIP_STATS")) fprintf(stderr, "[trayhip] wavefront launch enqueued in %.1f ms (pool %s: %u slots)\n",
                                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), had_pool ? "kept" : "allocated", s->pool.n_slots);
        return rc;
    }
    // Items per tile (k_path_tiles: progressive slices, level-major: the launch ends with its smallest items). A slice costs its own film resolve and
    // flush and keeps >= 64 samples per pixel (>= 256 in a launch with many tiles per workgroup). Measured (profiles/r06_tile_slices_progressive_ab.txt, items per tile 1 / 2 / 3 / 4 / 5): the whole
    // dragon frame 718.5 / 740.6 / 750.3 / 752.7 / 755.1 Msamples/s (tiles that show the mesh cost several times a wall tile: with whole tiles the last
    // round of the 768 workgroups is one such tile), the whole cornell_box frame 1160.9 / 1163.5 / 1163.6 / 1159.9 / 1154.3; a GPU's eighth of the
    // frame (4050 tiles, slowest of the eight shards against an eighth of the whole frame): dragon 0.449 / 0.632 / 0.784 / 0.862 / 0.872,
    // cornell_box 0.895 / 0.942 / 0.960 / 0.971 / 0.975. So: three items per tile for a launch with many tiles per workgroup, up to five for a small one.
    // TRAYHIP_TILE_SLICES=<items per tile> overrides.
    uint32_t levels = 1u;
    {
        // (a launch with many tiles per workgroup keeps >= 256 samples per slice: at 256 spp three items per tile cost cornell_box 2.8 %, 1114 against 1146
        // Msamples/s, profiles/r06_c2_kept_gate_distance_ab.txt -- the resolves outweigh a tail that is 1 / 42 of the launch there)
        const bool small = tile_count < 12u * (uint32_t)s->n_blocks;
        const uint32_t most = small ? 5u : 3u, least = small ? 64u : 256u;
        while (levels < most && (spp >> levels) >= least) ++levels;   // (the last two slices are spp >> (levels - 1) samples each)
    }
    if (const char* e = getenv("TRAYHIP_TILE_SLICES")) { levels = 1u; const uint32_t want = (uint32_t)std::max(1, atoi(e)); while (levels < want && (spp >> levels) >= 1u) ++levels; }
    int blocks = (int)std::min<uint64_t>((uint64_t)s->n_blocks, (uint64_t)tile_count * levels);
#ifdef TR_SAMPLE_DUMP
    void* dump_buf = nullptr;
    const size_t dump_bytes = (size_t)s->dev.width * s->dev.height * spp * 8 * sizeof(float);
    if (getenv("TRAYHIP_SAMPLE_DUMP")) {
        HIP_CHECK(hipMalloc(&dump_buf, dump_bytes));
        HIP_CHECK(hipMemset(dump_buf, 0, dump_bytes));
    }
    s->launch_dev.sample_dump = static_cast<float4*>(dump_buf);
#endif
    HIP_CHECK(hipEventRecord(s->ev0, stream));
#define PATH_TILES_L(A, F, L) hipLaunchKernelGGL((k_path_tiles<A, F, TRAY_INTEGRATOR_PATH, L>), dim3(blocks), dim3(TR_BLOCK), s->stack_bytes, stream, s->launch_dev, s->d_tiles + tile_start, \
                                                 tile_count, chunk, chunk_stride, spp, kf, levels, rgbw_dev, s->d_counter, s->d_stats)
#define PATH_TILES(A, F) do { if (s->light_filter) PATH_TILES_L(A, F, true); else PATH_TILES_L(A, F, false); } while (0)
#define PATH_TILES_F(A) do { if (s->feat == FEAT_NONE) PATH_TILES(A, FEAT_NONE); else if (s->feat == FEAT_MERL) PATH_TILES(A, FEAT_MERL); \
                             else if (s->feat == FEAT_SPEC) PATH_TILES(A, FEAT_SPEC); else if (s->feat == (FEAT_MERL | FEAT_SPEC)) PATH_TILES(A, FEAT_MERL | FEAT_SPEC); \
                             else if (s->feat == (FEAT_ALL | FEAT_TEX)) PATH_TILES(A, FEAT_ALL | FEAT_TEX); else PATH_TILES(A, FEAT_ALL); } while (0)
#define WHITTED_TILES(A) hipLaunchKernelGGL((k_path_tiles<A, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_WHITTED>), dim3(blocks), dim3(TR_BLOCK), s->stack_bytes, stream, s->launch_dev, \
                                             s->d_tiles + tile_start, tile_count, chunk, chunk_stride, spp, kf, levels, rgbw_dev, s->d_counter, s->d_stats)
    if (s->dev.integrator == TRAY_INTEGRATOR_WHITTED) { if (s->animated) WHITTED_TILES(1); else WHITTED_TILES(0); }
    else if (s->animated) PATH_TILES_F(1);
    else PATH_TILES_F(0);
#undef WHITTED_TILES
#undef PATH_TILES_F
#undef PATH_TILES
#undef PATH_TILES_L
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(s->ev1, stream));
    s->timing_valid = true;
    s->launches = 1;
#ifdef TR_SAMPLE_DUMP   // instrumented builds only: TRAYHIP_SAMPLE_DUMP=<file> receives width x height x spp records of two float4 (radiance, vertices | final throughput, bounce) of the launch
    if (dump_buf) {
        HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<float> host(dump_bytes / sizeof(float));
        HIP_CHECK(hipMemcpy(host.data(), dump_buf, dump_bytes, hipMemcpyDeviceToHost));
        (void)hipFree(dump_buf);
        if (FILE* fp = std::fopen(getenv("TRAYHIP_SAMPLE_DUMP"), "wb")) { std::fwrite(host.data(), 1, dump_bytes, fp); std::fclose(fp); }
    }
#endif
    return TRAY_OK;
}

int tray_render_tiles(TrayDeviceScene* s, uint32_t tile_start, uint32_t tile_count, uint32_t spp, uint64_t seed, float* rgbw_host) {
    if (!s || !rgbw_host) { set_error("tray_render_tiles: null argument"); return TRAY_E_INVALID; }
    HIP_CHECK(hipSetDevice(s->device));
    size_t n = (size_t)s->dev.width * s->dev.height * 4;
    float* d = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    HIP_CHECK(hipMalloc(&d, n * sizeof(float)));
    int rc = TRAY_OK;
    hipError_t e = hipMemset(d, 0, n * sizeof(float));
    double t_alloc = ms(), t_enq = 0.0, t_copy = 0.0;
    if (e == hipSuccess) {
        rc = tray_render_tiles_device(s, tile_start, tile_count, spp, seed, d, nullptr);
        t_enq = ms();
        if (rc == TRAY_OK) {
            std::vector<float> tmp(n);
            e = hipMemcpy(tmp.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
            t_copy = ms();
            if (e == hipSuccess)
                for (size_t i = 0; i < n; ++i) rgbw_host[i] += tmp[i];
        }
    }
    (void)hipFree(d);
    if (getenv("TRAYHIP_STATS")) fprintf(stderr, "[trayhip] tray_render_tiles: film buffer %.1f ms, enqueue %.1f, kernels + copy %.1f, add + free %.1f\n", t_alloc, t_enq - t_alloc, t_copy - t_enq, ms() - t_copy);
    if (e != hipSuccess) { set_error(std::string("tray_render_tiles: ") + hipGetErrorString(e)); return TRAY_E_DEVICE; }
    return rc;
}

// ---- several GPUs of this process: shard + RCCL sum-reduce inside the library ----------------------------------------------
namespace {
struct Rccl {   // the six entry points, resolved from librccl.so on first use
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("librccl.so could not be loaded: ") + dlerror(); return false; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        Reduce = reinterpret_cast<decltype(Reduce)>(dlsym(lib, "ncclReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !GetErrorString) { err = "librccl.so lacks an expected entry point"; lib = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
}  // namespace

struct TrayMultiScene {
    int n_dev = 0;
    std::vector<int> dev_ids;
    std::vector<TrayDeviceScene*> scenes;
    std::vector<float*> films;          // one full-frame RGBW buffer per device
    std::vector<hipStream_t> streams;
    std::vector<void*> comms;           // ncclComm_t
    size_t n_floats = 0;
    float reduce_ms = 0.0f;
    hipEvent_t r0 = nullptr, r1 = nullptr;   // around the reduce, on the first device's stream
};

void tray_multi_destroy(TrayMultiScene* m) {
    if (!m) return;
    int current = 0;
    const bool have_current = hipGetDevice(&current) == hipSuccess;   // the caller's current device is left as it was
    for (int d = 0; d < (int)m->scenes.size(); ++d) {
        (void)hipSetDevice(m->dev_ids[d]);
        if (d < (int)m->comms.size() && m->comms[d] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comms[d]);
        if (d < (int)m->films.size() && m->films[d]) (void)hipFree(m->films[d]);
        if (d < (int)m->streams.size() && m->streams[d]) (void)hipStreamDestroy(m->streams[d]);
        if (d == 0) { if (m->r0) (void)hipEventDestroy(m->r0); if (m->r1) (void)hipEventDestroy(m->r1); }
        tray_scene_destroy(m->scenes[d]);
    }
    if (have_current) (void)hipSetDevice(current);
    delete m;
}

int tray_multi_create(const TrayFlatScene* f, int n_dev, const int* dev_ids, TrayMultiScene** out) {
    if (!f || !out || !dev_ids || n_dev < 1) { set_error("tray_multi_create: null argument or no device"); return TRAY_E_INVALID; }
    *out = nullptr;
    int have = 0;
    HIP_CHECK(hipGetDeviceCount(&have));
    for (int d = 0; d < n_dev; ++d) {
        if (dev_ids[d] < 0 || dev_ids[d] >= have) { set_error("tray_multi_create: no such HIP device " + std::to_string(dev_ids[d])); return TRAY_E_INVALID; }
        for (int e = 0; e < d; ++e) if (dev_ids[e] == dev_ids[d]) { set_error("tray_multi_create: a device is listed twice"); return TRAY_E_INVALID; }
    }
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        std::string err;
        if (!g_rccl.load(err)) { set_error("tray_multi_create: " + err); return TRAY_E_UNSUPPORTED; }
    }
    TrayMultiScene* m = new TrayMultiScene();
    m->n_dev = n_dev;
    m->dev_ids.assign(dev_ids, dev_ids + n_dev);
    m->n_floats = (size_t)f->film.width * f->film.height * 4;
    const int before = g_device;
    int rc = TRAY_OK;
    for (int d = 0; d < n_dev && rc == TRAY_OK; ++d) {
        rc = tray_init(dev_ids[d]);
        TrayDeviceScene* s = nullptr;
        if (rc == TRAY_OK) rc = tray_scene_create(f, &s);
        if (rc != TRAY_OK) break;
        m->scenes.push_back(s);
        float* film = nullptr;
        hipStream_t st = nullptr;
        if (hipMalloc(&film, m->n_floats * sizeof(float)) != hipSuccess || hipStreamCreate(&st) != hipSuccess) { set_error("tray_multi_create: film / stream allocation failed"); rc = TRAY_E_NOMEM; }
        m->films.push_back(film); m->streams.push_back(st);
    }
    if (rc == TRAY_OK) {
        m->comms.assign(n_dev, nullptr);
        const int nr = g_rccl.CommInitAll(m->comms.data(), n_dev, dev_ids);
        if (nr != 0) { set_error(std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(nr)); rc = TRAY_E_DEVICE; }
    }
    if (rc == TRAY_OK) {
        (void)hipSetDevice(dev_ids[0]);
        if (hipEventCreate(&m->r0) != hipSuccess || hipEventCreate(&m->r1) != hipSuccess) { set_error("hipEventCreate failed"); rc = TRAY_E_DEVICE; }
    }
    (void)tray_init(before);
    if (rc != TRAY_OK) { tray_multi_destroy(m); return rc; }
    *out = m;
    return TRAY_OK;
}

// The same scene at another frame on every device (tray_scene_update_frame); communicators, films and streams stay.
int tray_multi_set_sampler(TrayMultiScene* m, uint32_t kind, uint32_t min_spp, uint32_t max_spp) {
    if (!m) { set_error("tray_multi_set_sampler: null argument"); return TRAY_E_INVALID; }
    for (TrayDeviceScene* s : m->scenes) {
        const int rc = tray_scene_set_sampler(s, kind, min_spp, max_spp);
        if (rc != TRAY_OK) return rc;
    }
    return TRAY_OK;
}

int tray_multi_update_frame(TrayMultiScene* m, const TrayFlatScene* f) {
    if (!m || !f) { set_error("tray_multi_update_frame: null argument"); return TRAY_E_INVALID; }
    int current = 0;
    const bool have_current = hipGetDevice(&current) == hipSuccess;
    int rc = TRAY_OK;
    for (int d = 0; d < m->n_dev && rc == TRAY_OK; ++d) rc = tray_scene_update_frame(m->scenes[d], f);
    if (have_current) (void)hipSetDevice(current);
    return rc;
}

struct CurrentDeviceGuard {   // the multi-device entry points leave the caller's current HIP device as they found it
    int dev = 0;
    bool ok = false;
    CurrentDeviceGuard() { ok = hipGetDevice(&dev) == hipSuccess; }
    ~CurrentDeviceGuard() { if (ok) (void)hipSetDevice(dev); }
};

int tray_render_frame_multi(TrayMultiScene* m, uint32_t spp, uint64_t seed, float* rgbw_host) {
    if (!m || !rgbw_host) { set_error("tray_render_frame_multi: null argument"); return TRAY_E_INVALID; }
    CurrentDeviceGuard keep_current;
    // one host thread per device: the wavefront schedule polls its stream, and the launches of different devices must overlap
    std::vector<int> rcs(m->n_dev, TRAY_OK);
    std::vector<std::string> errs(m->n_dev);
    std::vector<std::thread> workers;
    for (int d = 0; d < m->n_dev; ++d)
        workers.emplace_back([&, d] {
            if (hipSetDevice(m->dev_ids[d]) != hipSuccess || hipMemsetAsync(m->films[d], 0, m->n_floats * sizeof(float), m->streams[d]) != hipSuccess) {
                rcs[d] = TRAY_E_DEVICE; errs[d] = "hipSetDevice / hipMemsetAsync failed"; return;
            }
            rcs[d] = tray_render_shard_device(m->scenes[d], (uint32_t)d, (uint32_t)m->n_dev, 16u, spp, seed, m->films[d], m->streams[d]);
            if (rcs[d] != TRAY_OK) errs[d] = tray_last_error();
        });
    for (std::thread& w : workers) w.join();
    for (int d = 0; d < m->n_dev; ++d)
        if (rcs[d] != TRAY_OK) { set_error("tray_render_frame_multi: device " + std::to_string(m->dev_ids[d]) + ": " + errs[d]); return rcs[d]; }
    // film::Image::add_blocks on the master == one sum-reduce onto the first device (in place on the root). Every device's shard is
    // finished before the reduce is timed, so reduce_ms is the collective alone, not the wait for the slowest shard
    for (int d = 0; d < m->n_dev; ++d) { HIP_CHECK(hipSetDevice(m->dev_ids[d])); HIP_CHECK(hipStreamSynchronize(m->streams[d])); }
    HIP_CHECK(hipSetDevice(m->dev_ids[0]));
    HIP_CHECK(hipEventRecord(m->r0, m->streams[0]));
    int nr = g_rccl.GroupStart();
    for (int d = 0; d < m->n_dev && nr == 0; ++d) {
        (void)hipSetDevice(m->dev_ids[d]);
        nr = g_rccl.Reduce(m->films[d], m->films[d], m->n_floats, /*ncclFloat*/ 7, /*ncclSum*/ 0, /*root*/ 0, m->comms[d], m->streams[d]);
    }
    const int ne = g_rccl.GroupEnd();
    if (nr == 0) nr = ne;
    if (nr != 0) { set_error(std::string("ncclReduce failed: ") + g_rccl.GetErrorString(nr)); return TRAY_E_DEVICE; }
    HIP_CHECK(hipSetDevice(m->dev_ids[0]));
    HIP_CHECK(hipEventRecord(m->r1, m->streams[0]));
    for (int d = 0; d < m->n_dev; ++d) { HIP_CHECK(hipSetDevice(m->dev_ids[d])); HIP_CHECK(hipStreamSynchronize(m->streams[d])); }
    HIP_CHECK(hipSetDevice(m->dev_ids[0]));
    HIP_CHECK(hipEventElapsedTime(&m->reduce_ms, m->r0, m->r1));
    std::vector<float> tmp(m->n_floats);
    HIP_CHECK(hipMemcpy(tmp.data(), m->films[0], m->n_floats * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m->n_floats; ++i) rgbw_host[i] += tmp[i];
    return TRAY_OK;
}

int tray_scene_set_transform_table(TrayDeviceScene* s, int mode) {
    if (!s) { set_error("tray_scene_set_transform_table: null argument"); return TRAY_E_INVALID; }
    if (mode < -1 || mode > 1) { set_error("tray_scene_set_transform_table: mode is -1 (by the launch's sample count), 0 (never) or 1 (always)"); return TRAY_E_INVALID; }
    s->xf_table_req = mode;
    return TRAY_OK;
}
int tray_multi_set_transform_table(TrayMultiScene* m, int mode) {
    if (!m) { set_error("tray_multi_set_transform_table: null argument"); return TRAY_E_INVALID; }
    for (TrayDeviceScene* s : m->scenes) {
        const int rc = tray_scene_set_transform_table(s, mode);
        if (rc != TRAY_OK) return rc;
    }
    return TRAY_OK;
}
int tray_debug_transform_table(TrayDeviceScene* s, uint32_t n, uint32_t* n_differ) {
    if (!s || !n_differ) { set_error("tray_debug_transform_table: null argument"); return TRAY_E_INVALID; }
    *n_differ = 0;
    if (!s->d_xf_table || !s->xf_table_built) { set_error("tray_debug_transform_table: this frame has no transform table (no launch wanted one yet)"); return TRAY_E_INVALID; }
    HIP_CHECK(hipSetDevice(s->device));
    uint32_t* d_bad = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_bad), sizeof(uint32_t)));
    hipError_t e = hipMemset(d_bad, 0, sizeof(uint32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_xf_table_check, dim3((n + TR_BLOCK - 1) / TR_BLOCK), dim3(TR_BLOCK), 0, nullptr, s->dev, s->d_xf_table, s->xf_table_stride, n, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(n_differ, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (e != hipSuccess) { set_error(std::string("tray_debug_transform_table: ") + hipGetErrorString(e)); return TRAY_E_DEVICE; }
    return TRAY_OK;
}
int tray_scene_set_wavefront(TrayDeviceScene* s, uint32_t pool_slots, uint32_t views, uint32_t slices) {
    if (!s) { set_error("tray_scene_set_wavefront: null argument"); return TRAY_E_INVALID; }
    if (s->broken) { set_error("tray_scene_set_wavefront: the handle is only good for tray_scene_destroy after a failed frame update"); return TRAY_E_INVALID; }
    if (views > (uint32_t)WF_PIPES_MAX) { set_error("tray_scene_set_wavefront: at most " + std::to_string(WF_PIPES_MAX) + " views"); return TRAY_E_INVALID; }
    if (slices > WF_MAX_SLICES || (slices & (slices - 1u)) != 0u) { set_error("tray_scene_set_wavefront: slices per tile must be 0 or a power of two up to " + std::to_string(WF_MAX_SLICES)); return TRAY_E_INVALID; }
    const uint32_t before = wf_slot_wish(s);
    s->wf_req_slots = pool_slots; s->wf_req_views = views; s->wf_req_slices = slices;
    if (s->wf_ready && wf_slot_wish(s) != before) {   // another pool size: the buffers go, the next render call allocates them anew
        HIP_CHECK(hipSetDevice(s->device));
        HIP_CHECK(hipDeviceSynchronize());
        wf_free(s);
        s->wf_shrunk = false;
    }
    return TRAY_OK;
}
int tray_multi_set_wavefront(TrayMultiScene* m, uint32_t pool_slots, uint32_t views, uint32_t slices) {
    if (!m) { set_error("tray_multi_set_wavefront: null argument"); return TRAY_E_INVALID; }
    for (TrayDeviceScene* s : m->scenes) {
        const int rc = tray_scene_set_wavefront(s, pool_slots, views, slices);
        if (rc != TRAY_OK) return rc;
    }
    return TRAY_OK;
}
int tray_last_schedule(TrayDeviceScene* s, TrayScheduleInfo* out) {
    if (!s || !out) { set_error("tray_last_schedule: null argument"); return TRAY_E_INVALID; }
    std::memset(out, 0, sizeof *out);
    out->wavefront = s->wavefront ? 1u : 0u;
    out->launched_wavefront = s->last_was_wavefront ? 1u : 0u;
    out->pool_slots = s->pool.n_slots; out->chunks = s->n_chunks;
    out->views = s->last_was_wavefront ? s->last_views : 0u; out->slices = s->last_was_wavefront ? s->last_slices : 0u;
    out->pool_bytes = s->pool.data ? (uint64_t)F_COUNT * s->pool.n_slots * sizeof(float) : 0u;
    out->schedule_bytes = s->pool.data ? (uint64_t)s->pool.n_slots * wf_bytes_per_slot() : 0u;
    out->xf_cache_bytes = s->xf_cache_bytes;
    out->n_moving = s->dev.n_moving;
    out->tile_workgroups = (uint32_t)s->n_blocks;
    out->transform_table = s->last_used_table ? 1u : 0u;
    out->binned_stages = (s->last_was_wavefront && WF_FOLD_C) ? s->wf_bin_stages : 0u;
    out->xf_table_bytes = s->d_xf_table ? ((uint64_t)1 << 24) * s->xf_table_cap * TR_XF_REC * sizeof(float) : 0u;   // (the buffer: sized for what any frame of the sequence can move)
    return TRAY_OK;
}

int tray_last_timing(TrayDeviceScene* s, TrayKernelTiming* t);
int tray_multi_timing(TrayMultiScene* m, TrayKernelTiming* per_device, float* reduce_ms) {
    if (!m || !per_device) { set_error("tray_multi_timing: null argument"); return TRAY_E_INVALID; }
    CurrentDeviceGuard keep_current;   // (tray_last_timing makes each scene's device current: found by tests/test_multi_stub.py)
    for (int d = 0; d < m->n_dev; ++d) {
        const int rc = tray_last_timing(m->scenes[d], per_device + d);
        if (rc != TRAY_OK) return rc;
    }
    if (reduce_ms) *reduce_ms = m->reduce_ms;
    return TRAY_OK;
}

int tray_last_timing(TrayDeviceScene* s, TrayKernelTiming* t) {
    if (!s || !t) { set_error("tray_last_timing: null argument"); return TRAY_E_INVALID; }
    std::memset(t, 0, sizeof *t);
    if (s->empty_launch) return TRAY_OK;
    if (!s->timing_valid) { set_error("tray_last_timing: no launch recorded"); return TRAY_E_INVALID; }
    HIP_CHECK(hipSetDevice(s->device));
    HIP_CHECK(hipEventSynchronize(s->ev1));
    HIP_CHECK(hipEventElapsedTime(&t->render_ms, s->ev0, s->ev1));
    std::vector<DevStats> all(WF_STAT_SLOTS);
    HIP_CHECK(hipMemcpy(all.data(), s->d_stats, all.size() * sizeof(DevStats), hipMemcpyDeviceToHost));
    DevStats st{};
    for (const DevStats& a : all) {
        st.samples += a.samples; st.vertices += a.vertices; st.rays += a.rays;
        for (int k = 0; k < 18; ++k) st.trav[k] += a.trav[k];
    }
#ifdef TR_STAGE_CLOCKS
    if (getenv("TRAYHIP_STATS") && st.trav[0]) {
        double tot = 0;
        for (int k = 0; k < 7; ++k) tot += (double)st.trav[k];
        const char* names[7] = {"trace A", "trace B", "trace C", "vertex_begin", "queries", "vertex_end", "regen+film"};
        for (int k = 0; k < 7; ++k) fprintf(stderr, "[trayhip] %-12s %5.1f %% of wave cycles\n", names[k], 100.0 * (double)st.trav[k] / tot);
        const char* parts[3] = {"sample head / light setup", "eval + pdf site", "epilogue of the query kind"};
        for (int k = 0; k < 3; ++k) fprintf(stderr, "[trayhip]   queries: %-28s %5.1f %% of wave cycles\n", parts[k], 100.0 * (double)st.trav[8 + k] / tot);
    }
#elif defined(WF_TRACE_CLOCKS)
    if (getenv("TRAYHIP_STATS") && st.rays)
        for (int g = 0; g < 3; ++g) {
            const unsigned long long* t = st.trav + g * 6;
            double tot = 0;
            for (int k = 0; k < 5; ++k) tot += (double)t[k];
            if (tot > 0) fprintf(stderr, "[trayhip] trace %c wave cycles: refill %.1f %%  node phase %.1f %%  leaf phase %.1f %%  pop phase %.1f %%  result write %.1f %%  (%.3e clock64 ticks in all waves)\n", "ABC"[g],
                                 100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot, 100 * t[4] / tot, tot);
        }
#else
    if (getenv("TRAYHIP_STATS") && st.rays)
        for (int g = 0; g < 3; ++g) {
            const unsigned long long* t = st.trav + g * 6;
            if (!t[5]) continue;
            fprintf(stderr, "[trayhip] stage %c: %llu rays; per ray: node steps %.2f  single-node visits %.2f  two-child expansions %.2f  instance entries %.2f  triangle tests %.2f\n",
                    "ABC"[g], t[5], (double)t[0] / t[5], (double)t[1] / t[5], (double)t[2] / t[5], (double)t[3] / t[5], (double)t[4] / t[5]);
        }
#endif
    uint32_t retraced = 0;
    HIP_CHECK(hipMemcpy(&retraced, s->d_retraced, sizeof retraced, hipMemcpyDeviceToHost));
    t->launches = s->launches;
    t->samples = st.samples; t->vertices = st.vertices; t->rays = st.rays; t->retraced = retraced;
    return TRAY_OK;
}

int tray_debug_intersect(TrayDeviceScene* s, uint32_t n, const TrayRay* rays, TrayHit* hits) {
    if (!s || !rays || !hits) { set_error("tray_debug_intersect: null argument"); return TRAY_E_INVALID; }
    if (n == 0) return TRAY_OK;
    HIP_CHECK(hipSetDevice(s->device));
    TrayRay* d_r = nullptr;
    TrayHit* d_h = nullptr;
    HIP_CHECK(hipMalloc(&d_r, n * sizeof(TrayRay)));
    hipError_t e = hipMalloc(&d_h, n * sizeof(TrayHit));
    if (e == hipSuccess) e = hipMemcpy(d_r, rays, n * sizeof(TrayRay), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        // ANIM = 2: debug grids are sized by the item count, not by the transform cache, so the spline stacks are evaluated at every use
        if (s->deforming) hipLaunchKernelGGL(k_debug_intersect<3>, dim3((n + TR_BLOCK - 1) / TR_BLOCK), dim3(TR_BLOCK), s->stack_bytes, 0, s->dev, n, d_r, d_h);
        else if (s->animated) hipLaunchKernelGGL(k_debug_intersect<2>, dim3((n + TR_BLOCK - 1) / TR_BLOCK), dim3(TR_BLOCK), s->stack_bytes, 0, s->dev, n, d_r, d_h);
        else hipLaunchKernelGGL(k_debug_intersect<0>, dim3((n + TR_BLOCK - 1) / TR_BLOCK), dim3(TR_BLOCK), s->stack_bytes, 0, s->dev, n, d_r, d_h);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(hits, d_h, n * sizeof(TrayHit), hipMemcpyDeviceToHost);
    (void)hipFree(d_r);
    (void)hipFree(d_h);
    if (e != hipSuccess) { set_error(std::string("tray_debug_intersect: ") + hipGetErrorString(e)); return TRAY_E_DEVICE; }
    return TRAY_OK;
}

int tray_debug_sample_radiance(TrayDeviceScene* s, uint32_t n, const uint32_t* px, const uint32_t* py, const uint32_t* si,
                               uint32_t spp, uint64_t seed, float* out) {
    if (!s || !px || !py || !si || !out) { set_error("tray_debug_sample_radiance: null argument"); return TRAY_E_INVALID; }
    if (spp == 0 || (spp & (spp - 1)) != 0) { set_error("spp must be a power of two"); return TRAY_E_INVALID; }
    if (n == 0) return TRAY_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (px[i] >= s->dev.width || py[i] >= s->dev.height || si[i] >= spp) { set_error("tray_debug_sample_radiance: item out of range"); return TRAY_E_INVALID; }
    HIP_CHECK(hipSetDevice(s->device));
    uint32_t* d_in = nullptr;
    float* d_out = nullptr;
    HIP_CHECK(hipMalloc(&d_in, 3 * (size_t)n * sizeof(uint32_t)));
    hipError_t e = hipMalloc(&d_out, 8 * (size_t)n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, px, n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + n, py, n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in + 2 * (size_t)n, si, n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
        uint32_t kf = mix((uint32_t)seed + 0x9E3779B9u);
        kf = mix(kf ^ (uint32_t)(seed >> 32));
        kf = mix(kf + s->dev.frame);
        if (s->deforming)
            hipLaunchKernelGGL(k_debug_sample_radiance<3>, dim3((n + TR_BLOCK - 1) / TR_BLOCK), dim3(TR_BLOCK), s->stack_bytes, 0, s->dev, n, d_in, d_in + n, d_in + 2 * (size_t)n, spp, kf, d_out);
        else if (s->animated)
            hipLaunchKernelGGL(k_debug_sample_radiance<2>, dim3((n + TR_BLOCK - 1) / TR_BLOCK), dim3(TR_BLOCK), s->stack_bytes, 0, s->dev, n, d_in, d_in + n, d_in + 2 * (size_t)n, spp, kf, d_out);
        else
            hipLaunchKernelGGL(k_debug_sample_radiance<0>, dim3((n + TR_BLOCK - 1) / TR_BLOCK), dim3(TR_BLOCK), s->stack_bytes, 0, s->dev, n, d_in, d_in + n, d_in + 2 * (size_t)n, spp, kf, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, 8 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) { set_error(std::string("tray_debug_sample_radiance: ") + hipGetErrorString(e)); return TRAY_E_DEVICE; }
    return TRAY_OK;
}

int tray_debug_bsdf(TrayDeviceScene* s, uint32_t material_id, uint32_t flags, uint32_t n, const float* dirs, const float* u3, float* out) {
    if (!s || !dirs || !u3 || !out) { set_error("tray_debug_bsdf: null argument"); return TRAY_E_INVALID; }
    if (n == 0) return TRAY_OK;
    if (material_id >= s->n_materials) { set_error("tray_debug_bsdf: no such material"); return TRAY_E_INVALID; }
    HIP_CHECK(hipSetDevice(s->device));
    // private one-instance table carrying the requested material
    TrayInstance fake;
    std::memset(&fake, 0, sizeof fake);
    fake.material_id = material_id;
    TrayInstance* d_fake = nullptr;
    float *d_dirs = nullptr, *d_u = nullptr, *d_out = nullptr;
    HIP_CHECK(hipMalloc(&d_fake, sizeof fake));
    hipError_t e = hipMalloc(&d_dirs, 6 * (size_t)n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_u, 3 * (size_t)n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_out, 12 * (size_t)n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_fake, &fake, sizeof fake, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_dirs, dirs, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_u, u3, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        DevScene tmp = s->dev;
        tmp.instances = d_fake;
        hipLaunchKernelGGL(k_debug_bsdf, dim3((n + 63) / 64), dim3(64), 0, 0, tmp, flags, n, d_dirs, d_u, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, 12 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_fake); (void)hipFree(d_dirs); (void)hipFree(d_u); (void)hipFree(d_out);
    if (e != hipSuccess) { set_error(std::string("tray_debug_bsdf: ") + hipGetErrorString(e)); return TRAY_E_DEVICE; }
    return TRAY_OK;
}

}  // extern "C"
