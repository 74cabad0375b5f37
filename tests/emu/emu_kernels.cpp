// TEST INFRASTRUCTURE ONLY: the device source of libtrayhip.so compiled for the host (hip_emu.h) and driven one thread at a
// time. Entry points mirror what the library does around the same kernels (scene upload, pool fields, queues, launch geometry).
#include "hip_emu.h"
#include <sys/mman.h>   // (the sparse transform table of emu_render_tiles)
#ifdef TR_COOP_HIST   // tools/coop_histogram.py: [rays staged][lanes that asked] of every cooperative small-mesh test (dev_geom.h: mesh_leaf_coop)
namespace tr { unsigned long long tr_coop_hist[65 * 65]; }
extern "C" unsigned long long* emu_coop_hist(void) { return tr::tr_coop_hist; }
#endif

#ifdef TR_EMU_PROFILE   // divergence profile: see hip_emu.h; block barriers are ignored (they separate tiles, not stages)
#include <dlfcn.h>
#include <map>
#include <unordered_map>
#define NOINSTR __attribute__((no_instrument_function))
namespace hip_emu {
struct ProfAgg { uint64_t lane_calls = 0, wave_calls = 0, lanes = 0, segments = 0; };
struct ProfState {
    std::unordered_map<void*, uint32_t> lane;                 // calls of the running lane in its current segment
    std::unordered_map<void*, std::pair<uint32_t, uint64_t>> wave[MAX_THREADS / WAVE];   // fn -> (max over lanes, sum over lanes) in the wave's segment
    std::unordered_map<void*, uint32_t> wave_lanes[MAX_THREADS / WAVE];
    std::unordered_map<void*, ProfAgg> total;
    bool on = false;
};
static ProfState g_prof;
NOINSTR inline ProfState& prof() { return g_prof; }
NOINSTR void prof_lane_done(uint32_t wave) {
    ProfState& p = prof();
    if (!p.on) return;
    for (auto& kv : p.lane) {
        auto& w = p.wave[wave][kv.first];
        w.first = std::max(w.first, kv.second); w.second += kv.second;
        p.wave_lanes[wave][kv.first] += 1;
    }
    p.lane.clear();
}
NOINSTR void prof_wave_done(uint32_t wave) {
    ProfState& p = prof();
    if (!p.on) return;
    for (auto& kv : p.wave[wave]) {
        ProfAgg& a = p.total[kv.first];
        a.wave_calls += kv.second.first; a.lane_calls += kv.second.second; a.lanes += p.wave_lanes[wave][kv.first]; a.segments += 1;
    }
    p.wave[wave].clear(); p.wave_lanes[wave].clear();
}
}  // namespace hip_emu
extern "C" {
NOINSTR void __cyg_profile_func_enter(void* fn, void*) { hip_emu::ProfState& p = hip_emu::prof(); if (p.on && hip_emu::block().simt) ++p.lane[reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(fn) << 3) | (hip_emu::prof_phase_of[hip_emu::block().current & 1023u] & 7u))]; }
NOINSTR void __cyg_profile_func_exit(void*, void*) {}
NOINSTR void emu_profile_start(void) { hip_emu::ProfState& p = hip_emu::prof(); p.total.clear(); p.on = true; }
// writes "lane_calls wave_calls lanes segments symbol" lines; returns the number of functions
NOINSTR int emu_profile_dump(const char* path) {
    hip_emu::ProfState& p = hip_emu::prof();
    p.on = false;
    FILE* f = std::fopen(path, "w");
    if (!f) return -1;
    std::unordered_map<void*, hip_emu::ProfAgg> by_fn;   // the phases of a function, added up
    for (auto& kv : p.total) {
        hip_emu::ProfAgg& a = by_fn[reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(kv.first) >> 3)];
        a.lane_calls += kv.second.lane_calls; a.wave_calls += kv.second.wave_calls; a.lanes += kv.second.lanes; a.segments += kv.second.segments;
    }
    for (auto& kv : by_fn) {
        Dl_info info;
        const char* name = (dladdr(kv.first, &info) && info.dli_sname) ? info.dli_sname : "?";
        std::fprintf(f, "%llu %llu %llu %llu %s\n", (unsigned long long)kv.second.lane_calls, (unsigned long long)kv.second.wave_calls,
                     (unsigned long long)kv.second.lanes, (unsigned long long)kv.second.segments, name);
    }
    if (std::getenv("TR_PROFILE_PHASES"))   // the same rows per phase (TR_EMU_PHASE: the query passes LIGHT = 1, MIS = 2, PATH = 3; 0 = everything else), name suffixed "@phase"
        for (auto& kv : p.total) {
            Dl_info info;
            void* fn = reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(kv.first) >> 3);
            const char* name = (dladdr(fn, &info) && info.dli_sname) ? info.dli_sname : "?";
            std::fprintf(f, "%llu %llu %llu %llu %s@%u\n", (unsigned long long)kv.second.lane_calls, (unsigned long long)kv.second.wave_calls,
                         (unsigned long long)kv.second.lanes, (unsigned long long)kv.second.segments, name, (unsigned)(reinterpret_cast<uintptr_t>(kv.first) & 7u));
        }
    std::fclose(f);
    return (int)by_fn.size();
}
}
#endif

#include "../../tray_rust_amd/csrc/hip/kernels.hip"

namespace trayh { void set_error(const std::string&) {} }

namespace {

uint32_t g_retraced = 0;   // accumulated over the calls of this process; read and reset by emu_retraced()
uint32_t g_wf_deferred = 0;   // rays k_wf_trace_dyn handed to k_wf_trace_fallback; read and reset by emu_wf_deferred()

struct EmuScene {
    DevScene d{};
    std::vector<DevMaterial> mats;
    std::vector<tray::FlatLeaf> flat_leaves;
    std::vector<tray::FlatInst> flat_insts;
    std::vector<uint8_t> tri_leaf;
    tray::PairedTrees paired;   // the trees in device order, as tray_scene_create uploads them
    tray::QuadTrees quads;      // ... and as the wavefront traversal's 128-byte records
    std::vector<tray::QuadNode> all_quads;
    std::vector<tray::WfInst> wf_insts;
    std::vector<uint8_t> perm_pool;
    uint32_t retraced = 0;   // rays the flat loop handed to trace_bvh
    uint32_t depth = 0;   // traversal stack entries per lane, as tray_scene_create sizes them (two-level worst case, generous)
    uint32_t quad_words = 0;   // stack words per lane of the wavefront traversal (node entries are two words, up to three per record)
};

uint32_t bvh_depth(const TrayBvhNode* nodes, uint32_t n) {
    uint32_t best = 0;
    std::vector<std::pair<uint32_t, uint32_t>> st;
    if (n) st.push_back({0u, 1u});
    while (!st.empty()) {
        auto [idx, dep] = st.back();
        st.pop_back();
        best = std::max(best, dep);
        if (idx < n && nodes[idx].count == 0) { st.push_back({idx + 1, dep + 1}); st.push_back({nodes[idx].offset, dep + 1}); }
    }
    return best;
}

// the scene holds an AnimatedMesh: the ANIM = 3 instantiations run (tray_scene_create: TrayDeviceScene::deforming)
static bool deforming(const TrayFlatScene* f) {
    for (uint32_t i = 0; i < f->n_instances; ++i) if (f->instances[i].geom_type == TRAY_GEOM_ANIMATED_MESH) return true;
    return false;
}

void make_scene(const TrayFlatScene* f, EmuScene& e) {
    DevScene& d = e.d;
    if (!tray::pair_trees(f, e.paired)) throw std::runtime_error("BVH arrays do not describe trees");
    d.instances = f->instances; d.top_nodes = e.paired.top.data(); d.top_order = f->top_order; d.meshes = e.paired.meshes.data();
    d.mesh_nodes = e.paired.mesh.data(); d.tri_verts = f->tri_verts; d.tri_attrs = f->tri_attrs;
    tray::quad_trees(f, e.quads);
    e.all_quads = e.quads.mesh;   // one buffer, as tray_scene_create uploads it: the BVH<Triangle>s, then BVH<Instance>
    e.all_quads.insert(e.all_quads.end(), e.quads.top.begin(), e.quads.top.end());
    d.quads = reinterpret_cast<const float4*>(e.all_quads.data()); d.top_quad_first = (uint32_t)e.quads.mesh.size();
    tray::wf_inst_records(f, e.quads.mesh_first, e.wf_insts);
    d.wf_insts = e.wf_insts.data();
    e.mats.resize(f->n_materials);
    for (uint32_t i = 0; i < f->n_materials; ++i) e.mats[i] = lower_material(f->materials[i], f->merl_tables);
    d.materials = e.mats.data(); d.merl_data = f->merl_data; d.lights = f->lights;
    d.textures = f->n_textures ? f->textures : nullptr; d.tex_frames = f->tex_frames; d.tex_data = f->tex_data;
    d.filter_table = f->film.table; d.filter_x = f->film.table_x; d.filter_y = f->film.table_y;
    d.xf_levels = f->xf_levels; d.keyframes = f->keyframes; d.knots = f->knots; d.color_keys = f->color_keys;
    d.xf_cache = nullptr; d.moving_ids = nullptr; d.n_moving = 0; d.xf_cache_lanes = 0; d.xf_aos = 0; d.xf_table = 0; d.xf_stride = 0; d.xf_tab = nullptr; d.xf_tab_stride = 0; d.pad_tab = 0;
    d.n_instances = f->n_instances; d.n_lights = f->n_lights; d.min_depth = f->min_depth; d.max_depth = f->max_depth;
    d.width = f->film.width; d.height = f->film.height; d.frame = f->frame; d.film_rows = 0; d.coop_offset = 0; d.integrator = f->integrator;
    d.filter_w = f->film.filter_w; d.filter_h = f->film.filter_h; d.inv_w = f->film.inv_w; d.inv_h = f->film.inv_h;
    d.fpw = f->film.filter_pixel_w; d.fph = f->film.filter_pixel_h;
    d.camera_p = &f->camera;
    e.perm_pool.resize(TR_PERM_BYTES);
    perm_pool_build(f->max_depth + 1u, e.perm_pool.data());
    d.perm_pool = e.perm_pool.data();
    tray::flat_loop_gates(f, e.paired, TR_COOP_MAX_TRIS, e.flat_leaves, e.flat_insts, e.tri_leaf);
    d.flat_leaves = e.flat_leaves.data(); d.flat_insts = e.flat_insts.data(); d.n_flat_leaves = (uint32_t)e.flat_leaves.size(); d.tri_leaf = e.tri_leaf.data();
    d.retraced = &g_retraced;
    d.mesh_keys = f->mesh_keys; d.key_times = f->key_times;
    uint32_t mesh_depth = 0;
    for (uint32_t m = 0; m < f->n_meshes; ++m) mesh_depth = std::max(mesh_depth, bvh_depth(f->mesh_nodes + f->meshes[m].node_offset, f->meshes[m].node_count));
    e.depth = mesh_depth + bvh_depth(f->top_nodes, f->n_top_nodes) + 8u;
    e.quad_words = 2u * (e.quads.top_pend + e.quads.mesh_pend) + 34u;
}

using hip_emu::launch;
using hip_emu::launch_simt;

// what tray_scene_create decides per scene: lobe feature set of the kernels, row-binned film, cooperative small-mesh test
int feature_set(const EmuScene& e) {
    int feat = FEAT_NONE;
    for (const DevMaterial& dm : e.mats)
        for (uint32_t l = 0; l < dm.n_lobes && l < 2u; ++l) {
            const uint32_t k = dm.lobe[l].kind;
            if (k == LB_MERL) feat |= FEAT_MERL;
            if (k == LB_MF_TRANS) feat |= FEAT_MF_TRANS;
            if (k == LB_SPEC_REFL_DIEL || k == LB_SPEC_REFL_COND || k == LB_SPEC_TRANS || k == LB_TS_COND) feat |= FEAT_SPEC;
        }
    for (const DevMaterial& dm : e.mats) if (dm.textured || dm.microfacet == TRAY_MF_GGX) return FEAT_ALL | FEAT_TEX;
    return (feat & FEAT_MF_TRANS) ? FEAT_ALL : feat;
}
bool film_rows_ok(const TrayFlatScene* f) {
    bool ok = f->film.separable != 0 && f->film.filter_h == 2.0f && f->film.inv_h == 0.5f && f->film.filter_pixel_h == 4;
    for (int y = 0; ok && y < TRAY_FILTER_TABLE_SIZE; ++y)
        for (int x = 0; x < TRAY_FILTER_TABLE_SIZE; ++x)
            if (f->film.table[y * TRAY_FILTER_TABLE_SIZE + x] != f->film.table_x[x] * f->film.table_y[y]) { ok = false; break; }
    return ok;
}
uint32_t key_frame_host(uint64_t seed, uint32_t frame) {   // as launch_tiles computes it
    auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
    uint32_t kf = mix((uint32_t)seed + 0x9E3779B9u);
    kf = mix(kf ^ (uint32_t)(seed >> 32));
    return mix(kf + frame);
}

}  // namespace

extern "C" {

// k_debug_intersect<0> on n rays (what tray_debug_intersect launches)
int emu_debug_intersect(const TrayFlatScene* f, uint32_t n, const TrayRay* rays, TrayHit* hits) {
    EmuScene e;
    make_scene(f, e);
    if (deforming(f)) launch((n + TR_BLOCK - 1) / TR_BLOCK, TR_BLOCK, [&] { k_debug_intersect<3>(e.d, n, rays, hits); });
    else launch((n + TR_BLOCK - 1) / TR_BLOCK, TR_BLOCK, [&] { k_debug_intersect<0>(e.d, n, rays, hits); });
    return 0;
}

// k_debug_sample_radiance<0 | 2> for n (pixel, sample) items (what tray_debug_sample_radiance launches; ANIM = 2 evaluates the
// spline stacks at every use, as the library does for its debug kernels on moving scenes)
int emu_debug_sample_radiance(const TrayFlatScene* f, uint32_t n, const uint32_t* px, const uint32_t* py, const uint32_t* si, uint32_t spp,
                              uint64_t seed, float* out) {
    EmuScene e;
    make_scene(f, e);
    const uint32_t kf = key_frame_host(seed, e.d.frame);
    bool moving = f->camera.animated != 0;
    for (uint32_t t_ = 0; t_ < f->n_textures; ++t_) moving = moving || f->textures[t_].n_frames >= 2u;   // animated_image needs ray.time (tray_scene_create)
    for (uint32_t i = 0; i < f->n_instances; ++i) moving = moving || f->instances[i].animated != 0 || f->instances[i].emis_count >= 2;
    if (deforming(f)) launch((n + TR_BLOCK - 1) / TR_BLOCK, TR_BLOCK, [&] { k_debug_sample_radiance<3>(e.d, n, px, py, si, spp, kf, out); });
    else if (moving) launch((n + TR_BLOCK - 1) / TR_BLOCK, TR_BLOCK, [&] { k_debug_sample_radiance<2>(e.d, n, px, py, si, spp, kf, out); });
    else launch((n + TR_BLOCK - 1) / TR_BLOCK, TR_BLOCK, [&] { k_debug_sample_radiance<0>(e.d, n, px, py, si, spp, kf, out); });
    return 0;
}

// launch_sampler (kernels.hip) for the tiles given: thread_work with sampler::Uniform / sampler::Adaptive. batch_tiles bounds the tiles per
// batch (0 = all at once), so that tests can see batches add up. stats_out: samples, vertices, rays.
int emu_render_sampler(const TrayFlatScene* f, const uint32_t* tiles_xy, uint32_t tile_count, uint32_t kind, uint32_t min_spp, uint32_t max_spp,
                       uint64_t seed, float* rgbw, uint32_t batch_tiles, unsigned long long* stats_out) {
    EmuScene e;
    make_scene(f, e);
    const uint32_t kf = key_frame_host(seed, e.d.frame);
    bool moving = f->camera.animated != 0;
    for (uint32_t t_ = 0; t_ < f->n_textures; ++t_) moving = moving || f->textures[t_].n_frames >= 2u;
    for (uint32_t i = 0; i < f->n_instances; ++i) moving = moving || f->instances[i].animated != 0 || f->instances[i].emis_count >= 2;
    std::vector<uint2> tiles(tile_count);
    for (uint32_t i = 0; i < tile_count; ++i) tiles[i] = make_uint2(tiles_xy[2 * i], tiles_xy[2 * i + 1]);
    DevStats stats;
    std::memset(&stats, 0, sizeof stats);
    auto round_up = [](uint32_t v) { uint32_t p = 1; while (p < v && p < 0x80000000u) p <<= 1; return p; };
    SamplerPass sp{};
    sp.kind = kind; sp.min_spp = round_up(min_spp); sp.max_spp = round_up(max_spp);
    uint32_t rounds = 1;
    if (kind == TRAY_SAMPLER_ADAPTIVE) {
        if (sp.max_spp < sp.min_spp) return -1;
        sp.step = round_up((sp.max_spp - sp.min_spp) / 5u);
        while (sp.min_spp + (rounds - 1u) * sp.step < sp.max_spp) ++rounds;
        sp.lum_cap = sp.min_spp + (rounds - 1u) * sp.step;
    } else if (kind == TRAY_SAMPLER_UNIFORM) { sp.min_spp = sp.max_spp = 1u; sp.step = 1u; sp.lum_cap = 0u; }
    else if (kind == TRAY_SAMPLER_LOW_DISCREPANCY) { sp.max_spp = sp.min_spp; sp.step = 1u; sp.lum_cap = 0u; }   // (scenes with an AnimatedMesh: min_spp = the render's spp)
    else return -1;
    const uint32_t batch = batch_tiles ? std::min(batch_tiles, std::max(tile_count, 1u)) : std::max(tile_count, 1u);
    std::vector<uint32_t> px_state((size_t)batch * 64u);
    std::vector<float> px_avg((size_t)batch * 64u), px_lum((size_t)batch * 64u * std::max(sp.lum_cap, 1u));
    const uint32_t chunk = tile_count ? tile_count : 1u;
    int rc = 0;   // (the window is shared by the block's threads: a SIMT emulation, as for k_path_tiles)
    for (uint32_t item0 = 0; item0 < tile_count; item0 += batch) {
        const uint32_t n_items = std::min(batch, tile_count - item0), n_px = n_items * 64u;
        std::fill(px_state.begin(), px_state.end(), 0u); std::fill(px_avg.begin(), px_avg.end(), 0.0f);
        for (uint32_t j = 0; j < rounds; ++j) {
            sp.pass = j;
            sp.count = kind == TRAY_SAMPLER_ADAPTIVE ? (j == 0u ? sp.min_spp : sp.step) : sp.min_spp;
            sp.taken = kind == TRAY_SAMPLER_ADAPTIVE ? sp.min_spp + j * sp.step : 0u;
            sp.before = j == 0u ? 0u : sp.min_spp + (j - 1u) * sp.step;
            const uint32_t per_tile = 64u * sp.count;   // launch_sampler's groups of tiles
            uint32_t group = std::max(1u, std::min<uint32_t>(SP_GROUP_MAX, 4096u / per_tile));
            if (const char* ge = getenv("TRAYHIP_SAMPLER_GROUP")) group = (uint32_t)std::max(1, std::min(SP_GROUP_MAX, atoi(ge)));   // (tests: ragged groups)
            const uint32_t grid = (n_items + group - 1u) / group;
#define EMU_SAMPLER_PASS(A, F) rc = launch_simt(grid, TR_BLOCK, [&] { k_sampler_pass<A, F>(e.d, tiles.data(), item0, n_items, chunk, 1u, kf, sp, px_state.data(), px_lum.data(), rgbw, &stats, group); })
            const bool lean = feature_set(e) == FEAT_NONE && f->integrator != TRAY_INTEGRATOR_WHITTED;   // launch_sampler's choice of the instantiation
            if (deforming(f)) { if (lean) EMU_SAMPLER_PASS(3, FEAT_NONE); else EMU_SAMPLER_PASS(3, FEAT_ALL | FEAT_TEX); }
            else if (moving) { if (lean) EMU_SAMPLER_PASS(2, FEAT_NONE); else EMU_SAMPLER_PASS(2, FEAT_ALL | FEAT_TEX); }
            else { if (lean) EMU_SAMPLER_PASS(0, FEAT_NONE); else EMU_SAMPLER_PASS(0, FEAT_ALL | FEAT_TEX); }
#undef EMU_SAMPLER_PASS
            if (rc != 0) return -3;
            if (kind == TRAY_SAMPLER_ADAPTIVE)
                launch((n_px + TR_BLOCK - 1) / TR_BLOCK, TR_BLOCK, [&] { k_sampler_decide(n_px, sp, px_state.data(), px_avg.data(), px_lum.data()); });
        }
    }
    if (stats_out) { stats_out[0] = stats.samples; stats_out[1] = stats.vertices; stats_out[2] = stats.rays; }
    return 0;
}

// Debugging aid: the loop of k_debug_sample_radiance<0> for ONE sample with the lane state printed after every vertex
int emu_trace_sample(const TrayFlatScene* f, uint32_t px, uint32_t py, uint32_t si, uint32_t spp, uint64_t seed) {
    EmuScene e;
    make_scene(f, e);
    const uint32_t kf = key_frame_host(seed, e.d.frame);
    hip_emu::launch(1, 1, [&] {
        const DevScene& sc = e.d;
        TR_DYN_LDS(uint32_t, s_stack);
        Counters cnt; cnt.rays = 0; cnt.vertices = 0;
        const uint32_t kp = key_pixel(kf, py * sc.width + px);
        float sx, sy, t;
        pixel_sample(kp, si, spp, px, py, sx, sy, t);
        Lane ln;
        lane_start_sample(ln, camera_ray<0>(sc, sx, sy, t), key_sample(kp, si));
        while (ln.flags & LF_ALIVE) {
            for (int stage = 0; stage < 3; ++stage) {
                const bool alive = (ln.flags & LF_ALIVE) != 0u;
                const bool want_ray = alive && (stage == 0 || (stage == 1 && (ln.flags & LF_SHADOW)) || (stage == 2 && (ln.flags & LF_MIS)));
                TraceResult tr_; tr_.hit = false; tr_.rec.t = 0.0f; tr_.rec.inst = 0xffffffffu; tr_.rec.prim = 0u; tr_.rec.b1 = 0.0f; tr_.rec.b2 = 0.0f;
                if (want_ray) { const Ray r = stage == 0 ? stage_a_ray(ln) : (stage == 1 ? stage_b_ray(ln) : stage_c_ray(ln)); tr_ = trace<0>(&e.d, s_stack, r, stage == 1, want_ray); }
                if (!alive) continue;
                if (stage == 0) { if (tr_.hit) vertex_begin<0>(sc, ln, tr_.rec, cnt); else ln.flags &= ~LF_ALIVE; if (tr_.hit) std::fprintf(stderr, "device  bounce %u inst %u", ln.bounce, tr_.rec.inst); }
                else if (stage == 1) {
                    std::fprintf(stderr, "\ndevice    light sample: pdf %.9g li %.9g %.9g %.9g shadow-ray %d occluded %d w_i %.9g %.9g %.9g\n", ln.pdf_l, ln.li.x, ln.li.y, ln.li.z,
                                 (int)((ln.flags & LF_SHADOW) != 0), (int)tr_.hit, ln.wi_l.x, ln.wi_l.y, ln.wi_l.z);
                    vertex_queries<0, FEAT_ALL>(sc, ln, tr_.hit);
                    std::fprintf(stderr, "device    after queries: direct %.9g %.9g %.9g mis %d (cos, w, pdf) %.9g %.9g %.9g f %.9g %.9g %.9g\n", ln.direct.x, ln.direct.y, ln.direct.z,
                                 (int)((ln.flags & LF_MIS) != 0), ln.li.x, ln.li.y, ln.li.z, ln.mis_f.x, ln.mis_f.y, ln.mis_f.z);
                }
                else {
                    const f3 tv = ln.t_vertex;
                    const uint32_t b = ln.bounce;
                    const bool cont = vertex_end<0>(sc, ln, tr_.hit, tr_.rec);
                    std::fprintf(stderr, " li %.9g %.9g %.9g  throughput %.9g %.9g %.9g  illum %.9g %.9g %.9g\n", ln.direct.x, ln.direct.y, ln.direct.z, tv.x, tv.y, tv.z,
                                 ln.illum.x, ln.illum.y, ln.illum.z);
                    (void)b;
                    if (!cont) ln.flags &= ~LF_ALIVE;
                }
            }
        }
    });
    return 0;
}

// k_debug_bsdf: BSDF::eval / pdf / sample of one material on the canonical frame (what tray_debug_bsdf launches)
int emu_debug_bsdf(const TrayFlatScene* f, uint32_t material_id, uint32_t flags, uint32_t n, const float* dirs, const float* u3, float* out) {
    if (material_id >= f->n_materials) return -1;
    EmuScene e;
    make_scene(f, e);
    TrayInstance fake;
    std::memset(&fake, 0, sizeof fake);
    fake.material_id = material_id;
    e.d.instances = &fake;
    launch((n + 63) / 64, 64, [&] { k_debug_bsdf(e.d, flags, n, dirs, u3, out); });
    return 0;
}

// One stage of the wavefront traversal over n rays, through the pool fields and the queue the stage kernels use:
//   k_wf_trace_dyn; stage 0 = A (closest hit from
//   F_O / F_D), 1 = B (any hit on the segment F_P + t F_AUX, t in (0.001, 0.999)), 2 = C (closest hit from F_P along F_AUX).
// lds_depth < full depth exercises the HBM overflow part of the stacks. Output per ray: hit flag, t, inst, prim, b1, b2.
int emu_wf_trace(const TrayFlatScene* f, int stage, uint32_t n, const TrayRay* rays, uint32_t lds_depth, uint32_t blocks,
                 uint32_t* hit, float* t, uint32_t* inst, uint32_t* prim, float* b1, float* b2) {
    EmuScene e;
    make_scene(f, e);
    const uint32_t n_slots = (n + TR_BLOCK - 1) / TR_BLOCK * TR_BLOCK;
    std::vector<float> pool_data((size_t)F_COUNT * n_slots, 0.0f);
    WfPool pool{pool_data.data(), n_slots, wf_seg_cap(n_slots / TR_BLOCK)};
    std::vector<uint32_t> queue((size_t)WF_SEGS * pool.seg_cap * WF_RAY_WORDS), qctl(WF_QCTL_WORDS, 0u);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t s = n - 1u - i;   // any order inside a segment: slots keep their place, only indices travel
        const uint32_t seg = (s / TR_BLOCK) & (WF_SEGS - 1u);   // the segment of the slot's chunk (wavefront.h)
        const f3 o = mk(rays[s].o[0], rays[s].o[1], rays[s].o[2]), dd = mk(rays[s].d[0], rays[s].d[1], rays[s].d[2]);
        wf_put_ray(queue.data(), (size_t)seg * pool.seg_cap + qctl[seg * WF_SEG_STRIDE + stage]++, s, o, dd,
                   LF_ALIVE | (stage == 0 && rays[s].min_t == 0.0f ? WF_CAMERA_RAY : 0u));   // an entry of a ray queue is the ray (wavefront.h)
        pu(pool, F_FLAGS, s) = LF_ALIVE;   // (the traversal kernels and the fallback kernel read the queue entry alone)
    }
    const uint32_t full = e.quad_words;
    if (lds_depth == 0 || lds_depth > full) lds_depth = full;
    std::vector<uint32_t> overflow((size_t)(full + 64u) * blocks * TR_BLOCK, 0u);
    std::vector<uint32_t> fallback((size_t)n_slots * WF_RAY_WORDS, 0u);   // records of the deferred rays (on the device: the buffer of a ray queue that is idle during the stage)
    std::vector<DevStats> stats(WF_STAT_SLOTS);
    std::memset(stats.data(), 0, stats.size() * sizeof(DevStats));
    // (the traversal, then the rays it handed to the reference's binary traversal: wf_round of kernels.hip)
#define EMU_TRACE(S) do { launch(blocks, TR_BLOCK, [&] { k_wf_trace_dyn<S, 0>(e.d, pool, queue.data(), qctl.data(), stats.data(), lds_depth, overflow.data(), fallback.data(), 0u); }); \
                          launch(2, TR_BLOCK, [&] { k_wf_trace_fallback<S, 0>(e.d, pool, qctl.data(), fallback.data(), 0u); }); } while (0)
    if (stage == 0) EMU_TRACE(0); else if (stage == 1) EMU_TRACE(1); else EMU_TRACE(2);
#undef EMU_TRACE
    g_wf_deferred += qctl[WF_FB_WORD + stage];
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t fl = pu(pool, F_FLAGS, s);
        hit[s] = stage == 0 ? (fl & WF_HIT_A) != 0u : (stage == 1 ? (fl & WF_OCCLUDED) != 0u : (fl & WF_HIT_C) != 0u);
        t[s] = pf(pool, F_REC_T, s); inst[s] = pu(pool, F_REC_INST, s); prim[s] = pu(pool, F_REC_PRIM, s);
        b1[s] = pf(pool, F_REC_B1, s); b2[s] = pf(pool, F_REC_B2, s);
    }
    return 0;
}

// TRAYHIP_EMU_XF_TABLE=1: the frame's transform table (device_api.hip: xf_table_prepare, k_xf_table_build) under the emulated kernels. The table of all
// 2^24 shutter-time indices is 2 GB per moving instance: here it is a sparse mapping whose records exist for the indices the rendered (pixel, sample)
// pairs draw -- evaluated with k_xf_table_build's expressions --, which are the only ones the kernels read.
struct SparseXfTable {
    float* data = nullptr;
    size_t bytes = 0;
    uint32_t stride = 0;
    ~SparseXfTable() { if (data) munmap(data, bytes); }
    bool build(const TrayFlatScene* f, uint32_t frame, const uint32_t* moving_ids, uint32_t n_moving, const uint32_t* tiles_xy, uint32_t tile_count, uint32_t spp, uint64_t seed) {
        stride = n_moving + (f->camera.animated ? 1u : 0u);
        if (!stride) return true;
        bytes = ((size_t)1 << 24) * stride * TR_XF_REC * sizeof(float);
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) return false;
        data = static_cast<float*>(p);
        std::vector<uint8_t> have((size_t)1 << 21, 0u);   // one bit per index
        const uint32_t kf = key_frame_host(seed, frame);
        const TrayCamera& c = f->camera;
        for (uint32_t i = 0; i < tile_count; ++i)
            for (uint32_t pix = 0; pix < 64u; ++pix) {
                const uint32_t px = tiles_xy[2 * i] * 8u + (pix & 7u), py = tiles_xy[2 * i + 1] * 8u + (pix >> 3);   // (a queue entry is a tile's place in units of tiles)
                if (px >= f->film.width || py >= f->film.height) continue;
                const uint32_t kp = key_pixel(kf, py * f->film.width + px);
                for (uint32_t s = 0; s < spp; ++s) {
                    float x, y, t;
                    pixel_sample(kp, s, spp, px, py, x, y, t);
                    const uint32_t index = xf_time_index(t);
                    if (have[index >> 3] & (1u << (index & 7u))) continue;
                    have[index >> 3] |= (uint8_t)(1u << (index & 7u));
                    const float frame_time = (c.shutter_close - c.shutter_open) * xf_index_time(index) + c.shutter_open;   // (k_xf_table_build)
                    for (uint32_t m = 0; m < stride; ++m) {
                        uint32_t first, count;
                        if (m < n_moving) { const TrayInstance& in = f->instances[moving_ids[m]]; first = in.xf_first; count = in.xf_count; }
                        else { first = c.xf_first; count = c.xf_count; }
                        eval_xform_stack(f->xf_levels, f->keyframes, f->knots, first, count, frame_time, data + ((size_t)index * stride + m) * TR_XF_REC);
                    }
                }
            }
        return true;
    }
};

// The tile worker itself: k_path_tiles<0, FEAT> over `tile_count` tiles of the given Morton queue, launched the way
// launch_tiles does (feature set, row-binned film and cooperative small-mesh test chosen as tray_scene_create chooses them),
// as a SIMT emulation: 256 fibers per workgroup, wave intrinsics and barriers are rendezvous. rgbw is accumulated into.
// coop / film_rows: -1 = as the library decides, 0 = off. Returns 0, or -3 if a rendezvous could not complete.
// shard / n_shards / chunk_tiles: the launch tray_render_shard_device makes for one rank (chunks shard, shard + n_shards, ... of chunk_tiles
// tiles each); n_shards = 0 renders the whole queue given.
int emu_render_tiles(const TrayFlatScene* f, const uint32_t* tiles_xy, uint32_t tile_count, uint32_t spp, uint64_t seed, float* rgbw,
                     uint32_t blocks, int coop, int film_rows, unsigned long long* stats_out, uint32_t shard, uint32_t n_shards, uint32_t chunk_tiles) {
    EmuScene e;
    make_scene(f, e);
    bool moving = f->camera.animated != 0;
    for (uint32_t t_ = 0; t_ < f->n_textures; ++t_) moving = moving || f->textures[t_].n_frames >= 2u;   // animated_image needs ray.time (tray_scene_create)
    for (uint32_t i = 0; i < f->n_instances; ++i) moving = moving || f->instances[i].animated != 0 || f->instances[i].emis_count >= 2;
    if (f->n_instances > TR_FLAT_MAX && !moving) return -4;   // the library runs the wavefront schedule for those
    uint32_t n_moving = 0;
    for (uint32_t i = 0; i < f->n_instances; ++i) if (f->instances[i].animated) ++n_moving;
    std::vector<uint32_t> moving_ids(std::max(n_moving, 1u), 0u);
    std::vector<float> xf_cache;
    if (moving && n_moving) {   // per-path transform cache, one column per thread of the grid (tray_scene_create)
        for (uint32_t i = 0; i < f->n_instances; ++i)
            if (f->instances[i].animated && f->instances[i].moving_slot < n_moving) moving_ids[f->instances[i].moving_slot] = i;
        xf_cache.assign((size_t)n_moving * TR_XF_WORDS * blocks * TR_BLOCK, 0.0f);
        e.d.xf_cache = xf_cache.data(); e.d.moving_ids = moving_ids.data(); e.d.n_moving = n_moving; e.d.xf_stride = n_moving; e.d.xf_cache_lanes = blocks * TR_BLOCK;
    }
    SparseXfTable table;   // TRAYHIP_EMU_XF_TABLE=1: the fill of the cache columns FROM THE TABLE (dev_geom.h: xf_cache_fill_wave, camera_ray) runs in the emulation
    if (moving && getenv("TRAYHIP_EMU_XF_TABLE") && atoi(getenv("TRAYHIP_EMU_XF_TABLE")) != 0) {
        if (!table.build(f, e.d.frame, moving_ids.data(), n_moving, tiles_xy, tile_count, spp, seed)) return -5;
        if (table.data) { e.d.moving_ids = moving_ids.data(); e.d.xf_tab = table.data; e.d.xf_tab_stride = table.stride; }
    }
    e.d.film_rows = (film_rows != 0 && film_rows_ok(f)) ? 1u : 0u;
    uint32_t stack_words = e.depth * TR_BLOCK;
    bool small_mesh = false;
    for (uint32_t m = 0; m < f->n_meshes; ++m) small_mesh = small_mesh || f->meshes[m].tri_count <= TR_COOP_MAX_TRIS;
    if (coop != 0 && small_mesh && f->n_instances <= TR_FLAT_MAX) { e.d.coop_offset = stack_words; stack_words += (TR_BLOCK / 64) * TR_COOP_WORDS; }
    if (e.d.film_rows) { e.d.win_offset = 0u; stack_words = std::max(stack_words, 4u * WIN_PLANE); }   // tray_scene_create: the film window over the stacks ...
    else { e.d.win_offset = stack_words; stack_words += 4u * WIN_PLANE; }                               // ... or in its own region
    std::vector<uint2> tiles(tile_count);
    for (uint32_t i = 0; i < tile_count; ++i) tiles[i] = make_uint2(tiles_xy[2 * i], tiles_xy[2 * i + 1]);
    // work-item mapping of launch_tiles: item w -> queue entry (w / chunk) * chunk_stride * chunk + (w % chunk), from tile_start on
    uint32_t tile_start = 0, work = tile_count, chunk = tile_count ? tile_count : 1u, chunk_stride = 1u;
    if (n_shards) {   // tray_render_shard_device
        const uint32_t n_chunks = (tile_count + chunk_tiles - 1) / chunk_tiles;
        const uint32_t my_chunks = shard < n_chunks ? (n_chunks - shard + n_shards - 1) / n_shards : 0;
        if (my_chunks == 0) { if (stats_out) stats_out[0] = stats_out[1] = stats_out[2] = stats_out[3] = 0; return 0; }
        const uint32_t last_chunk = shard + (my_chunks - 1) * n_shards;
        uint32_t tail = tile_count - last_chunk * chunk_tiles;
        if (tail > chunk_tiles) tail = chunk_tiles;
        tile_start = shard * chunk_tiles; work = (my_chunks - 1) * chunk_tiles + tail; chunk = chunk_tiles; chunk_stride = n_shards;
    }
    uint32_t counter = 0;
    DevStats stats;
    std::memset(&stats, 0, sizeof stats);
    const uint32_t kf = key_frame_host(seed, e.d.frame);
    const int feat = feature_set(e);
    uint32_t levels = 1u;   // launch_tiles' rule: tiles are cut into progressive sample slices when there are few of them per workgroup
    { const bool small = work < 12u * blocks; const uint32_t most = small ? 5u : 3u, least = small ? 64u : 256u; while (levels < most && (spp >> levels) >= least) ++levels; }
    if (const char* e_ = getenv("TRAYHIP_TILE_SLICES")) { levels = 1u; const uint32_t want = (uint32_t)std::max(1, atoi(e_)); while (levels < want && (spp >> levels) >= 1u) ++levels; }
    int rc;
    // tray_scene_create: the instantiation with mis_ray_filter for scenes with a sphere light or specular lobes
    bool light_filter = (feat & FEAT_SPEC) != 0;
    for (uint32_t l = 0; l < f->n_lights; ++l)
        if (f->instances[f->lights[l]].kind != TRAY_INST_POINT_EMITTER && f->instances[f->lights[l]].geom_type == TRAY_GEOM_SPHERE) light_filter = true;
#define EMU_TILES_L(A, F, L) launch_simt(blocks, TR_BLOCK, [&] { k_path_tiles<A, F, TRAY_INTEGRATOR_PATH, L>(e.d, tiles.data() + tile_start, work, chunk, chunk_stride, spp, kf, levels, rgbw, &counter, &stats); }, (size_t)stack_words * 4)
#define EMU_TILES(F) rc = moving ? (light_filter ? EMU_TILES_L(1, F, true) : EMU_TILES_L(1, F, false)) : (light_filter ? EMU_TILES_L(0, F, true) : EMU_TILES_L(0, F, false))
#define EMU_WHITTED(A) launch_simt(blocks, TR_BLOCK, [&] { k_path_tiles<A, FEAT_ALL | FEAT_TEX, TRAY_INTEGRATOR_WHITTED>(e.d, tiles.data() + tile_start, work, chunk, chunk_stride, spp, kf, levels, rgbw, &counter, &stats); }, (size_t)stack_words * 4)
    if (e.d.integrator == TRAY_INTEGRATOR_WHITTED) rc = moving ? EMU_WHITTED(1) : EMU_WHITTED(0);   // launch_tiles: one instantiation per ANIM
    else if (feat == FEAT_NONE) EMU_TILES(FEAT_NONE);
    else if (feat == FEAT_MERL) EMU_TILES(FEAT_MERL);
    else if (feat == FEAT_SPEC) EMU_TILES(FEAT_SPEC);
    else if (feat == (FEAT_MERL | FEAT_SPEC)) EMU_TILES(FEAT_MERL | FEAT_SPEC);
    else if (feat == (FEAT_ALL | FEAT_TEX)) EMU_TILES(FEAT_ALL | FEAT_TEX);
    else EMU_TILES(FEAT_ALL);
#undef EMU_TILES
#undef EMU_TILES_L
#undef EMU_WHITTED
    if (stats_out) { stats_out[0] = stats.samples; stats_out[1] = stats.vertices; stats_out[2] = stats.rays; stats_out[3] = (unsigned long long)feat; }
    return rc;
}

// The wavefront schedule (launch_wavefront + wf_round of kernels.hip): rounds of k_wf_advance -> k_wf_regen -> trace A ->
// k_wf_begin -> trace B -> k_wf_query -> trace C over an HBM-style path pool until every tile is done. All kernels run as SIMT
// emulations (`trace` is kept in the signature: 0 = k_wf_trace_dyn, the only traversal kernel). Moving scenes run the ANIM = 1 kernels with
// the per-slot transform cache.
// n_chunks = 256-slot chunks of the pool (<= tile_count); lds_depth as in emu_wf_trace.
int emu_render_wavefront(const TrayFlatScene* f, const uint32_t* tiles_xy, uint32_t tile_count, uint32_t spp, uint64_t seed, float* rgbw,
                         int trace, uint32_t n_chunks, uint32_t trace_blocks, uint32_t lds_depth, unsigned long long* stats_out) {
    EmuScene e;
    make_scene(f, e);
    bool moving = f->camera.animated != 0;
    for (uint32_t t_ = 0; t_ < f->n_textures; ++t_) moving = moving || f->textures[t_].n_frames >= 2u;   // animated_image needs ray.time (tray_scene_create)
    uint32_t n_moving = 0;
    for (uint32_t i = 0; i < f->n_instances; ++i) {
        moving = moving || f->instances[i].animated != 0 || f->instances[i].emis_count >= 2;
        if (f->instances[i].animated) ++n_moving;
    }
    e.d.film_rows = film_rows_ok(f) ? 1u : 0u;
    // launch_wavefront's rule: tiles are cut into slices of their samples while the pool has more chunks than work items (k_wf_advance)
    uint32_t slice_shift = 0u;
    while ((1u << (slice_shift + 1u)) <= 16u && ((uint64_t)tile_count << (slice_shift + 1u)) <= n_chunks && (spp >> (slice_shift + 1u)) >= 16u) ++slice_shift;
    if (const char* sl = getenv("TRAYHIP_WF_SLICES")) { slice_shift = 0u; while ((2u << slice_shift) <= (uint32_t)std::max(1, atoi(sl)) && (2u << slice_shift) <= 16u && (spp >> (slice_shift + 1u)) >= 1u) ++slice_shift; }
    const uint32_t n_items = tile_count << slice_shift;
    n_chunks = std::max(1u, std::min(n_chunks, n_items));
    const uint32_t n_slots = n_chunks * TR_BLOCK, n_active = n_slots;
    std::vector<float> pool_data((size_t)F_COUNT * n_slots, 0.0f);
    WfPool pool{pool_data.data(), n_slots, wf_seg_cap(n_chunks)};
    std::vector<uint32_t> moving_ids(std::max(n_moving, 1u), 0u);
    std::vector<float> xf_cache;
    if (moving && n_moving) {   // per-path transform cache, one column per pool slot (tray_scene_create)
        for (uint32_t i = 0; i < f->n_instances; ++i)
            if (f->instances[i].animated && f->instances[i].moving_slot < n_moving) moving_ids[f->instances[i].moving_slot] = i;
        xf_cache.assign((size_t)n_moving * TR_XF_REC * n_slots, 0.0f);
        e.d.xf_cache = xf_cache.data(); e.d.moving_ids = moving_ids.data(); e.d.n_moving = n_moving; e.d.xf_stride = n_moving; e.d.xf_cache_lanes = n_slots; e.d.xf_aos = 1u;
    }
    SparseXfTable table;   // TRAYHIP_EMU_XF_TABLE=1: the stage kernels index the frame's table by the path's time index (device_api.hip: xf_table_prepare's wavefront branch)
    if (moving && getenv("TRAYHIP_EMU_XF_TABLE") && atoi(getenv("TRAYHIP_EMU_XF_TABLE")) != 0) {
        if (!table.build(f, e.d.frame, moving_ids.data(), n_moving, tiles_xy, tile_count, spp, seed)) return -5;
        if (table.data) {
            e.d.moving_ids = moving_ids.data(); e.d.n_moving = n_moving; e.d.xf_tab = table.data; e.d.xf_tab_stride = table.stride;
            e.d.xf_cache = table.data; e.d.xf_table = 1u; e.d.xf_aos = 1u; e.d.xf_stride = table.stride;
        }
    }
    std::vector<WfChunk> chunks(n_chunks, WfChunk{WF_TILE_NEED, 0u});
    std::vector<float> bins((size_t)n_chunks * ROWBIN_SIZE, 0.0f);
    const size_t q_cap = (size_t)WF_SEGS * pool.seg_cap;
    const uint32_t q_blocks = (n_chunks + WF_SEGS - 1u) / WF_SEGS * WF_SEGS;   // grid of the one-thread-per-entry kernels (launch_wavefront)
    std::vector<uint32_t> queues((3 * WF_RAY_WORDS + 1) * q_cap + WF_QCTL_WORDS, 0u);   // ray queues A, B, C (the rays themselves), regeneration queue (slot indices)
    uint32_t* const qa = queues.data(), * const qb = qa + WF_RAY_WORDS * q_cap, * const qc = qb + WF_RAY_WORDS * q_cap, * const qr = qc + WF_RAY_WORDS * q_cap, * const qctl = qr + q_cap;
    uint32_t counters[2] = {0u, 0u};
    std::vector<DevStats> stats(WF_STAT_SLOTS);
    std::memset(stats.data(), 0, stats.size() * sizeof(DevStats));
    std::vector<uint2> tiles(tile_count);
    for (uint32_t i = 0; i < tile_count; ++i) tiles[i] = make_uint2(tiles_xy[2 * i], tiles_xy[2 * i + 1]);
    const uint32_t kf = key_frame_host(seed, e.d.frame);
    const uint32_t full = e.quad_words;
    if (lds_depth == 0 || lds_depth > full) lds_depth = full;
    trace_blocks = std::max(1u, std::min(trace_blocks, n_chunks));
    std::vector<uint32_t> overflow((size_t)(full + 64u) * trace_blocks * TR_BLOCK, 0u);
    const size_t fb_lds = (size_t)e.depth * TR_BLOCK * 4;
    const size_t dyn_lds = (size_t)lds_depth * TR_BLOCK * 4;
    const int feat = feature_set(e);
    // the material sort of the shading stage (default of the library for the compacted schedule; trace == 2 is the slot form without queues)
    if (trace != 0) return -6;
    const bool sorted = !(feat & FEAT_TEX);
    std::vector<uint32_t> kind_queues((size_t)WF_MAT_KINDS * q_cap, 0u);
    uint32_t kinds_present = 0;
    for (const DevMaterial& dm : e.mats) kinds_present |= 1u << dm.mat_kind;
    const uint64_t max_rounds = (uint64_t)((n_items + n_chunks - 1) / n_chunks) * (((uint64_t)spp + 3) / 4 * ((WF_FOLD_C ? 2u : 1u) * e.d.max_depth + 3) + 4) + 32;
    int rc = 0;
    uint64_t rounds = 0;
    // ray binning before the traversal stages, as launch_wavefront sets it up (TRAYHIP_WF_BIN: bit 0 = stage A, bit 1 = stage B)
    bool fused = sorted && WF_FOLD_C && WF_FUSED_DEFAULT != 0;   // launch_wavefront's choice of the shading form (TRAYHIP_WF_FUSED)
    if (const char* fe = getenv("TRAYHIP_WF_FUSED")) fused = sorted && WF_FOLD_C && atoi(fe) != 0;
    uint32_t bin_stages = WF_BIN_DEFAULT;
    if (const char* be = getenv("TRAYHIP_WF_BIN")) bin_stages = (uint32_t)std::max(0, atoi(be)) & 3u;
    std::vector<uint32_t> bin_ctl((size_t)2u * 2u * WF_SEGS * WF_BINS, 0u);
    const WfBinGrid bin_grid = f->n_top_nodes ? wf_bin_grid(f->top_nodes[0].bmin, f->top_nodes[0].bmax) : WfBinGrid{};
    const uint32_t bin_blocks = (pool.seg_cap + WF_BIN_EPB - 1u) / WF_BIN_EPB * WF_SEGS;
#define EMU_K(...) do { if (rc == 0) rc = launch_simt(__VA_ARGS__); } while (0)
#define EMU_ROUND(A, F)                                                                                                                     \
    do {                                                                                                                                    \
        EMU_K(n_chunks, TR_BLOCK, [&] { k_wf_advance<A>(e.d, pool, chunks.data(), bins.data(), tiles.data(), n_items, tile_count, 1u, spp, kf, rgbw, \
                                                         counters, counters + 1, stats.data(), qa, qr, qctl, slice_shift); });                          \
        EMU_K(q_blocks, TR_BLOCK, [&] { k_wf_regen<A>(e.d, pool, chunks.data(), tiles.data(), tile_count, 1u, spp, kf, stats.data(), qr, qa, qctl, slice_shift); }); \
        if (WF_FOLD_C && (bin_stages & 1u)) { EMU_BIN(0, qa, qc, bin_ctl.data()); EMU_TRACE_STAGE(0, A, qc, qb); }   /* wf_round: the binned copy lies in the idle queue's buffer */ \
        else EMU_TRACE_STAGE(0, A, qa, qb);                                                                                                   \
        std::memset(qctl, 0, WF_QCTL_WORDS * sizeof(uint32_t));   /* wf_round: the control words are cleared between trace A and k_wf_begin */ \
        if (fused) {   /* wf_round: k_wf_sort + k_wf_shade_kind, then the occlusion stage */                                                  \
            EMU_K(n_chunks, TR_BLOCK, [&] { k_wf_sort<0>(e.d, pool, n_active, qctl, kind_queues.data()); });                                 \
            EMU_SHADE_KIND(A, TRAY_MAT_MATTE); EMU_SHADE_KIND(A, TRAY_MAT_PLASTIC); EMU_SHADE_KIND(A, TRAY_MAT_METAL); EMU_SHADE_KIND(A, TRAY_MAT_GLASS); \
            EMU_SHADE_KIND(A, TRAY_MAT_ROUGH_GLASS); EMU_SHADE_KIND(A, TRAY_MAT_SPECULAR_METAL); EMU_SHADE_KIND(A, TRAY_MAT_MERL);          \
            EMU_TRACE_STAGE(1, A, qb, qc);                                                                                                  \
            break;                                                                                                                          \
        }                                                                                                                                   \
        EMU_K(n_chunks, TR_BLOCK, [&] { k_wf_begin<A>(e.d, pool, n_active, stats.data(), qb, qctl, sorted ? kind_queues.data() : nullptr); }); \
        if (WF_FOLD_C && (bin_stages & 2u)) { EMU_BIN(1, qb, qa, bin_ctl.data() + 2u * WF_SEGS * WF_BINS); EMU_TRACE_STAGE(1, A, qa, qc); }     \
        else EMU_TRACE_STAGE(1, A, qb, qc);                                                                                                   \
        if (sorted) {   /* wf_round of kernels.hip: one kind-pure shading launch per material kind of the scene */                        \
            EMU_QUERY_KIND(A, TRAY_MAT_MATTE); EMU_QUERY_KIND(A, TRAY_MAT_PLASTIC); EMU_QUERY_KIND(A, TRAY_MAT_METAL); EMU_QUERY_KIND(A, TRAY_MAT_GLASS); \
            EMU_QUERY_KIND(A, TRAY_MAT_ROUGH_GLASS); EMU_QUERY_KIND(A, TRAY_MAT_SPECULAR_METAL); EMU_QUERY_KIND(A, TRAY_MAT_MERL);          \
        } else EMU_K(n_chunks, TR_BLOCK, [&] { k_wf_query<A, FEAT_ALL | FEAT_TEX>(e.d, pool, n_active, WF_FOLD_C ? nullptr : qc, qctl, stats.data(), qa); });  \
        if (!WF_FOLD_C) EMU_TRACE_STAGE(2, A, qc, qb);                                                                                                        \
    } while (0)
#define EMU_SHADE_KIND(A, K) do { if (kinds_present & (1u << K)) EMU_K(q_blocks, TR_BLOCK, [&] { k_wf_shade_kind<A, K>(e.d, pool, kind_queues.data(), qctl, stats.data(), qa, qb); }); } while (0)
#define EMU_QUERY_KIND(A, K) do { if (kinds_present & (1u << K)) EMU_K(q_blocks, TR_BLOCK, [&] { k_wf_query_kind<A, K>(e.d, pool, kind_queues.data(), WF_FOLD_C ? nullptr : qc, qctl, stats.data(), qa); }); } while (0)
#define EMU_TRACE_STAGE(S, A, Q, FB) /* FB: the queue buffer that is idle during stage S takes the deferred rays' records (wf_round) */                                                                                                          \
    do {                                                                                                                                    \
        const uint32_t fu_ = (S == 1 && fused) ? 1u : 0u;                                                                                   \
        EMU_K(trace_blocks, TR_BLOCK, [&] { k_wf_trace_dyn<S, A>(e.d, pool, Q, qctl, stats.data(), lds_depth, overflow.data(), FB, fu_); }, dyn_lds); \
        EMU_K(1u, TR_BLOCK, [&] { k_wf_trace_fallback<S, A>(e.d, pool, qctl, FB, fu_); }, fb_lds);                                    \
        g_wf_deferred += qctl[WF_FB_WORD + S];                                                                                              \
    } while (0)
#define EMU_BIN(S, Q, OUT, CTL) /* ray binning before stage S (wavefront.h: k_wf_bin_hist / k_wf_bin_scatter) */                          \
    do {                                                                                                                                    \
        EMU_K(bin_blocks, TR_BLOCK, [&] { k_wf_bin_hist<S>(pool, Q, qctl, CTL, bin_grid); });                                               \
        EMU_K(bin_blocks, TR_BLOCK, [&] { k_wf_bin_scatter<S>(pool, Q, OUT, qctl, CTL, bin_grid); });                                       \
    } while (0)
#define EMU_ROUND_F(A)                                                                                                                      \
    do {                                                                                                                                    \
        if (feat == FEAT_NONE) EMU_ROUND(A, FEAT_NONE); else if (feat == FEAT_MERL) EMU_ROUND(A, FEAT_MERL);                                \
        else if (feat == FEAT_SPEC) EMU_ROUND(A, FEAT_SPEC); else if (feat == (FEAT_MERL | FEAT_SPEC)) EMU_ROUND(A, FEAT_MERL | FEAT_SPEC);  \
        else if (feat == (FEAT_ALL | FEAT_TEX)) EMU_ROUND(A, FEAT_ALL | FEAT_TEX); else EMU_ROUND(A, FEAT_ALL);                                                                                                        \
    } while (0)
    std::memset(qctl, 0, WF_QCTL_WORDS * sizeof(uint32_t));
    while (rc == 0 && counters[1] < n_items) {
        std::fill(bin_ctl.begin(), bin_ctl.end(), 0u);
        if (moving) EMU_ROUND_F(1); else EMU_ROUND_F(0);
        if (++rounds > max_rounds) rc = -5;   // "wavefront schedule did not terminate"
    }
#undef EMU_ROUND_F
#undef EMU_BIN
#undef EMU_TRACE_STAGE
#undef EMU_QUERY_KIND
#undef EMU_SHADE_KIND
#undef EMU_ROUND
#undef EMU_K
    if (stats_out) {
        DevStats st{};
        for (const DevStats& a : stats) { st.samples += a.samples; st.vertices += a.vertices; st.rays += a.rays; }
        stats_out[0] = st.samples; stats_out[1] = st.vertices; stats_out[2] = st.rays; stats_out[3] = rounds;
    }
    return rc;
}

uint32_t emu_wf_bin_cells(void) { return 1u << WF_BIN_CELL_BITS; }
// The two binning kernels alone (tests/test_device_emulation.py): a queue of n_chunks x 256 slots whose segment `s` holds counts[s] ray records
// (8 words each, at queue[(s * seg_cap + k) * 8]); `sorted` receives every segment counting-sorted by wf_bin_key over the box (bmin, bmax), `keys`
// the key of every sorted entry. Returns the segments' capacity.
uint32_t emu_wf_bin(uint32_t n_chunks, const uint32_t* counts, const uint32_t* queue, uint32_t* sorted, uint32_t* keys, const float* bmin, const float* bmax, int stage) {
    WfPool pool{nullptr, n_chunks * TR_BLOCK, wf_seg_cap(n_chunks)};
    if (!queue) return pool.seg_cap;
    std::vector<uint32_t> qctl(WF_QCTL_WORDS, 0u), ctl((size_t)2u * WF_SEGS * WF_BINS, 0u);
    for (uint32_t s = 0; s < WF_SEGS; ++s) qctl[s * WF_SEG_STRIDE + (uint32_t)stage] = counts[s];
    const WfBinGrid g = wf_bin_grid(bmin, bmax);
    const uint32_t blocks = (pool.seg_cap + WF_BIN_EPB - 1u) / WF_BIN_EPB * WF_SEGS;
    int rc = 0;
    if (stage == 0) {
        rc = launch_simt(blocks, TR_BLOCK, [&] { k_wf_bin_hist<0>(pool, queue, qctl.data(), ctl.data(), g); });
        if (rc == 0) rc = launch_simt(blocks, TR_BLOCK, [&] { k_wf_bin_scatter<0>(pool, queue, sorted, qctl.data(), ctl.data(), g); });
    } else {
        rc = launch_simt(blocks, TR_BLOCK, [&] { k_wf_bin_hist<1>(pool, queue, qctl.data(), ctl.data(), g); });
        if (rc == 0) rc = launch_simt(blocks, TR_BLOCK, [&] { k_wf_bin_scatter<1>(pool, queue, sorted, qctl.data(), ctl.data(), g); });
    }
    if (rc != 0) return 0u;
    for (uint32_t s = 0; s < WF_SEGS; ++s)
        for (uint32_t k = 0; k < counts[s]; ++k) keys[(size_t)s * pool.seg_cap + k] = wf_bin_entry_key(g, sorted, (size_t)s * pool.seg_cap + k);
    return pool.seg_cap;
}

}  // extern "C"

// The device order of the trees and the wavefront traversal's instance records, as tray_scene_create uploads them (host/gates.hpp), for
// tests/test_device_tree_order.py. Two calls: with null outputs the counts come back (top nodes, mesh nodes, meshes, records).
extern "C" int emu_device_trees(const TrayFlatScene* f, uint32_t* counts, TrayBvhNode* top, TrayBvhNode* mesh, TrayMesh* meshes, void* wf_insts, int* narrow) {
    tray::PairedTrees p;
    if (!tray::pair_trees(f, p)) return 1;
    std::vector<tray::WfInst> recs;
    tray::QuadTrees q;
    tray::quad_trees(f, q);
    tray::wf_inst_records(f, q.mesh_first, recs);
    counts[0] = (uint32_t)p.top.size(); counts[1] = (uint32_t)p.mesh.size(); counts[2] = (uint32_t)p.meshes.size(); counts[3] = (uint32_t)recs.size();
    if (top) std::memcpy(top, p.top.data(), p.top.size() * sizeof(TrayBvhNode));
    if (mesh) std::memcpy(mesh, p.mesh.data(), p.mesh.size() * sizeof(TrayBvhNode));
    if (meshes) std::memcpy(meshes, p.meshes.data(), p.meshes.size() * sizeof(TrayMesh));
    if (wf_insts) std::memcpy(wf_insts, recs.data(), recs.size() * sizeof(tray::WfInst));
    if (narrow) *narrow = p.narrow ? 1 : 0;
    return 0;
}
// AnimatedTransform::transform(time) of a spline stack as the DEVICE evaluates it (dev_anim.h: eval_xform_stack): rows 0..2 of mat, rows 0..2 of inv
extern "C" int emu_stack_transform(const TrayFlatScene* f, uint32_t xf_first, uint32_t xf_count, uint32_t n, const float* times, float* out) {
    if (!f || (uint64_t)xf_first + xf_count > f->n_xf_levels) return -1;
    for (uint32_t k = 0; k < n; ++k) eval_xform_stack(f->xf_levels, f->keyframes, f->knots, xf_first, xf_count, times[k], out + (size_t)TR_XF_WORDS * k);
    return 0;
}
extern "C" unsigned emu_retraced(void) { const unsigned r = g_retraced; g_retraced = 0; return r; }
extern "C" unsigned emu_wf_deferred(void) { const unsigned r = g_wf_deferred; g_wf_deferred = 0; return r; }
// The wavefront traversal's quad records (host/gates.hpp: QuadTrees), for tests/test_device_tree_order.py. counts: top records, mesh records,
// meshes, top_pend, mesh_pend; null outputs = counts only. bfs_levels < 0: the library's default.
extern "C" int emu_quad_trees(const TrayFlatScene* f, int bfs_levels, uint32_t* counts, void* top, void* mesh, uint32_t* mesh_first, int* narrow) {
    tray::PairedTrees p;
    if (!tray::pair_trees(f, p)) return 1;
    tray::QuadTrees q;
    if (bfs_levels < 0) tray::quad_trees(f, q); else tray::quad_trees(f, q, false, (uint32_t)bfs_levels);
    counts[0] = (uint32_t)q.top.size(); counts[1] = (uint32_t)q.mesh.size(); counts[2] = (uint32_t)q.mesh_first.size(); counts[3] = q.top_pend; counts[4] = q.mesh_pend;
    if (top) std::memcpy(top, q.top.data(), q.top.size() * sizeof(tray::QuadNode));
    if (mesh) std::memcpy(mesh, q.mesh.data(), q.mesh.size() * sizeof(tray::QuadNode));
    if (mesh_first) std::memcpy(mesh_first, q.mesh_first.data(), q.mesh_first.size() * sizeof(uint32_t));
    if (narrow) *narrow = q.narrow ? 1 : 0;
    return 0;
}
