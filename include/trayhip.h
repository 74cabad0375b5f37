/*
 * trayhip.h — C ABI of libtrayhip.so, the MI355X (gfx950) path-tracing core that
 * replaces tray_rust's tile worker.
 *
 * Drop-in seam (reference, /root/reference):
 *   trait Exec::render(&mut self, &mut Scene, &mut RenderTarget, &Config)   src/exec/mod.rs:41-49
 *   MultiThreaded::render_parallel + thread_work                            src/exec/multithreaded.rs:30-114
 *   Scene::load_file                                                        src/scene.rs:101-145
 *   RenderTarget::{write,get_render,get_renderf32}                          src/film/render_target.rs:77,185,243
 *   film::Image::add_pixels (merge of per-worker RGBW)                      src/film/image.rs:21-34
 *   BlockQueue::new (Morton tile list + select_blocks)                      src/sampler/block_queue.rs:28-48
 *
 * Everything is plain C: PODs, pointers and sizes. No torch / C++ types cross this line.
 * All entry points return 0 on success and a negative TRAY_E_* code on failure; the message is
 * available from tray_last_error() (thread local). Nothing aborts across the ABI: the reference's
 * panics (scene.rs:104-136, block_queue.rs:29-31, multithreaded.rs:39) become checked errors.
 *
 * The same TrayFlatScene POD is what the CPU oracle (oracle/, test infrastructure only) consumes,
 * so oracle and HIP path are fed bit-identical scene data.
 */
#ifndef TRAYHIP_H
#define TRAYHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRAY_ABI_VERSION 4

enum {
    TRAY_OK = 0,
    TRAY_E_INVALID = -1,   /* bad argument / precondition (reference would panic) */
    TRAY_E_IO = -2,        /* file could not be opened / read */
    TRAY_E_PARSE = -3,     /* JSON / OBJ / MERL parse error */
    TRAY_E_UNSUPPORTED = -4, /* feature outside the hot-path scope */
    TRAY_E_DEVICE = -5,    /* HIP runtime error */
    TRAY_E_NOMEM = -6
};

/* ---------------------------------------------------------------- flat scene (SoA/AoS PODs) */

/* Flattened BVH2 node, 32 B: reference FlatNode (src/geometry/bvh.rs:278-295).
 * count > 0  : leaf, prims ordered[offset .. offset+count)
 * count == 0 : interior, first child = this+1, second child = offset, split axis = axis */
typedef struct TrayBvhNode {
    float bmin[3];
    float bmax[3];
    uint32_t offset;
    uint16_t count;
    uint8_t axis;
    uint8_t pad;
} TrayBvhNode;

/* Hot triangle record (48 B, 3 x float4) in mesh-BVH leaf order: positions only.
 * Reference Triangle{a,b,c,positions} (src/geometry/mesh.rs:88-111), pre-gathered. */
typedef struct TrayTriVerts {
    float pa[3]; uint32_t tri_id;  /* index of this triangle in the OBJ order (for debugging) */
    float pb[3]; uint32_t pad0;
    float pc[3]; uint32_t pad1;
} TrayTriVerts;

/* Cold triangle record (64 B): shading normals + texcoords, read once for the final hit. */
typedef struct TrayTriAttrs {
    float na[3], nb[3], nc[3];
    float ta[2], tb[2], tc[2];
    float pad;
} TrayTriAttrs;

typedef struct TrayMesh {
    uint32_t node_offset;  /* into TrayFlatScene.mesh_nodes */
    uint32_t node_count;
    uint32_t tri_offset;   /* into tri_verts / tri_attrs (leaf order) */
    uint32_t tri_count;
} TrayMesh;

/* AnimatedMesh (src/geometry/animated_mesh.rs:93-143): a mesh whose vertex positions, normals and texcoords are interpolated linearly
 * between keyframes at ray.time (AnimatedMeshData::position / normal / texcoord, :72-107), behind ONE BVH<AnimatedTriangle> built over the
 * triangles' bounds at times[0] and times[1] (AnimatedMesh::new, :124-126 -- Boundable::update_deformation, :140-142, has no caller in the
 * reference, so that tree serves every frame: quirk Q13). meshes[m] describes the tree and ONE keyframe's triangles (tri_count); keyframe k
 * of the mesh lies at tri_verts / tri_attrs [tri_offset + k * tri_count ...], all in the leaf order of that tree; mesh_keys[m] names the
 * keyframe times. n_keys == 1 for a plain Mesh. */
typedef struct TrayMeshKeys {
    uint32_t n_keys;       /* >= 2 for an animated mesh */
    uint32_t time_first;   /* into TrayFlatScene.key_times (ascending per mesh) */
} TrayMeshKeys;

enum { TRAY_GEOM_SPHERE = 0, TRAY_GEOM_DISK = 1, TRAY_GEOM_RECT = 2, TRAY_GEOM_MESH = 3, TRAY_GEOM_NONE = 4, TRAY_GEOM_ANIMATED_MESH = 5 };
enum { TRAY_INST_RECEIVER = 0, TRAY_INST_AREA_EMITTER = 1, TRAY_INST_POINT_EMITTER = 2 };

/* One Instance (src/geometry/instance.rs:72-105) with its transform evaluated for the frame.
 * mat/inv are the full row-major 4x4 of Transform{mat,inv} (src/linalg/transform.rs:11-14) built
 * in the reference's order: per spline level translate*rot*scale (keyframe.rs:60-63), stacked
 * object-first (animated_transform.rs:40-56). Row 3 is kept because Transform*Point divides by w
 * only when |w-1| < eps (transform.rs:211-215). */
typedef struct TrayInstance {
    uint32_t kind;        /* TRAY_INST_* */
    uint32_t geom_type;   /* TRAY_GEOM_* */
    uint32_t mesh_id;     /* valid for TRAY_GEOM_MESH */
    uint32_t material_id; /* 0xffffffff for point emitters */
    float geom_params[4]; /* sphere: radius | disk: radius, inner_radius | rect: width, height */
    float emission[4];    /* AnimatedColor::color(shutter_open) (rgb, a); exact for every ray when emis_count <= 1 */
    float mat[16];        /* transform(shutter_open); exact for every ray when animated == 0 */
    float inv[16];
    uint32_t light_index; /* index in lights[] or 0xffffffff */
    uint32_t xf_first;    /* first TrayXformLevel of this instance's spline stack */
    uint32_t xf_count;    /* number of levels (object first, then group parents) */
    uint32_t animated;    /* 1 => some level varies while the shutter is open: evaluate the stack at ray.time
                           * (receiver.rs:30, emitter.rs:122,161,169,190) */
    uint32_t emis_first;  /* AnimatedColor keyframes in color_keys[] (film/animated_color.rs:45-49), sorted by time */
    uint32_t emis_count;
    uint32_t moving_slot; /* index among the animated instances of this frame (0xffffffff if not animated) */
    uint32_t pad;
} TrayInstance;

/* TRS keyframe (src/linalg/keyframe.rs:13-17) */
typedef struct TrayKeyframe {
    float translation[3];
    float rotation[4];    /* quaternion v.xyz, w */
    float scaling[3];
} TrayKeyframe;

/* One B-spline level of an AnimatedTransform (src/linalg/animated_transform.rs:15-19).
 * BSpline<Keyframe> of the bspline crate (0.2.2, not vendored by the reference): clamped de Boor evaluation over the
 * sorted knot vector, domain (knots[degree], knots[n_knots-1-degree]); a level with ONE control point is a constant
 * (animated_transform.rs:47-48) whose Transform is stored in mat/inv. */
typedef struct TrayXformLevel {
    uint32_t kf_first, kf_count;     /* control points in keyframes[] */
    uint32_t knot_first, knot_count; /* knots in knots[] (sorted ascending) */
    uint32_t degree;
    uint32_t is_const;               /* 1 => the level has the same value for every ray of this frame: one control point, or
                                      * the open shutter lies outside the knot domain, where transform() clamps the time
                                      * (animated_transform.rs:49-50). mat/inv hold that value. */
    uint32_t pad[2];
    float mat[16];                   /* Keyframe::transform() of the constant level (keyframe.rs:60-63) */
    float inv[16];
} TrayXformLevel;

/* ColorKeyframe (src/film/animated_color.rs:10-14) */
typedef struct TrayColorKey {
    float color[4];
    float time;
    float pad[3];
} TrayColorKey;

enum {
    TRAY_MAT_MATTE = 0, TRAY_MAT_PLASTIC = 1, TRAY_MAT_METAL = 2, TRAY_MAT_GLASS = 3,
    TRAY_MAT_ROUGH_GLASS = 4, TRAY_MAT_SPECULAR_METAL = 5, TRAY_MAT_MERL = 6
};

/* Image textures (src/texture/image.rs:9-47, animated_image.rs:7-58; loader scene.rs:317-394). A texture is one frame (type
 * "image") or >= 2 keyed frames ("animated_image", "movie": lerp between the two frames around ray.time, the first / last frame
 * outside the keys). Frames are RGBA8 as image::DynamicImage::get_pixel presents them (row 0 on top; grey -> l,l,l,255). */
typedef struct TrayTexture {
    uint32_t first_frame, n_frames;   /* in tex_frames[] */
} TrayTexture;
typedef struct TrayTexFrame {
    float time;                        /* keyframe time (0 for a plain image) */
    uint32_t width, height;
    uint32_t pad;
    uint64_t offset;                   /* byte offset of the frame's width*height*4 bytes in tex_data */
} TrayTexFrame;
#define TRAY_NO_TEXTURE 0xffffffffu

/* Closed lowering of the reference's Material trait objects (src/material/ *.rs). Every parameter is a Texture in the
 * reference (texture/mod.rs:15-20): here a constant (c0 / c1 / f0 / f1; scalars read from constant colours are their
 * luminance, texture/mod.rs:70-72) or, when the matching tex_* field is not TRAY_NO_TEXTURE, an image texture sampled at the
 * hit's (u, v, time) -- sample_color for colour parameters, sample_f32 (red channel) for scalar ones.
 *   MATTE          c0 = diffuse,            f0 = roughness            (matte.rs:52-65)
 *   PLASTIC        c0 = diffuse, c1 = gloss, f0 = roughness           (plastic.rs:59-88)
 *   METAL          c0 = eta, c1 = k,        f0 = roughness            (metal.rs:56-67)
 *   GLASS          c0 = reflect, c1 = transmit, f0 = eta              (glass.rs:51-78)
 *   ROUGH_GLASS    c0 = reflect, c1 = transmit, f0 = eta, f1 = roughness (rough_glass.rs:57-85)
 *   SPECULAR_METAL c0 = eta, c1 = k                                   (specular_metal.rs:49-58)
 *   MERL           table = index into merl tables                     (material/merl.rs:88-92) */
typedef struct TrayMaterial {
    uint32_t kind;
    uint32_t table;
    float f0, f1;
    float c0[4];
    float c1[4];
    uint32_t tex_c0, tex_c1, tex_f0, tex_f1;   /* texture ids or TRAY_NO_TEXTURE */
    uint32_t microfacet;   /* TRAY_MF_*: the MicrofacetDistribution of plastic / metal / rough_glass. The reference's materials
                            * always build Beckmann (plastic.rs:83, metal.rs:63, rough_glass.rs:73); its GGX (bxdf/microfacet/
                            * ggx.rs:20-57) is selected here by the scene-file extension key "microfacet": "ggx" */
    uint32_t pad[3];
} TrayMaterial;
enum { TRAY_MF_BECKMANN = 0, TRAY_MF_GGX = 1 };
/* Integrator (src/integrator): the Path tracer (path.rs) or NormalsDebug (normals_debug.rs:28-33: (bsdf.n + 1) / 2 of the
 * camera ray's hit), or Whitted (whitted.rs:41-68 with Integrator::specular_reflection / specular_transmission, mod.rs:49-97). */
enum { TRAY_INTEGRATOR_PATH = 0, TRAY_INTEGRATOR_NORMALS_DEBUG = 1,
       TRAY_INTEGRATOR_WHITTED = 2 };   /* integrator/whitted.rs: max_depth = the recursion limit (<= 16), min_depth unused */

/* MERL table header: 90*90*180 RGB-interleaved f32, already scaled (material/merl.rs:60-82) */
typedef struct TrayMerlTable {
    uint64_t offset;  /* float offset into merl_data */
    uint32_t n_theta_h, n_theta_d, n_phi_d;
    uint32_t pad;
} TrayMerlTable;

/* Camera for one frame (src/film/camera.rs:64-157).
 * raster_to_cam = (proj_div_inv * raster_screen).mat, a genuinely projective 4x4 (Q5). */
typedef struct TrayCamera {
    float raster_to_cam[16];
    float scaling[3];
    float shutter_open, shutter_close;
    float cam_world[16];   /* cam_world.transform(shutter_open).mat; exact for every ray when animated == 0 */
    uint32_t animated;     /* 1 => cam_world.transform(frame_time) is evaluated per ray (camera.rs:156) */
    uint32_t xf_first, xf_count;
} TrayCamera;

#define TRAY_FILTER_TABLE_SIZE 16

/* Film + reconstruction filter (src/film/render_target.rs:41-75) */
typedef struct TrayFilm {
    uint32_t width, height;
    float filter_w, filter_h, inv_w, inv_h;
    int32_t filter_pixel_w, filter_pixel_h;   /* floor(w/0.5), floor(h/0.5) */
    float table[TRAY_FILTER_TABLE_SIZE * TRAY_FILTER_TABLE_SIZE];
    /* 1-D factors when the filter is a product of per-axis weights (both reference filters are):
     * table[y*16 + x] == table_x[x] * table_y[y] in f32. separable = 0 if no such factors exist. */
    float table_x[TRAY_FILTER_TABLE_SIZE];
    float table_y[TRAY_FILTER_TABLE_SIZE];
    uint32_t separable;
} TrayFilm;

typedef struct TrayFlatScene {
    uint32_t abi_version;
    uint32_t frame;
    TrayFilm film;
    TrayCamera camera;
    uint32_t min_depth, max_depth;       /* Path integrator (src/integrator/path.rs:33-43) */

    uint32_t n_instances;   const TrayInstance* instances;
    uint32_t n_top_nodes;   const TrayBvhNode* top_nodes;       /* BVH<Instance>, leaf <= 4 */
    uint32_t n_top_order;   const uint32_t* top_order;          /* ordered_geom of the top BVH */
    uint32_t n_meshes;      const TrayMesh* meshes;
    uint32_t n_mesh_nodes;  const TrayBvhNode* mesh_nodes;      /* all BVH<Triangle>, leaf <= 16 */
    uint32_t n_tris;        const TrayTriVerts* tri_verts;      /* leaf order, per mesh */
                            const TrayTriAttrs* tri_attrs;
    uint32_t n_materials;   const TrayMaterial* materials;
    uint32_t n_merl;        const TrayMerlTable* merl_tables;
    uint64_t n_merl_floats; const float* merl_data;
    uint32_t n_lights;      const uint32_t* lights;             /* instance ids, scene order */
    uint32_t n_xf_levels;   const TrayXformLevel* xf_levels;
    uint32_t n_keyframes;   const TrayKeyframe* keyframes;
    uint32_t n_knots;       const float* knots;
    uint32_t n_color_keys;  const TrayColorKey* color_keys;
    uint32_t animated;      /* 1 if the camera, any instance transform or any emission varies over the open shutter */
    uint32_t integrator;    /* TRAY_INTEGRATOR_* */
    uint32_t n_textures;    const TrayTexture* textures;
    uint32_t n_tex_frames;  const TrayTexFrame* tex_frames;
    uint64_t n_tex_bytes;   const uint8_t* tex_data;            /* RGBA8 texels of all frames */
    uint32_t n_mesh_keys;   const TrayMeshKeys* mesh_keys;      /* n_meshes entries, or 0 / NULL when no mesh is animated */
    uint32_t n_key_times;   const float* key_times;
} TrayFlatScene;

/* ---------------------------------------------------------------- host side: loader (scene.rs) */

typedef struct TrayHostScene TrayHostScene;

typedef struct TraySceneInfo {
    uint32_t width, height;
    uint32_t spp;                /* film.samples as written in the file */
    uint32_t frames, start_frame, end_frame;
    float scene_time;
    uint32_t n_instances, n_lights, n_meshes, n_tris;
} TraySceneInfo;

/* Scene::load_file (src/scene.rs:101-145). The JSON schema is the reference's. */
int tray_scene_load_file(const char* path, TrayHostScene** out);
/* Same, from an in-memory JSON string; base_dir resolves relative mesh / MERL paths. */
int tray_scene_load_string(const char* json, const char* base_dir, TrayHostScene** out);
int tray_host_scene_info(const TrayHostScene* s, TraySceneInfo* info);
/* Scene::update_frame (scene.rs:152-176) + flattening for frame `frame`: camera shutter, top-level
 * BVH rebuilt over the shutter interval, per-instance matrices. The returned view borrows from `s`
 * and is valid until the next flatten / free on `s`. */
int tray_host_scene_flatten(TrayHostScene* s, uint32_t frame, const TrayFlatScene** out);
void tray_host_scene_free(TrayHostScene* s);

/* BlockQueue::new (src/sampler/block_queue.rs:28-48): 8x8 tiles sorted by Morton code of the
 * tile index. Writes up to `cap` (x,y) tile coordinates into xy (2*u32 each), returns the number
 * of tiles through n_out. select (start,count) applies after sorting; count 0 = all. */
int tray_block_queue(uint32_t width, uint32_t height, uint32_t select_start, uint32_t select_count,
                     uint32_t* xy, uint32_t cap, uint32_t* n_out);

/* LowDiscrepancy::new rounding (src/sampler/ld.rs:22-25) */
uint32_t tray_round_spp(uint32_t spp);

/* RenderTarget::get_render (render_target.rs:185-210): RGBW f32 -> sRGB8 (3 bytes / pixel). */
int tray_resolve_srgb8(const float* rgbw, uint32_t width, uint32_t height, uint8_t* rgb8);

/* ---------------------------------------------------------------- device side: the tile worker */

typedef struct TrayDeviceScene TrayDeviceScene;

/* Bind the calling thread's library state to HIP device `device` (one process per GPU). */
int tray_init(int device);
int tray_device_count(int* n);

/* Deep-copies the flat scene to the current device. */
int tray_scene_create(const TrayFlatScene* flat, TrayDeviceScene** out);
/* Scene::update_frame (src/scene.rs:152-176; the frame loop of src/main.rs:91-106 keeps the Scene and rebuilds the instance
 * transforms and BVH<Instance> per frame): `flat` is the SAME scene flattened at another frame. Instances, BVH<Instance>, camera,
 * spline tables, emission keys and the set of moving instances are uploaded anew; meshes, MERL tables, textures, the tile queue,
 * the wavefront pool / queues and the per-path transform cache stay on the device. Waits for the device to be idle. On an error
 * the handle can only be passed to tray_scene_destroy. */
int tray_scene_update_frame(TrayDeviceScene* s, const TrayFlatScene* flat);
void tray_scene_destroy(TrayDeviceScene* s);

/* thread_work over tiles [tile_start, tile_start+tile_count) of the Morton queue
 * (exec/multithreaded.rs:72-114 with Config.select_blocks, exec/mod.rs:25-27).
 * Adds filtered samples into rgbw_dev: device pointer, width*height*4 f32, the layout of
 * RenderTarget::get_renderf32 (render_target.rs:243-266). Asynchronous on `stream`
 * (a hipStream_t; NULL = default stream). spp must already be a power of two (it is not used when tray_scene_set_sampler chose
 * Uniform or Adaptive). */
/* tile_count == 0 selects the whole queue whatever tile_start is (BlockQueue::new, block_queue.rs:39-41).
 * ONE render may be in flight per TrayDeviceScene: the tile counter, the statistics, the per-path transform cache, the wavefront
 * pool and queues and the timing events belong to the handle. Calls on one handle must be serialised by the caller (the
 * reference blocks inside Exec::render too, multithreaded.rs:54-70); use one handle per stream for concurrent renders.
 * Scenes that traverse BVH<Instance> (more than 16 instances) run the wavefront schedule: it polls for completion (the call
 * returns when the tiles are done), uses one internal stream beside `stream` (forked from and joined back to it with events) and
 * returns TRAY_E_UNSUPPORTED for a mesh of more than 8 388 607 BVH nodes or triangles (its traversal keeps a node as a 32-bit word).
 * Its path pool is sized for the device: up to 32 M paths in flight (8.9 GB + 5 GB of queues and film bins) and, for a scene with instances that
 * move while the shutter is open, 112 B per path and such instance of cached transforms within two fifths of the free memory (TRAYHIP_WF_SLOTS /
 * TRAYHIP_XF_CACHE_BYTES bound both; a failed allocation is TRAY_E_NOMEM). */
int tray_render_tiles_device(TrayDeviceScene* s, uint32_t tile_start, uint32_t tile_count,
                             uint32_t spp, uint64_t seed, float* rgbw_dev, void* stream);

/* Multi-GPU sharding of one frame: shard g of n_shards renders chunks g, g+n_shards, g+2*n_shards...
 * of the Morton queue, chunk_tiles tiles per chunk (round-robin instead of the reference's contiguous
 * per = n/workers split, src/exec/distrib/master.rs:91-93,218-227, for load balance; per-pixel results
 * do not depend on the partition because the RNG is keyed by pixel and sample). The per-shard buffers
 * are merged by addition (film/image.rs:21-50), e.g. ncclReduce(sum). */
int tray_render_shard_device(TrayDeviceScene* s, uint32_t shard, uint32_t n_shards, uint32_t chunk_tiles,
                             uint32_t spp, uint64_t seed, float* rgbw_dev, void* stream);

/* Host-side enumeration of the Morton-queue indices tray_render_shard_device renders for `shard`
 * (same mapping; lets callers and tests reason about the partition without a GPU). */
int tray_shard_tiles(uint32_t n_tiles, uint32_t shard, uint32_t n_shards, uint32_t chunk_tiles,
                     uint32_t* out, uint32_t cap, uint32_t* n_out);

/* Synchronous convenience wrapper: renders into a zeroed device buffer, adds it into rgbw_host
 * (host, width*height*4 f32) — semantics of film::Image::add_pixels. */
int tray_render_tiles(TrayDeviceScene* s, uint32_t tile_start, uint32_t tile_count,
                      uint32_t spp, uint64_t seed, float* rgbw_host);

/* ---- the other Samplers (src/sampler/mod.rs:20-49) -----------------------------------------------------------------
 * thread_work constructs `sampler::LowDiscrepancy::new(queue.block_dim(), spp)` (exec/multithreaded.rs:74); sampler/uniform.rs and
 * sampler/adaptive.rs implement the same trait and are what a maintainer would write there instead. tray_scene_set_sampler selects
 * which one the render calls on this handle stand for:
 *   TRAY_SAMPLER_LOW_DISCREPANCY  LowDiscrepancy::new(dim, spp) -- the default; `spp` of the render call (ld.rs:20-31)
 *   TRAY_SAMPLER_UNIFORM          Uniform::new(dim): one sample at the centre of every pixel, every other number an independent
 *                                 uniform draw (uniform.rs:22-47); the render call's spp is not used
 *   TRAY_SAMPLER_ADAPTIVE         Adaptive::new(dim, min_spp, max_spp): min_spp samples per pixel, then step_size more while the
 *                                 luminance of any of the pixel's samples so far lies more than 50 % off their running average and fewer
 *                                 than max_spp have been taken (adaptive.rs:34-75, 133-143 incl. its (i - 1) / i averaging and the
 *                                 sample-index offsets of adaptive.rs:112-121); min_spp / max_spp are rounded up to powers of two as
 *                                 Adaptive::new does (0 counts as 1); the render call's spp is not used
 * Every sample is still keyed by (seed, frame, pixel, pass, index) (TRAY-CBRNG), so shards and tile ranges add up
 * to the same film. Uniform and Adaptive run one thread per camera sample of a pass (k_sampler_pass) for every scene; the per-pixel
 * sample counts of the last render are reported through TrayKernelTiming.samples (their sum). min_spp / max_spp are ignored for the
 * other two kinds. Returns TRAY_E_INVALID for an unknown kind or max_spp < min_spp after rounding. */
enum { TRAY_SAMPLER_LOW_DISCREPANCY = 0, TRAY_SAMPLER_UNIFORM = 1, TRAY_SAMPLER_ADAPTIVE = 2 };
int tray_scene_set_sampler(TrayDeviceScene* s, uint32_t kind, uint32_t min_spp, uint32_t max_spp);
/* Adaptive::new's step_size (adaptive.rs:48): ((max_spp - min_spp) / 5).next_power_of_two(), of the ROUNDED min / max */
uint32_t tray_adaptive_step(uint32_t min_spp, uint32_t max_spp);

/* Timing of the most recent tray_render_tiles_device launch sequence on this scene, measured with
 * HIP events on the launch stream. Blocks until those kernels finished. */
typedef struct TrayKernelTiming {
    float render_ms;        /* k_path_tiles */
    uint32_t launches;
    uint64_t samples;       /* camera samples traced */
    uint64_t vertices;      /* path vertices shaded (iterations of path.rs:69) */
    uint64_t rays;          /* Scene::intersect calls of the reference's algorithm. A few of them are proven irrelevant before they are traced
                               -- a BSDF-sampled light ray that misses the light's own primitive, an occlusion ray whose BSDF value is
                               black -- and only counted */
    uint64_t retraced;      /* rays of the flat instance loop whose closest candidates tied (or sat inside one another's bounding-box
                             * window) and that were therefore traced again with the reference's BVH<Instance> traversal
                             * (geometry/bvh.rs:81-130), which decides by its visiting order; about 1 in 1e7 on cornell_box */
} TrayKernelTiming;
int tray_last_timing(TrayDeviceScene* s, TrayKernelTiming* t);

/* Footprint and shape of the wavefront schedule (scenes that traverse BVH<Instance>: more than 16 instances) -- the reference has no
 * counterpart: its workers keep one path per thread on the stack (exec/multithreaded.rs:72-114); here up to 32 M paths are in flight in a
 * pool in HBM. pool_slots: paths in flight (0 = the library's rule: 32 M, never more than 4096 per tile, the pool with its queues within a
 * third of the device's free memory, a moving scene's per-path transform cache within two fifths); views: independent halves / thirds /
 * quarters of the pool on streams of their own (0 = rule: 1 from 24 M slots, else 2); slices: work items a tile's samples are cut into
 * (0 = rule; a power of two <= 16). A host that keeps several device scenes on one GPU sets pool_slots so that they fit beside each other
 * (0.47 KB per slot + 112 B per slot and instance that moves within a frame). Takes effect at the next render call (the buffers are freed
 * and allocated anew if the pool's size changes); for a moving scene the pool cannot grow beyond the transform cache allocated at
 * tray_scene_create / tray_scene_update_frame, which follow the setting. If the allocation fails the library halves the pool down to
 * 16 384 slots before it returns TRAY_E_NOMEM, and the handle stays usable. The environment switches TRAYHIP_WF_SLOTS / _PIPES / _SLICES
 * (measurement only) override these. TRAY_E_INVALID for views > 4 or slices not a power of two <= 16. */
int tray_scene_set_wavefront(TrayDeviceScene* s, uint32_t pool_slots, uint32_t views, uint32_t slices);
/* The schedule the last render call on this scene ran with (what tools/pmc_workloads.py records beside its counters). */
typedef struct TrayScheduleInfo {
    uint32_t wavefront;           /* 1: the scene takes the wavefront schedule, 0: the tile kernel */
    uint32_t launched_wavefront;  /* 1: the last render call ran the wavefront schedule (0 also for the Uniform / Adaptive sampler passes) */
    uint32_t pool_slots, chunks;  /* paths in flight (0 until the first wavefront launch allocated the pool), chunks of 256 */
    uint32_t views, slices;       /* of the last wavefront launch */
    uint32_t n_moving;            /* instances that move within the frame (columns of the per-path transform cache) */
    uint32_t tile_workgroups;     /* persistent workgroups of the tile kernel */
    uint64_t pool_bytes, schedule_bytes, xf_cache_bytes;   /* pool alone; pool + queues + bins; per-path transform cache */
    uint32_t transform_table;     /* 1: the last launch read the frame's transform table (tray_scene_set_transform_table) */
    uint32_t binned_stages;       /* wavefront schedule: traversal stages whose rays are sorted by (origin cell, direction octant) before they are
                                   * traced -- bit 0: camera / continuation rays, bit 1: occlusion rays (round 6; this word was padding before) */
    uint64_t xf_table_bytes;                               /* the transform table's buffer, if there is one: sized once for every instance that any frame of the sequence can move */
} TrayScheduleInfo;
int tray_last_schedule(TrayDeviceScene* s, TrayScheduleInfo* out);

/* Moving scenes: where a path's AnimatedTransform::transform(ray.time) comes from (linalg/animated_transform.rs:40-56; the reference
 * rebuilds it at every instance visit, geometry/receiver.rs:30). A camera sample's shutter time is one of 2^24 values (sampler/ld.rs:100-104),
 * every ray of the path inherits it (path.rs:110), so the transform is a function of a 24-bit index. mode 1: build, per frame and on the
 * first launch, the TABLE of every moving instance's (and a moving camera's) transform at all 2^24 times -- 2.1 GB each (128-byte records), ~1.5 ms each to
 * build, the same evaluation at the same times: the same bits -- and read it; mode 0: evaluate per camera sample into a per-path cache
 * (128 B per path and instance); mode -1 (default): the table for launches of >= 3e7 camera samples (a 1080p frame at 16 spp: each index is needed about twice or more),
 * for every later launch of the frame once it exists, and for the frames that follow it through tray_scene_update_frame (the buffer -- sized once for every
 * instance whose transform has several keyframes -- and the wavefront pool are handed on). If the table cannot be allocated the per-path cache serves, and vice versa.
 * TRAYHIP_XF_TABLE=0|1 (measurement) overrides. tray_debug_transform_table compares n pseudo-random records of the frame's table with a fresh
 * evaluation, bit for bit, and reports how many differ (test hook; TRAY_E_INVALID if the frame has no table yet). */
int tray_scene_set_transform_table(TrayDeviceScene* s, int mode);
int tray_debug_transform_table(TrayDeviceScene* s, uint32_t n, uint32_t* n_differ);

/* ---- one frame on several GPUs of this process ------------------------------------------------------
 * The reference's distributed mode hands every worker a slice of the block queue and sums the returned RGBW blocks on the
 * master (src/exec/distrib/master.rs:91-93,124-163; film::Image::add_blocks, src/film/image.rs:36-50). Here the workers are
 * the GPUs of one node: tray_multi_create deep-copies the scene to each listed device and creates one RCCL communicator per
 * device (ncclCommInitAll); the calls leave the calling thread's current HIP device as they found it; tray_render_frame_multi renders shard d of n_dev on device d (tray_render_shard_device, 16-tile
 * chunks round-robin, one host thread and one stream per device), sums the per-device films onto the first device with ONE
 * ncclReduce(sum) over xGMI and adds the result into rgbw_host (width*height*4 f32, get_renderf32 layout). A Rust
 * exec::Hip that owns a whole node calls these three instead of spawning worker processes. RCCL is loaded with dlopen
 * (librccl.so) when the first TrayMultiScene is created: a build or a box without it still serves the single-GPU calls. */
typedef struct TrayMultiScene TrayMultiScene;
int tray_multi_create(const TrayFlatScene* f, int n_dev, const int* dev_ids, TrayMultiScene** out);
int tray_render_frame_multi(TrayMultiScene* m, uint32_t spp, uint64_t seed, float* rgbw_host);
/* tray_scene_set_sampler on every device of m */
int tray_multi_set_sampler(TrayMultiScene* m, uint32_t kind, uint32_t min_spp, uint32_t max_spp);
/* tray_scene_set_wavefront on every device of m */
int tray_multi_set_wavefront(TrayMultiScene* m, uint32_t pool_slots, uint32_t views, uint32_t slices);
/* tray_scene_set_transform_table on every device of m */
int tray_multi_set_transform_table(TrayMultiScene* m, int mode);
/* tray_scene_update_frame on every device of m; the communicators, films and streams are kept (scene.rs:152-176 per worker) */
int tray_multi_update_frame(TrayMultiScene* m, const TrayFlatScene* f);
/* per-device timings of the last tray_render_frame_multi (n_dev entries) and the duration of the reduce (ms) */
int tray_multi_timing(TrayMultiScene* m, TrayKernelTiming* per_device, float* reduce_ms);
void tray_multi_destroy(TrayMultiScene* m);

/* ---- parity / debug entry points (same device code as the renderer, one thread per item) ---- */

typedef struct TrayRay { float o[3]; float d[3]; float min_t, max_t, time; } TrayRay;   /* ray.rs:9-22 */
typedef struct TrayHit {
    float t;
    uint32_t inst;      /* instance id or 0xffffffff on miss */
    uint32_t prim;      /* triangle slot (leaf order) for meshes, else 0 */
    float p[3], n[3], ng[3];
    float u, v;
    float dp_du[3], dp_dv[3];
} TrayHit;
/* Scene::intersect (scene.rs:148-150) for n host rays -> n host hits. */
int tray_debug_intersect(TrayDeviceScene* s, uint32_t n, const TrayRay* rays, TrayHit* hits);

/* Radiance of individual camera samples: for item i, pixel (px[i], py[i]), sample index si[i]:
 * out[i*8 + 0..2] = clamped rgb (multithreaded.rs:98-99), [3] = sample x, [4] = sample y,
 * [5] = number of path vertices, [6] = number of rays, [7] = 0. */
int tray_debug_sample_radiance(TrayDeviceScene* s, uint32_t n, const uint32_t* px, const uint32_t* py,
                               const uint32_t* si, uint32_t spp, uint64_t seed, float* out);

/* BSDF::eval / pdf / sample (src/bxdf/bsdf.rs:66-125) of material `material_id` on a canonical
 * frame (n = +z, dp_du = +x). For item i: wo = dirs[i*6..+3], wi = dirs[i*6+3..+6], u = u3[i*3..+3]
 * (two_d.0, two_d.1, one_d). out[i*12]: eval rgb(3), pdf(1), sample f rgb(3), sample wi(3),
 * sample pdf(1), sampled type bits(1, as float). flags: 0 = BxDFType::all(), 1 = non_specular(). */
int tray_debug_bsdf(TrayDeviceScene* s, uint32_t material_id, uint32_t flags, uint32_t n,
                    const float* dirs, const float* u3, float* out);

const char* tray_last_error(void);
const char* tray_version(void);

/* sizeof() of the structs above as this library was compiled, for bindings that restate the layouts (ctypes, repr(C)):
 * name is the struct name ("TrayInstance", ...); 0 for an unknown name. */
uint32_t tray_abi_sizeof(const char* name);

#ifdef __cplusplus
}
#endif
#endif /* TRAYHIP_H */
