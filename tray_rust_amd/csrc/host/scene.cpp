// Host-side scene loader and per-frame flattening: the C++ stand-in for src/scene.rs (no Rust
// toolchain exists in the build image). It keeps the reference's JSON schema, turns
// the reference's panics into error codes, and lowers the loaded scene to the TrayFlatScene POD that
// both the HIP tile worker and the CPU oracle consume.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/trayhip.h"
#include "bvh.hpp"
#include "validate.hpp"
#include "json.hpp"
#include "linalg.hpp"
#include "image.hpp"

namespace trayh {

void set_error(const std::string& msg);   // capi_host.cpp

struct LoadError : std::runtime_error {
    int code;
    LoadError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] static void fail(int code, const std::string& m) { throw LoadError(code, m); }

// ------------------------------------------------------------------ loaded (pre-flatten) scene

struct SplineLevel {   // BSpline<Keyframe> (animated_transform.rs:15-33)
    std::vector<Keyframe> kfs;
    std::vector<float> knots;
    uint32_t degree = 0;
};
struct AnimXform {
    std::vector<SplineLevel> levels;   // object first, then parents
    static AnimXform unanimated(const Xform& t) {   // animated_transform.rs:34-37
        AnimXform a;
        SplineLevel l;
        l.kfs.push_back(decompose(t));
        l.knots = {0.0f, 1.0f};
        l.degree = 0;
        a.levels.push_back(l);
        return a;
    }
    bool is_animated() const {   // animated_transform.rs:73-75 (AND over the stack, quirk Q12)
        if (levels.empty()) return true;
        bool b = true;
        for (auto& l : levels) b = b && l.kfs.size() > 1;
        return b;
    }
    bool any_animated() const {
        for (auto& l : levels) if (l.kfs.size() > 1) return true;
        return false;
    }
    // A level whose value is the same for every time in [open, close]: a single control point, or a spline whose knot
    // domain the interval does not enter (transform() clamps the time, animated_transform.rs:49-50)
    static bool level_const_over(const SplineLevel& l, float open, float close, float& t_eval) {
        t_eval = open;
        if (l.kfs.size() == 1) return true;
        float lo = l.knots[l.degree], hi = l.knots[l.knots.size() - 1 - l.degree];
        if (open == close) { t_eval = clampf(open, lo, hi); return true; }
        if (open >= hi) { t_eval = hi; return true; }
        // frame_time = (close - open) * u + open (camera.rs:155) can round one ulp past `close`: keep a margin
        float close_max = std::nextafter(std::nextafter(close, INFINITY), INFINITY);
        if (close_max <= lo) { t_eval = lo; return true; }
        return false;
    }
    bool varies_over(float open, float close) const {
        float t;
        for (auto& l : levels) if (!level_const_over(l, open, close, t)) return true;
        return false;
    }
    static Xform level_transform(const SplineLevel& l, float time) {
        if (l.kfs.size() == 1) return l.kfs[0].transform();
        size_t nk = l.knots.size();
        float t_val = clampf(time, l.knots[l.degree], l.knots[nk - 1 - l.degree]);
        return bspline_point(l.kfs.data(), l.knots.data(), nk, l.degree, t_val, kf_interpolate).transform();
    }
    // AnimatedTransform::transform (animated_transform.rs:40-56)
    Xform transform(float time) const {
        Xform t = Xform::identity();
        for (auto& l : levels) {
            if (l.kfs.size() == 1) { t = l.kfs[0].transform() * t; continue; }
            size_t nk = l.knots.size();
            float lo = l.knots[l.degree], hi = l.knots[nk - 1 - l.degree];   // BSpline::knot_domain
            float t_val = clampf(time, lo, hi);
            Keyframe k = bspline_point(l.kfs.data(), l.knots.data(), nk, l.degree, t_val, kf_interpolate);
            t = k.transform() * t;
        }
        return t;
    }
    // AnimatedTransform::animation_bounds (animated_transform.rs:58-71): 128 time samples, only when EVERY level is animated
    BBox animation_bounds(const BBox& b, float start, float end) const {
        if (!is_animated()) return transform(start).bbox(b);
        BBox ret;
        for (int i = 0; i < 128; ++i) {
            float time = lerpf((float)i / 127.0f, start, end);
            ret = ret.box_union(transform(time).bbox(b));
        }
        return ret;
    }
};

struct ColorKey { float c[4]; float time; };

struct HostMesh {
    std::string key;   // file + '\n' + model
    std::string file, model;   // build_meshes() fills bvh / verts / attrs after the whole scene is parsed
    BvhBuild bvh;
    std::vector<TrayTriVerts> verts;   // leaf order; an AnimatedMesh: keyframe after keyframe, each in the leaf order of the ONE tree
    std::vector<TrayTriAttrs> attrs;
    // AnimatedMesh (geometry/animated_mesh.rs): `file` is the first keyframe's; every keyframe's file and time, ascending
    std::vector<std::string> key_files;
    std::vector<float> key_times;
};

struct HostInstance {
    std::string name;
    uint32_t kind = TRAY_INST_RECEIVER;
    uint32_t geom_type = TRAY_GEOM_NONE;
    uint32_t mesh_id = 0;
    uint32_t material_id = 0xffffffffu;
    float geom_params[4] = {0, 0, 0, 0};
    std::vector<ColorKey> emission;
    AnimXform xf;
};

struct HostCamera {
    AnimXform cam_world;
    float fov = 0;
    bool animated_fov = false;
    std::vector<float> fovs, fov_knots;   // CameraFov::Animated(BSpline<f32>)
    uint32_t fov_degree = 0;
    float shutter_size = 0.5f;
    uint32_t active_at = 0;
};

}  // namespace trayh

using namespace trayh;

struct TrayHostScene {
    // film (scene.rs:185-206)
    uint32_t width = 0, height = 0, spp = 0, frames = 0, start_frame = 0, end_frame = 0;
    float scene_time = 0;
    TrayFilm film{};
    std::vector<HostCamera> cameras;
    uint32_t min_depth = 0, max_depth = 0, integrator = TRAY_INTEGRATOR_PATH;
    std::vector<TrayMaterial> materials;
    std::map<std::string, uint32_t> material_names;
    std::vector<TrayTexture> textures;          // image / animated_image / movie (scene.rs:317-394)
    std::vector<TrayTexFrame> tex_frames;
    std::vector<uint8_t> tex_data;
    std::map<std::string, uint32_t> texture_names;
    std::vector<TrayMerlTable> merl_tables;
    std::vector<float> merl_data;
    std::vector<HostMesh> meshes;
    std::vector<HostInstance> instances;

    // flattened view for one frame
    TrayFlatScene flat{};
    std::vector<TrayInstance> f_instances;
    std::vector<TrayBvhNode> f_top_nodes, f_mesh_nodes;
    std::vector<uint32_t> f_top_order, f_lights;
    std::vector<TrayMesh> f_meshes;
    std::vector<TrayTriVerts> f_verts;
    std::vector<TrayTriAttrs> f_attrs;
    std::vector<TrayMeshKeys> f_mesh_keys;
    std::vector<float> f_key_times;
    bool f_meshes_done = false, f_any_keys = false, f_trees_checked = false;   // the mesh arrays above are filled by the first flatten and kept
    std::vector<TrayXformLevel> f_levels;
    std::vector<TrayKeyframe> f_keyframes;
    std::vector<float> f_knots;
    std::vector<TrayColorKey> f_color_keys;
};

namespace trayh {

// ------------------------------------------------------------------ JSON helpers (expect -> error)

static const Json& need(const Json& o, const char* key, const char* msg) {
    const Json* v = o.get(key);
    if (!v) fail(TRAY_E_PARSE, msg);
    return *v;
}
static float need_f32(const Json& o, const char* key, const char* miss, const char* bad) {
    double d;
    if (!need(o, key, miss).as_f64(d)) fail(TRAY_E_PARSE, bad);
    return (float)d;
}
static uint64_t need_u64(const Json& o, const char* key, const char* miss, const char* bad) {
    unsigned long long u;
    if (!need(o, key, miss).as_u64(u)) fail(TRAY_E_PARSE, bad);
    return u;
}
static const std::string& need_str(const Json& o, const char* key, const char* miss, const char* bad) {
    const Json& v = need(o, key, miss);
    if (!v.is_string()) fail(TRAY_E_PARSE, bad);
    return v.str;
}

static bool load_vec3(const Json& e, V3& out) {   // scene.rs:658-694
    if (!e.is_array() || e.arr.size() != 3) return false;
    for (int i = 0; i < 3; ++i) {
        double d;
        if (!e.arr[i].as_f64(d)) return false;
        out[i] = (float)d;
    }
    return true;
}
static bool load_color(const Json& e, float c[4]) {   // scene.rs:698-718
    if (!e.is_array() || (e.arr.size() != 3 && e.arr.size() != 4)) return false;
    float v[4];
    for (size_t i = 0; i < e.arr.size(); ++i) {
        double d;
        if (!e.arr[i].as_f64(d)) return false;
        v[i] = (float)d;
    }
    c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = 1.0f;   // Colorf::new sets a = 1
    if (e.arr.size() == 4)
        for (int i = 0; i < 4; ++i) c[i] = c[i] * v[3];
    return true;
}

static bool load_animated_color(const Json& e, std::vector<ColorKey>& out) {   // scene.rs:722-747
    if (!e.is_array() || e.arr.empty()) return false;
    if (e.arr[0].is_number()) {
        ColorKey k{};
        if (!load_color(e, k.c)) return false;
        k.time = 0.0f;
        out.push_back(k);
        return true;
    }
    for (auto& c : e.arr) {
        ColorKey k{};
        k.time = need_f32(c, "time", "A time must be specified for a color keyframe", "Time for color keyframe must be a number");
        if (!load_color(need(c, "color", "A color must be specified for a color keyframe"), k.c))
            fail(TRAY_E_PARSE, "A valid color is required for a color keyframe");
        out.push_back(k);
    }
    std::stable_sort(out.begin(), out.end(), [](const ColorKey& a, const ColorKey& b) { return a.time < b.time; });
    return true;
}

static Xform load_transform(const Json& e) {   // scene.rs:751-821; each op pre-multiplies
    if (!e.is_array()) fail(TRAY_E_PARSE, "Invalid transform specified");
    Xform t = Xform::identity();
    for (auto& op : e.arr) {
        const std::string& ty = need_str(op, "type", "A type is required for a transform", "Transform type must be a string");
        if (ty == "translate") {
            V3 v;
            if (!load_vec3(need(op, "translation", "A translation vector is required for translate"), v))
                fail(TRAY_E_PARSE, "Invalid vector specified for translation direction");
            t = Xform::translate(v) * t;
        } else if (ty == "scale") {
            const Json& s = need(op, "scaling", "A scaling value or vector is required for scale");
            V3 v;
            if (s.is_array()) {
                if (!load_vec3(s, v)) fail(TRAY_E_PARSE, "Invalid vector specified for scaling vector");
            } else if (s.is_number()) {
                float f = (float)s.num;
                v = V3(f, f, f);
            } else {
                fail(TRAY_E_PARSE, "Scaling value should be an array of 3 floats or a single float");
            }
            t = Xform::scale(v) * t;
        } else if (ty == "rotate_x" || ty == "rotate_y" || ty == "rotate_z") {
            float r = need_f32(op, "rotation", "A rotation in degrees is required", "rotation must be a number");
            Xform rot = ty == "rotate_x" ? Xform::rotate_x(r) : (ty == "rotate_y" ? Xform::rotate_y(r) : Xform::rotate_z(r));
            t = rot * t;
        } else if (ty == "rotate") {
            float r = need_f32(op, "rotation", "A rotation in degrees is required for rotate", "rotation for rotate must be a number");
            V3 axis;
            if (!load_vec3(need(op, "axis", "An axis vector is required for rotate"), axis))
                fail(TRAY_E_PARSE, "Invalid vector specified for rotation axis");
            t = Xform::rotate(axis, r) * t;
        } else if (ty == "matrix") {
            const Json& mat = need(op, "matrix", "The rows of the matrix are required for matrix transform");
            if (!mat.is_array()) fail(TRAY_E_PARSE, "The rows should be an array");
            M4 m = M4::zero();
            size_t n = 0;
            for (auto& row : mat.arr) {
                if (!row.is_array()) fail(TRAY_E_PARSE, "Each row of the matrix transform must be an array, specifying the row");
                if (row.arr.size() != 4) fail(TRAY_E_PARSE, "Each row of the transformation matrix must contain 4 elements");
                for (auto& el : row.arr) {
                    double d;
                    if (!el.as_f64(d)) fail(TRAY_E_PARSE, "Each element of a matrix row must be a float");
                    if (n < 16) m.m[n] = (float)d;
                    ++n;
                }
            }
            bool ok = true;
            Xform mt = Xform::from_mat(m, &ok);
            if (!ok) fail(TRAY_E_INVALID, "matrix transform is singular (reference asserts det != 0)");
            t = mt * t;
        } else {
            fail(TRAY_E_PARSE, "Unrecognized transform type '" + ty + "'");
        }
    }
    return t;
}

static AnimXform load_keyframes(const Json& e) {   // scene.rs:825-850 + AnimatedTransform::with_keyframes
    const Json& pts = need(e, "control_points", "Control points are required for bspline keyframes");
    const Json& knots = need(e, "knots", "knots are required for bspline keyframes");
    if (!pts.is_array() || !knots.is_array()) fail(TRAY_E_PARSE, "Invalid keyframes specified");
    SplineLevel l;
    for (auto& p : pts.arr)
        l.kfs.push_back(decompose(load_transform(need(p, "transform", "A transform is required for a keyframe"))));
    for (auto& k : knots.arr) {
        double d;
        if (!k.as_f64(d)) fail(TRAY_E_PARSE, "Knots must be numbers");
        l.knots.push_back((float)d);
    }
    l.degree = 3;
    if (const Json* d = e.get("degree")) {
        unsigned long long u;
        if (!d->as_u64(u)) fail(TRAY_E_PARSE, "Curve degree must be a positive integer");
        l.degree = (uint32_t)u;
    }
    // BSpline::new (bspline 0.2.2): panics restated as load errors; knots are sorted
    if (l.kfs.size() <= l.degree) fail(TRAY_E_INVALID, "Too few control points for curve");
    if (l.knots.size() != l.kfs.size() + l.degree + 1)
        fail(TRAY_E_INVALID, "Invalid number of knots, got " + std::to_string(l.knots.size()) + ", expected " + std::to_string(l.kfs.size() + l.degree + 1));
    if (l.degree > 7) fail(TRAY_E_UNSUPPORTED, "B-spline keyframes of degree > 7 are not supported");
    std::sort(l.knots.begin(), l.knots.end());
    for (size_t i = 1; i < l.kfs.size(); ++i)   // shortest-arc flip, animated_transform.rs:26-31
        if (qdot(l.kfs[i - 1].rotation, l.kfs[i].rotation) < 0.0f) {
            l.kfs[i].rotation.v = -l.kfs[i].rotation.v;
            l.kfs[i].rotation.w = -l.kfs[i].rotation.w;
        }
    AnimXform a;
    a.levels.push_back(l);
    return a;
}

static AnimXform load_object_transform(const Json& o, const std::string& name) {   // scene.rs:526-535
    if (const Json* k = o.get("keyframes")) return load_keyframes(*k);
    const Json* t = o.get("transform");
    if (!t) fail(TRAY_E_PARSE, "No keyframes or transform specified for object " + name);
    return AnimXform::unanimated(load_transform(*t));
}

// ------------------------------------------------------------------ film / filter / camera

static float mn_weight_1d(float x, float b, float c) {   // mitchell_netravali.rs:35-50
    float ax = std::fabs(x);
    float x3 = std::pow(ax, 3.0f), x2 = std::pow(ax, 2.0f);   // f32::powf
    if (x >= 2.0f) return 0.0f;
    if (x >= 1.0f)
        return 1.0f / 6.0f * ((-b - 6.0f * c) * x3 + (6.0f * b + 30.0f * c) * x2 + (-12.0f * b - 48.0f * c) * ax + (8.0f * b + 24.0f * c));
    return 1.0f / 6.0f * ((12.0f - 9.0f * b - 6.0f * c) * x3 + (-18.0f + 12.0f * b + 6.0f * c) * x2 + (6.0f - 2.0f * b));
}

static void load_film(const Json& e, TrayHostScene& s) {   // scene.rs:185-228, render_target.rs:41-75
    s.width = (uint32_t)need_u64(e, "width", "The film must specify the image width", "Image width must be a number");
    s.height = (uint32_t)need_u64(e, "height", "The film must specify the image height", "Image height must be a number");
    s.spp = (uint32_t)need_u64(e, "samples", "The film must specify the number of samples per pixel", "Samples per pixel must be a number");
    s.start_frame = (uint32_t)need_u64(e, "start_frame", "The film must specify the starting frame", "Start frame must be a number");
    s.end_frame = (uint32_t)need_u64(e, "end_frame", "The film must specify the frame to end on", "End frame must be a number");
    if (s.end_frame < s.start_frame) fail(TRAY_E_INVALID, "End frame must be greater or equal to the starting frame");
    s.frames = (uint32_t)need_u64(e, "frames", "The film must specify the total number of frames", "Frames must be a number");
    s.scene_time = need_f32(e, "scene_time", "The film must specify the overall scene time", "Scene time must be a number");
    if (s.width == 0 || s.height == 0 || s.frames == 0) fail(TRAY_E_INVALID, "film width, height and frames must be non-zero");
    // RenderTarget::new panics unless divisible by the (2,2) lock size; BlockQueue::new unless by (8,8)
    if (s.width % 8 != 0 || s.height % 8 != 0) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "Image with dimension (%u, %u) not evenly divided by blocks of (8, 8)", s.width, s.height);
        fail(TRAY_E_INVALID, buf);
    }
    const Json& f = need(e, "filter", "The film must specify a reconstruction filter");
    float w = need_f32(f, "width", "The filter must specify the filter width", "Filter width must be a number");
    float h = need_f32(f, "height", "The filter must specify the filter height", "Filter height must be a number");
    const std::string& ty = need_str(f, "type", "A type is required for the filter", "Filter type must be a string");
    TrayFilm& film = s.film;
    film.width = s.width; film.height = s.height;
    film.filter_w = w; film.filter_h = h; film.inv_w = 1.0f / w; film.inv_h = 1.0f / h;
    film.filter_pixel_w = (int32_t)std::floor(w / 0.5f);
    film.filter_pixel_h = (int32_t)std::floor(h / 0.5f);
    const int N = TRAY_FILTER_TABLE_SIZE;
    if (ty == "mitchell_netravali") {
        float b = need_f32(f, "b", "A b parameter is required for the Mitchell-Netravali filter", "b must be a number");
        float c = need_f32(f, "c", "A c parameter is required for the Mitchell-Netravali filter", "c must be a number");
        b = b < 0.0f ? 0.0f : (b > 1.0f ? 1.0f : b);
        c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
        for (int y = 0; y < N; ++y) {
            float fy = ((float)y + 0.5f) * h / (float)N;
            for (int x = 0; x < N; ++x) {
                float fx = ((float)x + 0.5f) * w / (float)N;
                film.table_x[x] = mn_weight_1d(2.0f * fx * film.inv_w, b, c);
                film.table_y[y] = mn_weight_1d(2.0f * fy * film.inv_h, b, c);
                film.table[y * N + x] = film.table_x[x] * film.table_y[y];
            }
        }
        film.separable = 1;
    } else if (ty == "gaussian") {   // filter/gaussian.rs
        float alpha = need_f32(f, "alpha", "An alpha parameter is required for the Gaussian filter", "alpha must be a number");
        float ex = std::exp(-alpha * w * w), ey = std::exp(-alpha * h * h);
        for (int y = 0; y < N; ++y) {
            float fy = ((float)y + 0.5f) * h / (float)N;
            for (int x = 0; x < N; ++x) {
                float fx = ((float)x + 0.5f) * w / (float)N;
                film.table_x[x] = std::fmax(0.0f, std::exp(-alpha * fx * fx) - ex);
                film.table_y[y] = std::fmax(0.0f, std::exp(-alpha * fy * fy) - ey);
                film.table[y * N + x] = film.table_x[x] * film.table_y[y];
            }
        }
        film.separable = 1;
    } else {
        fail(TRAY_E_PARSE, "Unrecognized filter type " + ty + "!");
    }
}

static HostCamera load_camera(const Json& e) {   // scene.rs:251-293
    HostCamera c;
    if (const Json* s = e.get("shutter_size")) {
        double d;
        if (!s->as_f64(d)) fail(TRAY_E_PARSE, "Shutter size should be a float from 0 to 1");
        c.shutter_size = (float)d;
    }
    if (const Json* a = e.get("active_at")) {
        unsigned long long u;
        if (!a->as_u64(u)) fail(TRAY_E_PARSE, "The camera activation frame 'active_at' must be an unsigned int");
        c.active_at = (uint32_t)u;
    }
    if (const Json* k = e.get("keyframes")) {
        c.cam_world = load_keyframes(*k);
    } else if (const Json* t = e.get("transform")) {
        c.cam_world = AnimXform::unanimated(load_transform(*t));
    } else {
        V3 pos, target, up;
        if (!load_vec3(need(e, "position", "The camera must specify a position"), pos)) fail(TRAY_E_PARSE, "position must be an array of 3 floats");
        if (!load_vec3(need(e, "target", "The camera must specify a target"), target)) fail(TRAY_E_PARSE, "target must be an array of 3 floats");
        if (!load_vec3(need(e, "up", "The camera must specify an up vector"), up)) fail(TRAY_E_PARSE, "up must be an array of 3 floats");
        c.cam_world = AnimXform::unanimated(Xform::look_at(pos, target, up));
    }
    const Json& fov = need(e, "fov", "The camera must specify a field of view");
    if (fov.is_array()) {   // Camera::animated_fov (camera.rs:97-125): BSpline<f32> sampled once per frame (camera.rs:130-141)
        c.animated_fov = true;
        const Json& kn = need(e, "fov_knots", "Animated field of view must specify spline knots");
        if (!kn.is_array()) fail(TRAY_E_PARSE, "Fov spline knots must be an array");
        unsigned long long deg;
        if (!need(e, "fov_spline_degree", "Animated fov spline must have degree").as_u64(deg)) fail(TRAY_E_PARSE, "Animated fov spline degree must be a u64");
        for (auto& v : fov.arr) { double d; if (!v.as_f64(d)) fail(TRAY_E_PARSE, "fovs must be a number"); c.fovs.push_back((float)d); }
        for (auto& v : kn.arr) { double d; if (!v.as_f64(d)) fail(TRAY_E_PARSE, "fov knots must be a number"); c.fov_knots.push_back((float)d); }
        c.fov_degree = (uint32_t)deg;
        if (c.fovs.size() <= c.fov_degree) fail(TRAY_E_INVALID, "Too few control points for curve");
        if (c.fov_knots.size() != c.fovs.size() + c.fov_degree + 1)
            fail(TRAY_E_INVALID, "Invalid number of knots, got " + std::to_string(c.fov_knots.size()) + ", expected " + std::to_string(c.fovs.size() + c.fov_degree + 1));
        if (c.fov_degree > 7) fail(TRAY_E_UNSUPPORTED, "B-spline keyframes of degree > 7 are not supported");
        std::sort(c.fov_knots.begin(), c.fov_knots.end());
        c.fov = c.fovs[0];
    } else {
        double d;
        if (!fov.as_f64(d)) fail(TRAY_E_PARSE, "Camera fov must be a number");
        c.fov = (float)d;
    }
    return c;
}

// ------------------------------------------------------------------ materials (scene.rs:404-511)

// LoadedTextures::find_color / find_scalar (scene.rs:57-88): a string names a loaded texture, an array / a number is a constant
static uint32_t find_texture(const TrayHostScene& s, const std::string& tex, const std::string& name, const char* key, const char* what) {
    auto it = s.texture_names.find(tex);
    if (it == s.texture_names.end()) fail(TRAY_E_INVALID, "Error loading material '" + name + "': Invalid color specified for " + key + " of " + what);
    return it->second;
}
static void color_param(const TrayHostScene& s, const Json& m, const char* key, const std::string& name, const char* what, float out[4], uint32_t& tex) {
    const Json* v = m.get(key);
    if (!v) fail(TRAY_E_PARSE, std::string(key) + " color/texture name is required for " + what);
    tex = TRAY_NO_TEXTURE;
    if (v->is_string()) { tex = find_texture(s, v->str, name, key, what); return; }
    if (!v->is_array()) fail(TRAY_E_PARSE, "Invalid JSON type for colorf texture");
    if (!load_color(*v, out)) fail(TRAY_E_PARSE, "Error loading material '" + name + "': Invalid color specified for " + key + " of " + what);
}
static float scalar_param(const TrayHostScene& s, const Json& m, const char* key, const std::string& name, const char* what, uint32_t& tex) {
    const Json* v = m.get(key);
    if (!v) fail(TRAY_E_PARSE, std::string(key) + " color/texture name is required for " + what);
    tex = TRAY_NO_TEXTURE;
    if (v->is_string()) { tex = find_texture(s, v->str, name, key, what); return 0.0f; }
    if (!v->is_number()) fail(TRAY_E_PARSE, "Invalid JSON type for scalar texture");
    return (float)v->num;
}

static std::string join_path(const std::string& base, const std::string& rel) {
    if (!rel.empty() && rel[0] == '/') return rel;
    if (base.empty()) return rel;
    return base + "/" + rel;
}

static uint32_t load_merl(TrayHostScene& s, const std::string& path) {   // material/merl.rs:51-84
    std::ifstream f(path, std::ios::binary);
    if (!f) fail(TRAY_E_IO, "material::Merl::load_file - failed to open \"" + path + "\"");
    int32_t dims[3];
    f.read(reinterpret_cast<char*>(dims), sizeof dims);
    if (!f || dims[0] != 90 || dims[1] != 90 || dims[2] != 180) fail(TRAY_E_PARSE, "material::Merl::load_file - Invalid MERL file header, aborting");
    const size_t n_vals = 90u * 90u * 180u;
    TrayMerlTable t{};
    t.offset = s.merl_data.size();
    t.n_theta_h = 90; t.n_theta_d = 90; t.n_phi_d = 180;
    s.merl_data.resize(s.merl_data.size() + 3 * n_vals, 0.0f);
    float* brdf = s.merl_data.data() + t.offset;
    const double scaling[3] = {1.0 / 1500.0, 1.0 / 1500.0, 1.66 / 1500.0};   // quirk Q10
    std::vector<double> plane(n_vals);
    for (int c = 0; c < 3; ++c) {
        f.read(reinterpret_cast<char*>(plane.data()), (std::streamsize)(n_vals * sizeof(double)));
        if (!f) fail(TRAY_E_PARSE, "material::Merl::load_file - truncated MERL file");
        for (size_t i = 0; i < n_vals; ++i) {
            float x = (float)(plane[i] * scaling[c]);
            brdf[3 * i + c] = std::fmax(0.0f, x);
        }
    }
    s.merl_tables.push_back(t);
    return (uint32_t)s.merl_tables.size() - 1;
}

// load_textures (scene.rs:317-394)
static void add_frame(TrayHostScene& s, const std::string& path, float time) {
    ImageRGBA8 img;
    std::string err;
    if (!load_image(path, img, err)) fail(TRAY_E_IO, "Failed to load image file: " + err);
    TrayTexFrame fr{};
    fr.time = time; fr.width = img.width; fr.height = img.height; fr.offset = s.tex_data.size();
    s.tex_data.insert(s.tex_data.end(), img.px.begin(), img.px.end());
    s.tex_frames.push_back(fr);
}
static void load_textures(TrayHostScene& s, const Json& e, const std::string& base) {
    if (!e.is_array()) fail(TRAY_E_PARSE, "The 'textures' must be an array of textures to load");
    for (size_t i = 0; i < e.arr.size(); ++i) {
        const Json& t = e.arr[i];
        const Json* nm = t.get("name");
        if (!nm) fail(TRAY_E_PARSE, "Error loading texture #" + std::to_string(i) + ": A name is required");
        if (!nm->is_string()) fail(TRAY_E_PARSE, "Error loading texture #" + std::to_string(i) + ": name must be a string");
        const std::string name = nm->str;
        const Json* tyj = t.get("type");
        if (!tyj) fail(TRAY_E_PARSE, "Error loading material '" + name + "': A texture type is required");
        if (!tyj->is_string()) fail(TRAY_E_PARSE, "Error loading material '" + name + "': Texture type must be a string");
        const std::string& ty = tyj->str;
        if (s.texture_names.count(name)) fail(TRAY_E_INVALID, "Error loading texture '" + name + "': name conflicts with an existing entry");
        TrayTexture tex{};
        tex.first_frame = (uint32_t)s.tex_frames.size();
        if (ty == "image") {
            add_frame(s, join_path(base, need_str(t, "file", "Image textures must specify an image file", "Image file name must be a string")), 0.0f);
        } else if (ty == "animated_image") {
            const Json* kf = t.get("keyframes");
            if (!kf) fail(TRAY_E_PARSE, "animated_image requires keyframes");
            if (!kf->is_array()) fail(TRAY_E_PARSE, "animated_image keyframes must be an array");
            if (kf->arr.size() < 2) fail(TRAY_E_INVALID, "animated_image must have at least 2 frames");
            for (const Json& f : kf->arr) {
                const std::string& file = need_str(f, "file", "Image textures must specify an image file", "Image file name must be a string");
                const Json* tm = f.get("time");
                if (!tm) fail(TRAY_E_PARSE, "animated_image keyframe requires time");
                if (!tm->is_number()) fail(TRAY_E_PARSE, "animated_image keyframe time must be a number");
                add_frame(s, join_path(base, file), (float)tm->num);
            }
        } else if (ty == "movie") {   // a generated animated_image: <prefix><frame, 5 digits><suffix> played back at `framerate`
            const std::string& prefix = need_str(t, "file_prefix", "A file_prefix for movie is required", "file_prefix for movie must be a string");
            const std::string& suffix = need_str(t, "file_suffix", "A file_suffix for movie is required", "file_suffix for movie must be a string");
            const Json* fr = t.get("frames");
            const Json* rate = t.get("framerate");
            if (!fr) fail(TRAY_E_PARSE, "# of frames for movie texture is required");
            if (!fr->is_number() || fr->num < 0 || fr->num != std::floor(fr->num)) fail(TRAY_E_PARSE, "frames for movie texture must be an int");
            if (!rate) fail(TRAY_E_PARSE, "A framerate for movie is required");
            if (!rate->is_number() || rate->num < 0 || rate->num != std::floor(rate->num)) fail(TRAY_E_PARSE, "framerate for movie must be an int");
            if (fr->num < 2) fail(TRAY_E_INVALID, "assertion failed: frames.len() >= 2");   // AnimatedImage::new
            for (uint64_t k = 0; k < (uint64_t)fr->num; ++k) {
                char num[32];
                std::snprintf(num, sizeof num, "%05llu", (unsigned long long)k);
                add_frame(s, join_path(base, prefix + num + suffix), (float)k / (float)(uint64_t)rate->num);
            }
        } else {
            fail(TRAY_E_PARSE, "Unrecognized texture type '" + ty + "' for texture '" + name + "'");
        }
        tex.n_frames = (uint32_t)s.tex_frames.size() - tex.first_frame;
        // AnimatedImage::active_keyframes (animated_image.rs:21-34) binary-searches the key times with partial_cmp().unwrap(): a NaN
        // time panics there at the first lookup, unsorted times make the search meaningless. Both are diagnosed here, at the scene
        // file, instead of at tray_scene_create (whose validator only sees indices)
        for (uint32_t k = 0; k < tex.n_frames; ++k) {
            const float tk = s.tex_frames[tex.first_frame + k].time;
            if (!std::isfinite(tk)) fail(TRAY_E_INVALID, "Error loading texture '" + name + "': keyframe time " + std::to_string(k) + " is not a finite number" +
                                                          (ty == "movie" ? " (framerate 0?)" : ""));
            if (k > 0 && !(s.tex_frames[tex.first_frame + k - 1].time < tk))
                fail(TRAY_E_INVALID, "Error loading texture '" + name + "': keyframe times must increase (keyframe " + std::to_string(k) + ")");
        }
        s.texture_names[name] = (uint32_t)s.textures.size();
        s.textures.push_back(tex);
    }
}

static void load_materials(TrayHostScene& s, const Json& e, const std::string& base) {
    if (!e.is_array()) fail(TRAY_E_PARSE, "The materials must be an array of materials used");
    for (size_t i = 0; i < e.arr.size(); ++i) {
        const Json& m = e.arr[i];
        const Json* nm = m.get("name");
        if (!nm || !nm->is_string()) fail(TRAY_E_PARSE, "Error loading material #" + std::to_string(i) + ": A name is required");
        const std::string name = nm->str;
        const std::string& ty = need_str(m, "type", "a type is required", "type must be a string");
        if (s.material_names.count(name)) fail(TRAY_E_INVALID, "Error loading material '" + name + "': name conflicts with an existing entry");
        TrayMaterial mat{};
        mat.tex_c0 = mat.tex_c1 = mat.tex_f0 = mat.tex_f1 = TRAY_NO_TEXTURE;
        if (ty == "glass") {
            mat.kind = TRAY_MAT_GLASS;
            color_param(s, m, "reflect", name, "glass", mat.c0, mat.tex_c0);
            color_param(s, m, "transmit", name, "glass", mat.c1, mat.tex_c1);
            mat.f0 = scalar_param(s, m, "eta", name, "glass", mat.tex_f0);
        } else if (ty == "rough_glass") {
            mat.kind = TRAY_MAT_ROUGH_GLASS;
            color_param(s, m, "reflect", name, "rough glass", mat.c0, mat.tex_c0);
            color_param(s, m, "transmit", name, "rough glass", mat.c1, mat.tex_c1);
            mat.f0 = scalar_param(s, m, "eta", name, "rough glass", mat.tex_f0);
            mat.f1 = scalar_param(s, m, "roughness", name, "rough glass", mat.tex_f1);
        } else if (ty == "matte") {
            mat.kind = TRAY_MAT_MATTE;
            color_param(s, m, "diffuse", name, "matte", mat.c0, mat.tex_c0);
            mat.f0 = scalar_param(s, m, "roughness", name, "matte", mat.tex_f0);
        } else if (ty == "merl") {
            mat.kind = TRAY_MAT_MERL;
            const std::string& file = need_str(m, "file", "A filename containing the MERL material data is required", "The MERL file must be a string");
            mat.table = load_merl(s, join_path(base, file));
        } else if (ty == "metal") {
            mat.kind = TRAY_MAT_METAL;
            color_param(s, m, "refractive_index", name, "metal", mat.c0, mat.tex_c0);
            color_param(s, m, "absorption_coefficient", name, "metal", mat.c1, mat.tex_c1);
            mat.f0 = scalar_param(s, m, "roughness", name, "metal", mat.tex_f0);
        } else if (ty == "plastic") {
            mat.kind = TRAY_MAT_PLASTIC;
            color_param(s, m, "diffuse", name, "plastic", mat.c0, mat.tex_c0);
            color_param(s, m, "gloss", name, "plastic", mat.c1, mat.tex_c1);
            mat.f0 = scalar_param(s, m, "roughness", name, "plastic", mat.tex_f0);
        } else if (ty == "specular_metal") {
            mat.kind = TRAY_MAT_SPECULAR_METAL;
            color_param(s, m, "refractive_index", name, "specular metal", mat.c0, mat.tex_c0);
            color_param(s, m, "absorption_coefficient", name, "specular metal", mat.c1, mat.tex_c1);
        } else {
            fail(TRAY_E_PARSE, "Error parsing material '" + name + "': unrecognized type '" + ty + "'");
        }
        if (const Json* mf = m.get("microfacet")) {   // extension: the reference's materials always use Beckmann; its GGX (microfacet/ggx.rs) is selectable here
            if (!mf->is_string() || (mf->str != "beckmann" && mf->str != "ggx")) fail(TRAY_E_PARSE, "Error loading material '" + name + "': microfacet must be \"beckmann\" or \"ggx\"");
            if (mat.kind != TRAY_MAT_PLASTIC && mat.kind != TRAY_MAT_METAL && mat.kind != TRAY_MAT_ROUGH_GLASS)
                fail(TRAY_E_INVALID, "Error loading material '" + name + "': only plastic, metal and rough_glass have a microfacet distribution");
            mat.microfacet = mf->str == "ggx" ? TRAY_MF_GGX : TRAY_MF_BECKMANN;
        }
        s.material_names[name] = (uint32_t)s.materials.size();
        s.materials.push_back(mat);
    }
}

// ------------------------------------------------------------------ OBJ (tobj 0.1.6 semantics) + Mesh::new

struct ObjModel {
    std::string name;
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> indices;
};

// Parses v / vt / vn / f / o / g. Like tobj: each o/g starts a new model; every distinct (v,vt,vn)
// triple of a model becomes one vertex in order of first use; quads and n-gons are fan-triangulated
// (a,b,c),(a,c,d),... (src/geometry/mesh.rs:50-78 consumes the result).
static std::vector<ObjModel> parse_obj(const std::string& path) {
    std::ifstream f(path);
    if (!f) fail(TRAY_E_IO, "Failed to load \"" + path + "\"");
    std::vector<float> pos, tex, nrm;
    std::vector<ObjModel> models;
    struct Idx { long v, vt, vn; bool operator<(const Idx& o) const { return v != o.v ? v < o.v : (vt != o.vt ? vt < o.vt : vn < o.vn); } };
    std::vector<std::vector<Idx>> faces;
    std::string name = "unnamed_object";
    auto flush = [&]() {
        if (faces.empty()) return;
        ObjModel m;
        m.name = name;
        std::map<Idx, uint32_t> index_map;
        auto add_vertex = [&](const Idx& ix) {
            auto it = index_map.find(ix);
            if (it != index_map.end()) { m.indices.push_back(it->second); return; }
            if (ix.v < 0 || (size_t)ix.v * 3 + 2 >= pos.size()) fail(TRAY_E_PARSE, "OBJ face references a missing position in " + path);
            for (int k = 0; k < 3; ++k) m.positions.push_back(pos[(size_t)ix.v * 3 + k]);
            if (ix.vt >= 0) {
                if ((size_t)ix.vt * 2 + 1 >= tex.size()) fail(TRAY_E_PARSE, "OBJ face references a missing texcoord in " + path);
                for (int k = 0; k < 2; ++k) m.texcoords.push_back(tex[(size_t)ix.vt * 2 + k]);
            }
            if (ix.vn >= 0) {
                if ((size_t)ix.vn * 3 + 2 >= nrm.size()) fail(TRAY_E_PARSE, "OBJ face references a missing normal in " + path);
                for (int k = 0; k < 3; ++k) m.normals.push_back(nrm[(size_t)ix.vn * 3 + k]);
            }
            uint32_t next = (uint32_t)index_map.size();
            index_map[ix] = next;
            m.indices.push_back(next);
        };
        for (auto& face : faces) {
            if (face.size() < 3) continue;
            for (size_t c = 2; c < face.size(); ++c) { add_vertex(face[0]); add_vertex(face[c - 1]); add_vertex(face[c]); }
        }
        models.push_back(std::move(m));
        faces.clear();
    };
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") { float x, y, z; if (!(ss >> x >> y >> z)) fail(TRAY_E_PARSE, "bad 'v' line in " + path); pos.insert(pos.end(), {x, y, z}); }
        else if (tag == "vt") { float u = 0, v = 0; ss >> u >> v; tex.insert(tex.end(), {u, v}); }
        else if (tag == "vn") { float x, y, z; if (!(ss >> x >> y >> z)) fail(TRAY_E_PARSE, "bad 'vn' line in " + path); nrm.insert(nrm.end(), {x, y, z}); }
        else if (tag == "f") {
            std::vector<Idx> face;
            std::string tok;
            while (ss >> tok) {
                long v[3] = {0, 0, 0};
                bool has[3] = {false, false, false};
                size_t start = 0;
                for (int k = 0; k < 3 && start <= tok.size(); ++k) {
                    size_t slash = tok.find('/', start);
                    std::string part = tok.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (!part.empty()) { v[k] = std::strtol(part.c_str(), nullptr, 10); has[k] = true; }
                    if (slash == std::string::npos) break;
                    start = slash + 1;
                }
                auto fix = [](long i, bool h, size_t count) -> long { return !h ? -1 : (i < 0 ? (long)count + i : i - 1); };
                face.push_back({fix(v[0], has[0], pos.size() / 3), fix(v[1], has[1], tex.size() / 2), fix(v[2], has[2], nrm.size() / 3)});
            }
            faces.push_back(std::move(face));
        } else if (tag == "o" || tag == "g") {
            flush();
            std::string rest;
            std::getline(ss, rest);
            size_t b = rest.find_first_not_of(" \t");
            name = b == std::string::npos ? std::string("unnamed_object") : rest.substr(b);
        }
    }
    flush();
    return models;
}

static uint32_t get_mesh(TrayHostScene& s, const std::string& file, const std::string& model) {   // scene.rs:604-625
    std::string key = file + "\n" + model;
    for (size_t i = 0; i < s.meshes.size(); ++i)
        if (s.meshes[i].key == key) return (uint32_t)i;
    HostMesh hm;
    hm.key = key; hm.file = file; hm.model = model;
    s.meshes.push_back(std::move(hm));
    return (uint32_t)s.meshes.size() - 1;
}

// an AnimatedMesh request (animated_mesh.rs:14-27: "model" + "keyframes": [{"file", "time"}, ...]); two requests with the same files, times
// and model share one
static uint32_t get_animated_mesh(TrayHostScene& s, const std::vector<std::string>& files, const std::vector<float>& times, const std::string& model) {
    std::string key = "animated\n" + model;
    for (size_t k = 0; k < files.size(); ++k) { key += "\n" + files[k] + "@"; key.append(reinterpret_cast<const char*>(&times[k]), sizeof(float)); }
    for (size_t i = 0; i < s.meshes.size(); ++i)
        if (s.meshes[i].key == key) return (uint32_t)i;
    HostMesh hm;
    hm.key = key; hm.file = files[0]; hm.model = model; hm.key_files = files; hm.key_times = times;
    s.meshes.push_back(std::move(hm));
    return (uint32_t)s.meshes.size() - 1;
}

static const ObjModel& find_model(const std::vector<ObjModel>& models, const std::string& model, const std::string& file) {
    for (auto& m : models) {
        if (m.normals.empty() || m.texcoords.empty()) continue;   // mesh.rs:57-61: skipped
        if (m.name == model) {
            if (m.normals.size() / 3 != m.positions.size() / 3 || m.texcoords.size() / 2 != m.positions.size() / 3)
                fail(TRAY_E_PARSE, "model '" + model + "' mixes vertices with and without normals/texcoords");
            return m;
        }
    }
    fail(TRAY_E_INVALID, "Requested model '" + model + "' was not found in \"" + file + "\"");
}

// AnimatedMesh::new (animated_mesh.rs:113-127) over the keyframes' Meshes (Mesh::load_obj each, mesh.rs:44-78): the triangles are
// meshes[0]'s -- its index triples, in file order (BVH::iter walks `geometry`, bvh.rs:131-133) -- and every keyframe contributes its
// positions / normals / texcoords by vertex index. ONE tree: BVH::new(16, tris, times[0], times[1]) over AnimatedTriangle::bounds
// (animated_mesh.rs:176-185): the triangle's vertices at times[0] and at times[1] -- exactly keyframes 0 and 1 (position() at a keyframe's
// own time returns that keyframe's vertex, :73-76).
static void build_animated_mesh(HostMesh& hm, const std::vector<const std::vector<ObjModel>*>& per_key) {
    const size_t n_keys = hm.key_files.size();
    std::vector<const ObjModel*> km(n_keys);
    for (size_t k = 0; k < n_keys; ++k) km[k] = &find_model(*per_key[k], hm.model, hm.key_files[k]);
    const ObjModel& m0 = *km[0];
    const size_t ntri = m0.indices.size() / 3, nvert = m0.positions.size() / 3;
    for (size_t k = 1; k < n_keys; ++k)   // ("the topology of the mesh being animated does not change", animated_mesh.rs:1-3: the reference would index out of bounds)
        if (km[k]->positions.size() / 3 != nvert)
            fail(TRAY_E_INVALID, "animated mesh '" + hm.model + "': keyframe \"" + hm.key_files[k] + "\" has " + std::to_string(km[k]->positions.size() / 3) +
                                     " vertices, the first keyframe " + std::to_string(nvert));
    std::vector<BBox> bounds(ntri);
    auto P = [&](size_t k, uint32_t i) { return V3(km[k]->positions[3 * i], km[k]->positions[3 * i + 1], km[k]->positions[3 * i + 2]); };
    for (size_t t = 0; t < ntri; ++t) {
        const uint32_t ia = m0.indices[3 * t], ib = m0.indices[3 * t + 1], ic = m0.indices[3 * t + 2];
        bounds[t] = BBox(P(0, ia), P(0, ia)).point_union(P(0, ib)).point_union(P(0, ic)).point_union(P(1, ia)).point_union(P(1, ib)).point_union(P(1, ic));
    }
    hm.bvh = build_bvh(bounds, 16);
    hm.verts.resize(n_keys * ntri);
    hm.attrs.resize(n_keys * ntri);
    for (size_t k = 0; k < n_keys; ++k) {
        const ObjModel& m = *km[k];
        for (size_t slot = 0; slot < ntri; ++slot) {
            const uint32_t t = hm.bvh.ordered[slot];
            TrayTriVerts& tv = hm.verts[k * ntri + slot];
            TrayTriAttrs& ta = hm.attrs[k * ntri + slot];
            std::memset(&tv, 0, sizeof tv);
            std::memset(&ta, 0, sizeof ta);
            tv.tri_id = t;
            const uint32_t ia = m0.indices[3 * t], ib = m0.indices[3 * t + 1], ic = m0.indices[3 * t + 2];
            for (int c = 0; c < 3; ++c) {
                tv.pa[c] = m.positions[3 * ia + c]; tv.pb[c] = m.positions[3 * ib + c]; tv.pc[c] = m.positions[3 * ic + c];
                ta.na[c] = m.normals[3 * ia + c]; ta.nb[c] = m.normals[3 * ib + c]; ta.nc[c] = m.normals[3 * ic + c];
            }
            for (int c = 0; c < 2; ++c) {
                ta.ta[c] = m.texcoords[2 * ia + c]; ta.tb[c] = m.texcoords[2 * ib + c]; ta.tc[c] = m.texcoords[2 * ic + c];
            }
        }
    }
}

// Mesh::load_obj + BVH::unanimated for one (file, model) request (mesh.rs:44-78)
static void build_mesh(HostMesh& hm, const std::vector<ObjModel>& models) {
    const ObjModel* found = nullptr;
    for (auto& m : models) {
        if (m.normals.empty() || m.texcoords.empty()) continue;   // mesh.rs:57-61: skipped
        if (m.name == hm.model) { found = &m; break; }
    }
    if (!found) fail(TRAY_E_INVALID, "Requested model '" + hm.model + "' was not found in \"" + hm.file + "\"");
    const ObjModel& m = *found;
    size_t ntri = m.indices.size() / 3, nvert = m.positions.size() / 3;
    if (m.normals.size() / 3 != nvert || m.texcoords.size() / 2 != nvert)
        fail(TRAY_E_PARSE, "model '" + hm.model + "' mixes vertices with and without normals/texcoords");
    std::vector<BBox> bounds(ntri);
    auto P = [&](uint32_t i) { return V3(m.positions[3 * i], m.positions[3 * i + 1], m.positions[3 * i + 2]); };
    for (size_t t = 0; t < ntri; ++t)   // Triangle::bounds, mesh.rs:129-133
        bounds[t] = BBox(P(m.indices[3 * t]), P(m.indices[3 * t])).point_union(P(m.indices[3 * t + 1])).point_union(P(m.indices[3 * t + 2]));
    hm.bvh = build_bvh(bounds, 16);   // BVH::unanimated(16, triangles), mesh.rs:44
    hm.verts.resize(ntri);
    hm.attrs.resize(ntri);
    for (size_t slot = 0; slot < ntri; ++slot) {
        uint32_t t = hm.bvh.ordered[slot];
        TrayTriVerts& tv = hm.verts[slot];
        TrayTriAttrs& ta = hm.attrs[slot];
        std::memset(&tv, 0, sizeof tv);
        std::memset(&ta, 0, sizeof ta);
        tv.tri_id = t;
        const uint32_t ia = m.indices[3 * t], ib = m.indices[3 * t + 1], ic = m.indices[3 * t + 2];
        for (int k = 0; k < 3; ++k) {
            tv.pa[k] = m.positions[3 * ia + k]; tv.pb[k] = m.positions[3 * ib + k]; tv.pc[k] = m.positions[3 * ic + k];
            ta.na[k] = m.normals[3 * ia + k]; ta.nb[k] = m.normals[3 * ib + k]; ta.nc[k] = m.normals[3 * ic + k];
        }
        for (int k = 0; k < 2; ++k) {
            ta.ta[k] = m.texcoords[2 * ia + k]; ta.tb[k] = m.texcoords[2 * ib + k]; ta.tc[k] = m.texcoords[2 * ic + k];
        }
    }
}

// Every OBJ file is parsed once (the reference caches per file too, scene.rs:604-625) and the meshes are built on all host
// cores: the results do not depend on the schedule, each mesh is built from its own model alone.
static void build_meshes(TrayHostScene& s) {
    std::vector<std::string> files;
    for (auto& hm : s.meshes) {
        if (std::find(files.begin(), files.end(), hm.file) == files.end()) files.push_back(hm.file);
        for (auto& kf : hm.key_files)
            if (std::find(files.begin(), files.end(), kf) == files.end()) files.push_back(kf);
    }
    std::vector<std::vector<ObjModel>> parsed(files.size());
    struct Err { int code = TRAY_OK; std::string msg; };
    std::vector<Err> file_err(files.size()), mesh_err(s.meshes.size());
    auto run = [](size_t n, auto&& job) {
        unsigned hw = std::thread::hardware_concurrency();
        size_t workers = std::max<size_t>(1, std::min<size_t>(n, hw ? hw : 1));
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (size_t w = 0; w < workers; ++w)
            pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) job(i); });
        for (auto& t : pool) t.join();
    };
    run(files.size(), [&](size_t i) {
        try { parsed[i] = parse_obj(files[i]); }
        catch (const LoadError& e) { file_err[i].code = e.code; file_err[i].msg = e.what(); }
        catch (const std::exception& e) { file_err[i].code = TRAY_E_PARSE; file_err[i].msg = e.what(); }
    });
    run(s.meshes.size(), [&](size_t i) {
        size_t fi = (size_t)(std::find(files.begin(), files.end(), s.meshes[i].file) - files.begin());
        if (file_err[fi].code != TRAY_OK) { mesh_err[i] = file_err[fi]; return; }
        std::vector<const std::vector<ObjModel>*> per_key;
        for (auto& kf : s.meshes[i].key_files) {
            const size_t ki = (size_t)(std::find(files.begin(), files.end(), kf) - files.begin());
            if (file_err[ki].code != TRAY_OK) { mesh_err[i] = file_err[ki]; return; }
            per_key.push_back(&parsed[ki]);
        }
        try { if (per_key.empty()) build_mesh(s.meshes[i], parsed[fi]); else build_animated_mesh(s.meshes[i], per_key); }
        catch (const LoadError& e) { mesh_err[i].code = e.code; mesh_err[i].msg = e.what(); }
        catch (const std::exception& e) { mesh_err[i].code = TRAY_E_PARSE; mesh_err[i].msg = e.what(); }
    });
    for (auto& e : mesh_err)   // the first failing request in scene order, as a sequential load would report it
        if (e.code != TRAY_OK) fail(e.code, e.msg);
}

// ------------------------------------------------------------------ objects (scene.rs:515-654)

static void load_geometry(TrayHostScene& s, const Json& e, const std::string& base, bool sampleable, HostInstance& inst) {
    const std::string& ty = need_str(e, "type", "A type is required for geometry", "Geometry type must be a string");
    if (ty == "sphere") {
        inst.geom_type = TRAY_GEOM_SPHERE;
        inst.geom_params[0] = need_f32(e, "radius", "A radius is required for a sphere", "radius must be a number");
    } else if (ty == "disk") {
        inst.geom_type = TRAY_GEOM_DISK;
        inst.geom_params[0] = need_f32(e, "radius", "A radius is required for a disk", "radius must be a number");
        inst.geom_params[1] = need_f32(e, "inner_radius", "An inner radius is required for a disk", "inner radius must be a number");
    } else if (ty == "plane" && !sampleable) {   // scene.rs:598-600
        inst.geom_type = TRAY_GEOM_RECT;
        inst.geom_params[0] = 2.0f; inst.geom_params[1] = 2.0f;
    } else if (ty == "rectangle") {
        inst.geom_type = TRAY_GEOM_RECT;
        inst.geom_params[0] = need_f32(e, "width", "A width is required for a rectangle", "width must be a number");
        inst.geom_params[1] = need_f32(e, "height", "A height is required for a rectangle", "height must be a number");
    } else if (ty == "mesh" && !sampleable) {
        const std::string& file = need_str(e, "file", "An OBJ file is required for meshes", "OBJ filename must be a string");
        const std::string& model = need_str(e, "model", "A model name is required for geometry", "Model name type must be a string");
        inst.geom_type = TRAY_GEOM_MESH;
        inst.mesh_id = get_mesh(s, join_path(base, file), model);
    } else if (ty == "animated_mesh" && !sampleable) {
        // geometry/animated_mesh.rs:14-27 documents this entry; scene.rs:588-628 has no branch that reads it, so the
        // messages are this loader's
        const std::string& model = need_str(e, "model", "A model name is required for geometry", "Model name type must be a string");
        const Json& kfs = need(e, "keyframes", "Keyframes are required for an animated mesh");
        if (!kfs.is_array()) fail(TRAY_E_PARSE, "The keyframes of an animated mesh must be an array");
        std::vector<std::string> files;
        std::vector<float> times;
        for (const Json& k : kfs.arr) {
            files.push_back(join_path(base, need_str(k, "file", "An OBJ file is required for every keyframe of an animated mesh", "OBJ filename must be a string")));
            times.push_back(need_f32(k, "time", "A time is required for every keyframe of an animated mesh", "keyframe time must be a number"));
        }
        if (files.size() < 2) fail(TRAY_E_INVALID, "An animated mesh needs at least two keyframes (AnimatedMesh::new reads times[1], animated_mesh.rs:125)");
        for (size_t k = 1; k < times.size(); ++k)   // "It's assumed the meshes are sorted in ascending time" (animated_mesh.rs:112): binary_search_by needs it
            if (!(times[k] > times[k - 1])) fail(TRAY_E_INVALID, "The keyframes of an animated mesh must be in ascending time");
        inst.geom_type = TRAY_GEOM_ANIMATED_MESH;
        inst.mesh_id = get_animated_mesh(s, files, times, model);
    } else if (sampleable) {
        fail(TRAY_E_INVALID, "Geometry of type '" + ty + "' is not sampleable and can't be used for area light geometry");
    } else {
        fail(TRAY_E_PARSE, "Unrecognized geometry type '" + ty + "'");
    }
}

static uint32_t find_material(const TrayHostScene& s, const Json& o) {
    const std::string& mat = need_str(o, "material", "A material is required for an object", "Object material name must be a string");
    auto it = s.material_names.find(mat);
    if (it == s.material_names.end()) fail(TRAY_E_INVALID, "Material " + mat + " was not found in the material list");
    return it->second;
}

static void load_objects(TrayHostScene& s, const Json& e, const std::string& base, std::vector<HostInstance>& out) {
    if (!e.is_array()) fail(TRAY_E_PARSE, "The objects must be an array of objects used");
    for (auto& o : e.arr) {
        const std::string name = need_str(o, "name", "A name is required for an object", "Object name must be a string");
        const std::string& ty = need_str(o, "type", "A type is required for an object", "Object type must be a string");
        AnimXform xf = load_object_transform(o, name);
        if (ty == "emitter") {
            const std::string& ety = need_str(o, "emitter", "An emitter type is required for emitters", "Emitter type must be a string");
            HostInstance inst;
            inst.name = name;
            inst.xf = xf;
            if (!load_animated_color(need(o, "emission", "An emission color is required for emitters"), inst.emission))
                fail(TRAY_E_PARSE, "Emitter emission must be a color");
            if (ety == "point") {
                inst.kind = TRAY_INST_POINT_EMITTER;
            } else if (ety == "area") {
                inst.kind = TRAY_INST_AREA_EMITTER;
                inst.material_id = find_material(s, o);
                load_geometry(s, need(o, "geometry", "Geometry is required for area lights"), base, true, inst);
            } else {
                fail(TRAY_E_PARSE, "Invalid emitter type specified: " + ety);
            }
            out.push_back(std::move(inst));
        } else if (ty == "receiver") {
            HostInstance inst;
            inst.name = name;
            inst.xf = xf;
            inst.kind = TRAY_INST_RECEIVER;
            inst.material_id = find_material(s, o);
            load_geometry(s, need(o, "geometry", "Geometry is required for receivers"), base, false, inst);
            out.push_back(std::move(inst));
        } else if (ty == "group") {
            std::vector<HostInstance> group;
            load_objects(s, need(o, "objects", "A group must specify an array of objects in the group"), base, group);
            for (auto& gi : group) {   // `transform.clone() * t`: child levels first, then the group's (animated_transform.rs:78-87)
                for (auto& l : xf.levels) gi.xf.levels.push_back(l);
                out.push_back(std::move(gi));
            }
        } else {
            fail(TRAY_E_PARSE, "Error parsing object '" + name + "': unrecognized type '" + ty + "'");
        }
    }
}

// ------------------------------------------------------------------ load + flatten

static TrayHostScene* load_scene(const std::string& text, const std::string& base) {
    std::unique_ptr<TrayHostScene> s(new TrayHostScene());
    Json data;
    try {
        data = Json::parse(text);
    } catch (const JsonError& e) {
        fail(TRAY_E_PARSE, e.what());
    }
    if (!data.is_object()) fail(TRAY_E_PARSE, "Expected a root JSON object. See example scenes");
    load_film(need(data, "film", "The scene must specify a film to write to"), *s);
    if (const Json* cams = data.get("cameras")) {   // scene.rs:231-247
        if (!cams->is_array()) fail(TRAY_E_PARSE, "cameras listing must be an array of cameras");
        for (auto& c : cams->arr) s->cameras.push_back(load_camera(c));
        std::stable_sort(s->cameras.begin(), s->cameras.end(), [](const HostCamera& a, const HostCamera& b) { return a.active_at < b.active_at; });
    } else {
        s->cameras.push_back(load_camera(need(data, "camera", "Error: A camera is required!")));
    }
    if (s->cameras.empty()) fail(TRAY_E_INVALID, "Error: A camera is required!");
    const Json& integ = need(data, "integrator", "The scene must specify the integrator to render with");
    const std::string& ity = need_str(integ, "type", "Integrator must specify a type", "Integrator type must be a string");
    if (ity == "pathtracer") {
        s->min_depth = (uint32_t)need_u64(integ, "min_depth", "The integrator must specify the minimum ray depth", "min_depth must be a number");
        s->max_depth = (uint32_t)need_u64(integ, "max_depth", "The integrator must specify the maximum ray depth", "max_depth must be a number");
        if (s->max_depth > 15) fail(TRAY_E_UNSUPPORTED, "pathtracer max_depth > 15 is not supported by the device sampler (sample arrays hold <= 16 entries)");
    } else if (ity == "normals_debug") {   // integrator/normals_debug.rs: no parameters
        s->integrator = TRAY_INTEGRATOR_NORMALS_DEBUG;
    } else if (ity == "whitted") {   // scene.rs:306-309: Whitted::new(min_depth) -- the recursion limit is read from the "min_depth" key
        s->integrator = TRAY_INTEGRATOR_WHITTED;
        s->max_depth = (uint32_t)need_u64(integ, "min_depth", "The integrator must specify the minimum ray depth", "min_depth must be a number");
        if (s->max_depth > 16) fail(TRAY_E_UNSUPPORTED, "whitted recursion depth > 16 is not supported (the device keeps one frame per level)");
    } else {
        fail(TRAY_E_PARSE, "Unrecognized integrator type '" + ity + "'");
    }
    if (const Json* tex = data.get("textures")) load_textures(*s, *tex, base);
    load_materials(*s, need(data, "materials", "An array of materials is required"), base);
    load_objects(*s, need(data, "objects", "The scene must specify a list of objects"), base, s->instances);
    build_meshes(*s);
    if (s->instances.empty()) fail(TRAY_E_INVALID, "Aborting: the scene does not have any objects!");
    return s.release();
}

static void color_at(const std::vector<ColorKey>& keys, float time, float out[4]) {   // animated_color.rs:52-78
    if (keys.empty()) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
    if (keys.size() == 1) { std::memcpy(out, keys[0].c, sizeof(float) * 4); return; }
    const ColorKey* first = nullptr;
    const ColorKey* second = nullptr;
    size_t i = 0;
    while (i < keys.size() && keys[i].time < time) { first = &keys[i]; ++i; }
    if (i < keys.size()) second = &keys[i];
    if (!first) { std::memcpy(out, keys.front().c, sizeof(float) * 4); return; }
    if (!second) { std::memcpy(out, keys.back().c, sizeof(float) * 4); return; }
    float t = (time - first->time) / (second->time - first->time);
    for (int k = 0; k < 4; ++k) out[k] = lerpf(t, first->c[k], second->c[k]);
}

static BBox geom_bounds(const TrayHostScene& s, const HostInstance& inst) {
    switch (inst.geom_type) {
        case TRAY_GEOM_SPHERE: { float r = inst.geom_params[0]; return BBox(V3(-r, -r, -r), V3(r, r, r)); }   // sphere.rs:84-89
        case TRAY_GEOM_DISK: { float r = inst.geom_params[0]; return BBox(V3(-r, -r, -0.1f), V3(r, r, 0.1f)); }   // disk.rs:78-82
        case TRAY_GEOM_RECT: {   // rectangle.rs:66-72
            float hw = inst.geom_params[0] / 2.0f, hh = inst.geom_params[1] / 2.0f;
            return BBox(V3(-hw, -hh, 0.0f), V3(hw, hh, 0.0f));
        }
        case TRAY_GEOM_ANIMATED_MESH:   // AnimatedMesh::bounds = self.bvh.bounds (animated_mesh.rs:136-138): the tree of times[0] .. times[1], whatever the frame (quirk Q13)
        case TRAY_GEOM_MESH: {   // BVH::bounds = root node bounds (bvh.rs:270-274)
            const TrayBvhNode& n = s.meshes[inst.mesh_id].bvh.nodes[0];
            return BBox(V3(n.bmin[0], n.bmin[1], n.bmin[2]), V3(n.bmax[0], n.bmax[1], n.bmax[2]));
        }
        default: return BBox(V3(0, 0, 0), V3(0, 0, 0));   // point light: BBox::singular(origin), emitter.rs:156
    }
}

static void flatten(TrayHostScene& s, uint32_t frame) {
    if (frame >= s.frames && !(s.frames == 1 && frame == 0)) {
        // the reference does not check this; frames past `frames` simply extrapolate time. Keep that.
    }
    // Exec::render frame timing (multithreaded.rs:57-60)
    float time_step = s.scene_time / (float)s.frames;
    float start = (float)frame * time_step;
    float end = ((float)frame + 1.0f) * time_step;
    // Scene::update_frame camera choice (scene.rs:153-170), stateless form
    size_t cam_idx = 0, n_active = 0;
    for (auto& c : s.cameras) { if (c.active_at <= frame) ++n_active; else break; }
    if (n_active == 0) fail(TRAY_E_INVALID, "no camera is active at the requested frame");
    cam_idx = n_active - 1;
    const HostCamera& cam = s.cameras[cam_idx];
    float shutter_open = start;
    float shutter_close = start + cam.shutter_size * (end - start);   // camera.rs:127-129

    TrayFlatScene& f = s.flat;
    std::memset(&f, 0, sizeof f);
    f.abi_version = TRAY_ABI_VERSION;
    f.frame = frame;
    f.film = s.film;
    f.integrator = s.integrator;
    f.min_depth = s.min_depth;
    f.max_depth = s.max_depth;

    s.f_levels.clear(); s.f_keyframes.clear(); s.f_knots.clear();
    auto push_levels = [&](const AnimXform& a, uint32_t& first, uint32_t& count) {
        first = (uint32_t)s.f_levels.size();
        count = (uint32_t)a.levels.size();
        for (auto& l : a.levels) {
            TrayXformLevel tl{};
            tl.kf_first = (uint32_t)s.f_keyframes.size(); tl.kf_count = (uint32_t)l.kfs.size();
            tl.knot_first = (uint32_t)s.f_knots.size(); tl.knot_count = (uint32_t)l.knots.size();
            tl.degree = l.degree;
            float t_eval;
            if (AnimXform::level_const_over(l, shutter_open, shutter_close, t_eval)) {
                Xform kt = AnimXform::level_transform(l, t_eval);
                tl.is_const = 1;
                std::memcpy(tl.mat, kt.mat.m, sizeof tl.mat);
                std::memcpy(tl.inv, kt.inv.m, sizeof tl.inv);
            }
            for (auto& k : l.kfs) {
                TrayKeyframe tk{};
                for (int i = 0; i < 3; ++i) { tk.translation[i] = k.translation[i]; tk.rotation[i] = k.rotation.v[i]; tk.scaling[i] = k.scaling[i]; }
                tk.rotation[3] = k.rotation.w;
                s.f_keyframes.push_back(tk);
            }
            for (float kn : l.knots) s.f_knots.push_back(kn);
            s.f_levels.push_back(tl);
        }
    };

    // Camera (camera.rs:64-91)
    {
        TrayCamera& c = f.camera;
        float aspect = (float)s.width / (float)s.height;
        float screen[4];
        if (aspect > 1.0f) { screen[0] = -aspect; screen[1] = aspect; screen[2] = -1.0f; screen[3] = 1.0f; }
        else { screen[0] = -1.0f; screen[1] = 1.0f; screen[2] = -1.0f / aspect; screen[3] = 1.0f / aspect; }
        Xform screen_raster = Xform::scale(V3((float)s.width, (float)s.height, 1.0f))
                            * Xform::scale(V3(1.0f / (screen[1] - screen[0]), 1.0f / (screen[2] - screen[3]), 1.0f))
                            * Xform::translate(V3(-screen[0], -screen[3], 0.0f));
        Xform raster_screen = screen_raster.inverse();
        const float far = 1.0f, near = 1000.0f;   // sic (camera.rs:77-78)
        M4 proj_div = M4::identity();
        proj_div.at(2, 2) = far / (far - near);
        proj_div.at(2, 3) = -far * near / (far - near);
        proj_div.at(3, 2) = 1.0f;
        proj_div.at(3, 3) = 0.0f;
        Xform proj_div_inv = Xform::from_mat(proj_div).inverse();
        Xform r2c = proj_div_inv * raster_screen;   // evaluated per ray in the reference (camera.rs:152)
        std::memcpy(c.raster_to_cam, r2c.mat.m, sizeof c.raster_to_cam);
        float fov = cam.fov;
        if (cam.animated_fov) {   // Camera::update_frame (camera.rs:130-141): the spline at the middle of the frame
            size_t nk = cam.fov_knots.size();
            float t_mid = clampf((start + end) / 2.0f, cam.fov_knots[cam.fov_degree], cam.fov_knots[nk - 1 - cam.fov_degree]);
            fov = bspline_point(cam.fovs.data(), cam.fov_knots.data(), nk, cam.fov_degree, t_mid,
                                [](float a, float b, float t) { return a * (1.0f - t) + b * t; });   // bspline's blanket Interpolate
        }
        float tan_fov = std::tan(to_radians(fov) / 2.0f);
        c.scaling[0] = tan_fov; c.scaling[1] = tan_fov; c.scaling[2] = 1.0f;
        c.shutter_open = shutter_open; c.shutter_close = shutter_close;
        c.animated = cam.cam_world.varies_over(shutter_open, shutter_close) ? 1u : 0u;
        Xform cw = cam.cam_world.transform(shutter_open);
        std::memcpy(c.cam_world, cw.mat.m, sizeof c.cam_world);
        push_levels(cam.cam_world, c.xf_first, c.xf_count);
    }

    // Instances, lights, top-level BVH over the shutter interval (scene.rs:171-175; receiver.rs:55-57)
    s.f_instances.clear(); s.f_lights.clear(); s.f_color_keys.clear();
    bool any_animated = f.camera.animated != 0;
    uint32_t n_moving = 0;
    std::vector<BBox> inst_bounds;
    for (size_t i = 0; i < s.instances.size(); ++i) {
        const HostInstance& hi = s.instances[i];
        TrayInstance ti{};
        ti.kind = hi.kind; ti.geom_type = hi.geom_type; ti.mesh_id = hi.mesh_id; ti.material_id = hi.material_id;
        std::memcpy(ti.geom_params, hi.geom_params, sizeof ti.geom_params);
        ti.light_index = 0xffffffffu;
        if (hi.kind != TRAY_INST_RECEIVER) {
            color_at(hi.emission, shutter_open, ti.emission);
            if (hi.emission.size() > 1 && shutter_open != shutter_close) {   // Emitter::radiance(.., time) per ray (emitter.rs:139-141)
                ti.emis_first = (uint32_t)s.f_color_keys.size(); ti.emis_count = (uint32_t)hi.emission.size();
                for (auto& k : hi.emission) {
                    TrayColorKey ck{};
                    std::memcpy(ck.color, k.c, sizeof ck.color);
                    ck.time = k.time;
                    s.f_color_keys.push_back(ck);
                }
                any_animated = true;
            }
            ti.light_index = (uint32_t)s.f_lights.size();
            s.f_lights.push_back((uint32_t)i);
        }
        // with a closed shutter every ray of the frame has time == shutter_open: the stack is evaluated once, here
        ti.animated = hi.xf.varies_over(shutter_open, shutter_close) ? 1u : 0u;
        ti.moving_slot = 0xffffffffu;
        if (ti.animated) { any_animated = true; ti.moving_slot = n_moving++; }
        if (hi.geom_type == TRAY_GEOM_ANIMATED_MESH) any_animated = true;   // its vertices are functions of ray.time (animated_mesh.rs:160-172), open shutter or not
        Xform t = hi.xf.transform(shutter_open);
        std::memcpy(ti.mat, t.mat.m, sizeof ti.mat);
        std::memcpy(ti.inv, t.inv.m, sizeof ti.inv);
        push_levels(hi.xf, ti.xf_first, ti.xf_count);
        s.f_instances.push_back(ti);
        // Receiver/Emitter::bounds (receiver.rs:55-57, emitter.rs:151-160): swept over the shutter interval
        inst_bounds.push_back(hi.xf.animation_bounds(geom_bounds(s, hi), shutter_open, shutter_close));
    }
    f.animated = any_animated ? 1u : 0u;
    BvhBuild top = build_bvh(inst_bounds, 4);   // BVH::new(4, instances, ..) scene.rs:141
    s.f_top_nodes = top.nodes;
    s.f_top_order = top.ordered;

    // Meshes: nothing in them depends on the frame (the loader's meshes are immutable), so the arrays of the first flatten serve every later one --
    // copying 3.1 M triangles and their trees anew was 0.1 s per frame of the tr15 stand-in, most of what a frame update cost beside its kernels
    bool any_keys = s.f_any_keys;
    if (!s.f_meshes_done) {
        // (a retry after a flatten that threw inside this loop -- bad_alloc on a large scene; the handle stays usable -- starts from empty arrays,
        // not on top of the partly filled ones: ADVICE round 5)
        s.f_meshes.clear(); s.f_mesh_nodes.clear(); s.f_verts.clear(); s.f_attrs.clear(); s.f_mesh_keys.clear(); s.f_key_times.clear();
        any_keys = false;
    }
    if (!s.f_meshes_done)
    for (auto& m : s.meshes) {
        TrayMesh tm{};
        const size_t n_keys = std::max<size_t>(m.key_times.size(), 1);
        TrayMeshKeys mk{(uint32_t)n_keys, (uint32_t)s.f_key_times.size()};
        s.f_key_times.insert(s.f_key_times.end(), m.key_times.begin(), m.key_times.end());
        s.f_mesh_keys.push_back(mk);
        any_keys = any_keys || n_keys > 1;
        tm.node_offset = (uint32_t)s.f_mesh_nodes.size(); tm.node_count = (uint32_t)m.bvh.nodes.size();
        tm.tri_offset = (uint32_t)s.f_verts.size(); tm.tri_count = (uint32_t)(m.verts.size() / n_keys);
        s.f_mesh_nodes.insert(s.f_mesh_nodes.end(), m.bvh.nodes.begin(), m.bvh.nodes.end());
        s.f_verts.insert(s.f_verts.end(), m.verts.begin(), m.verts.end());
        s.f_attrs.insert(s.f_attrs.end(), m.attrs.begin(), m.attrs.end());
        s.f_meshes.push_back(tm);
    }
    s.f_meshes_done = true; s.f_any_keys = any_keys;

    f.n_instances = (uint32_t)s.f_instances.size(); f.instances = s.f_instances.data();
    f.n_top_nodes = (uint32_t)s.f_top_nodes.size(); f.top_nodes = s.f_top_nodes.data();
    f.n_top_order = (uint32_t)s.f_top_order.size(); f.top_order = s.f_top_order.data();
    f.n_meshes = (uint32_t)s.f_meshes.size(); f.meshes = s.f_meshes.data();
    f.n_mesh_nodes = (uint32_t)s.f_mesh_nodes.size(); f.mesh_nodes = s.f_mesh_nodes.data();
    f.n_tris = (uint32_t)s.f_verts.size(); f.tri_verts = s.f_verts.data(); f.tri_attrs = s.f_attrs.data();
    f.n_mesh_keys = any_keys ? (uint32_t)s.f_mesh_keys.size() : 0u; f.mesh_keys = any_keys ? s.f_mesh_keys.data() : nullptr;
    f.n_key_times = (uint32_t)s.f_key_times.size(); f.key_times = s.f_key_times.data();
    f.n_materials = (uint32_t)s.materials.size(); f.materials = s.materials.data();
    f.n_textures = (uint32_t)s.textures.size(); f.textures = s.textures.data();
    f.n_tex_frames = (uint32_t)s.tex_frames.size(); f.tex_frames = s.tex_frames.data();
    f.n_tex_bytes = s.tex_data.size(); f.tex_data = s.tex_data.data();
    f.n_merl = (uint32_t)s.merl_tables.size(); f.merl_tables = s.merl_tables.data();
    f.n_merl_floats = s.merl_data.size(); f.merl_data = s.merl_data.data();
    f.n_lights = (uint32_t)s.f_lights.size(); f.lights = s.f_lights.data();
    f.n_xf_levels = (uint32_t)s.f_levels.size(); f.xf_levels = s.f_levels.data();
    f.n_keyframes = (uint32_t)s.f_keyframes.size(); f.keyframes = s.f_keyframes.data();
    f.n_knots = (uint32_t)s.f_knots.size(); f.knots = s.f_knots.data();
    f.n_color_keys = (uint32_t)s.f_color_keys.size(); f.color_keys = s.f_color_keys.data();
}

template <class F>
static int guarded(F&& fn) {
    try {
        fn();
        return TRAY_OK;
    } catch (const LoadError& e) {
        set_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_error("out of memory");
        return TRAY_E_NOMEM;
    } catch (const std::exception& e) {
        set_error(e.what());
        return TRAY_E_INVALID;
    }
}

}  // namespace trayh

extern "C" {

int tray_scene_load_string(const char* json, const char* base_dir, TrayHostScene** out) {
    if (!json || !out) { set_error("tray_scene_load_string: null argument"); return TRAY_E_INVALID; }
    *out = nullptr;
    return guarded([&] { *out = load_scene(json, base_dir ? base_dir : ""); });
}

int tray_scene_load_file(const char* path, TrayHostScene** out) {
    if (!path || !out) { set_error("tray_scene_load_file: null argument"); return TRAY_E_INVALID; }
    *out = nullptr;
    return guarded([&] {
        std::ifstream f(path, std::ios::binary);
        if (!f) fail(TRAY_E_IO, std::string("Failed to open scene file: ") + path);
        std::stringstream ss;
        ss << f.rdbuf();
        std::string p(path);
        size_t slash = p.find_last_of('/');
        std::string base = slash == std::string::npos ? std::string("") : p.substr(0, slash);   // Path::parent, scene.rs:116-119
        *out = load_scene(ss.str(), base);
    });
}

int tray_host_scene_info(const TrayHostScene* s, TraySceneInfo* info) {
    if (!s || !info) { set_error("tray_host_scene_info: null argument"); return TRAY_E_INVALID; }
    info->width = s->width; info->height = s->height; info->spp = s->spp;
    info->frames = s->frames; info->start_frame = s->start_frame; info->end_frame = s->end_frame;
    info->scene_time = s->scene_time;
    info->n_instances = (uint32_t)s->instances.size();
    uint32_t nl = 0;
    for (auto& i : s->instances) if (i.kind != TRAY_INST_RECEIVER) ++nl;
    info->n_lights = nl;
    info->n_meshes = (uint32_t)s->meshes.size();
    uint32_t nt = 0;
    for (auto& m : s->meshes) nt += (uint32_t)m.verts.size();
    info->n_tris = nt;
    return TRAY_OK;
}

int tray_host_scene_flatten(TrayHostScene* s, uint32_t frame, const TrayFlatScene** out) {
    if (!s || !out) { set_error("tray_host_scene_flatten: null argument"); return TRAY_E_INVALID; }
    *out = nullptr;
    int rc = guarded([&] { flatten(*s, frame); });
    if (rc == TRAY_OK) {   // the loader's own output goes through the check tray_scene_create applies to any caller's scene
        const std::string bad = tray::validate_flat_scene(&s->flat, !s->f_trees_checked);   // (the mesh arrays are the first flatten's: walked once)
        s->f_trees_checked = s->f_trees_checked || bad.empty();
        if (!bad.empty()) { set_error("internal error: the flattened scene is inconsistent: " + bad); return TRAY_E_INVALID; }
        *out = &s->flat;
    }
    return rc;
}

void tray_host_scene_free(TrayHostScene* s) { delete s; }

}  // extern "C"
