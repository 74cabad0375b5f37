"""ctypes binding of libtrayhip.so (include/trayhip.h). Fails loudly when the library is missing:
there is no CPU fallback in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRAYHIP_LIB") or os.path.join(_HERE, "libtrayhip.so")   # TRAYHIP_LIB: A/B builds of the same library

TRAY_OK = 0
TRAY_E_INVALID, TRAY_E_IO, TRAY_E_PARSE, TRAY_E_UNSUPPORTED, TRAY_E_DEVICE, TRAY_E_NOMEM = -1, -2, -3, -4, -5, -6
GEOM_SPHERE, GEOM_DISK, GEOM_RECT, GEOM_MESH, GEOM_NONE = 0, 1, 2, 3, 4
INST_RECEIVER, INST_AREA_EMITTER, INST_POINT_EMITTER = 0, 1, 2
MAT_MATTE, MAT_PLASTIC, MAT_METAL, MAT_GLASS, MAT_ROUGH_GLASS, MAT_SPECULAR_METAL, MAT_MERL = range(7)
FILTER_TABLE_SIZE = 16


class TrayError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"trayhip error {code}: {message}")
        self.code = code
        self.message = message


class TrayBvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_uint32), ("count", C.c_uint16),
                ("axis", C.c_uint8), ("pad", C.c_uint8)]


class TrayTriVerts(C.Structure):
    _fields_ = [("pa", C.c_float * 3), ("tri_id", C.c_uint32), ("pb", C.c_float * 3), ("pad0", C.c_uint32),
                ("pc", C.c_float * 3), ("pad1", C.c_uint32)]


class TrayTriAttrs(C.Structure):
    _fields_ = [("na", C.c_float * 3), ("nb", C.c_float * 3), ("nc", C.c_float * 3), ("ta", C.c_float * 2),
                ("tb", C.c_float * 2), ("tc", C.c_float * 2), ("pad", C.c_float)]


class TrayMesh(C.Structure):
    _fields_ = [("node_offset", C.c_uint32), ("node_count", C.c_uint32), ("tri_offset", C.c_uint32), ("tri_count", C.c_uint32)]


class TrayInstance(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("geom_type", C.c_uint32), ("mesh_id", C.c_uint32), ("material_id", C.c_uint32),
                ("geom_params", C.c_float * 4), ("emission", C.c_float * 4), ("mat", C.c_float * 16), ("inv", C.c_float * 16),
                ("light_index", C.c_uint32), ("xf_first", C.c_uint32), ("xf_count", C.c_uint32), ("animated", C.c_uint32),
                ("emis_first", C.c_uint32), ("emis_count", C.c_uint32), ("moving_slot", C.c_uint32), ("pad", C.c_uint32)]


class TrayKeyframe(C.Structure):
    _fields_ = [("translation", C.c_float * 3), ("rotation", C.c_float * 4), ("scaling", C.c_float * 3)]


class TrayXformLevel(C.Structure):
    _fields_ = [("kf_first", C.c_uint32), ("kf_count", C.c_uint32), ("knot_first", C.c_uint32), ("knot_count", C.c_uint32),
                ("degree", C.c_uint32), ("is_const", C.c_uint32), ("pad", C.c_uint32 * 2), ("mat", C.c_float * 16), ("inv", C.c_float * 16)]


class TrayColorKey(C.Structure):
    _fields_ = [("color", C.c_float * 4), ("time", C.c_float), ("pad", C.c_float * 3)]


class TrayMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("table", C.c_uint32), ("f0", C.c_float), ("f1", C.c_float), ("c0", C.c_float * 4),
                ("c1", C.c_float * 4), ("tex_c0", C.c_uint32), ("tex_c1", C.c_uint32), ("tex_f0", C.c_uint32), ("tex_f1", C.c_uint32),
                ("microfacet", C.c_uint32), ("pad", C.c_uint32 * 3)]


class TrayTexture(C.Structure):
    _fields_ = [("first_frame", C.c_uint32), ("n_frames", C.c_uint32)]


class TrayTexFrame(C.Structure):
    _fields_ = [("time", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32), ("pad", C.c_uint32), ("offset", C.c_uint64)]


class TrayMerlTable(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("n_theta_h", C.c_uint32), ("n_theta_d", C.c_uint32), ("n_phi_d", C.c_uint32),
                ("pad", C.c_uint32)]


class TrayCamera(C.Structure):
    _fields_ = [("raster_to_cam", C.c_float * 16), ("scaling", C.c_float * 3), ("shutter_open", C.c_float),
                ("shutter_close", C.c_float), ("cam_world", C.c_float * 16), ("animated", C.c_uint32), ("xf_first", C.c_uint32),
                ("xf_count", C.c_uint32)]


class TrayFilm(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("filter_w", C.c_float), ("filter_h", C.c_float),
                ("inv_w", C.c_float), ("inv_h", C.c_float), ("filter_pixel_w", C.c_int32), ("filter_pixel_h", C.c_int32),
                ("table", C.c_float * (FILTER_TABLE_SIZE * FILTER_TABLE_SIZE)),
                ("table_x", C.c_float * FILTER_TABLE_SIZE), ("table_y", C.c_float * FILTER_TABLE_SIZE), ("separable", C.c_uint32)]


def _P(t):
    return C.POINTER(t)


class TrayMeshKeys(C.Structure):
    _fields_ = [("n_keys", C.c_uint32), ("time_first", C.c_uint32)]


class TrayFlatScene(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("frame", C.c_uint32), ("film", TrayFilm), ("camera", TrayCamera),
        ("min_depth", C.c_uint32), ("max_depth", C.c_uint32),
        ("n_instances", C.c_uint32), ("instances", _P(TrayInstance)),
        ("n_top_nodes", C.c_uint32), ("top_nodes", _P(TrayBvhNode)),
        ("n_top_order", C.c_uint32), ("top_order", _P(C.c_uint32)),
        ("n_meshes", C.c_uint32), ("meshes", _P(TrayMesh)),
        ("n_mesh_nodes", C.c_uint32), ("mesh_nodes", _P(TrayBvhNode)),
        ("n_tris", C.c_uint32), ("tri_verts", _P(TrayTriVerts)), ("tri_attrs", _P(TrayTriAttrs)),
        ("n_materials", C.c_uint32), ("materials", _P(TrayMaterial)),
        ("n_merl", C.c_uint32), ("merl_tables", _P(TrayMerlTable)),
        ("n_merl_floats", C.c_uint64), ("merl_data", _P(C.c_float)),
        ("n_lights", C.c_uint32), ("lights", _P(C.c_uint32)),
        ("n_xf_levels", C.c_uint32), ("xf_levels", _P(TrayXformLevel)),
        ("n_keyframes", C.c_uint32), ("keyframes", _P(TrayKeyframe)),
        ("n_knots", C.c_uint32), ("knots", _P(C.c_float)),
        ("n_color_keys", C.c_uint32), ("color_keys", _P(TrayColorKey)), ("animated", C.c_uint32),
        ("integrator", C.c_uint32), ("n_textures", C.c_uint32), ("textures", _P(TrayTexture)), ("n_tex_frames", C.c_uint32), ("tex_frames", _P(TrayTexFrame)),
        ("n_tex_bytes", C.c_uint64), ("tex_data", _P(C.c_uint8)),
        ("n_mesh_keys", C.c_uint32), ("mesh_keys", _P(TrayMeshKeys)), ("n_key_times", C.c_uint32), ("key_times", _P(C.c_float)),
    ]


class TraySceneInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("frames", C.c_uint32),
                ("start_frame", C.c_uint32), ("end_frame", C.c_uint32), ("scene_time", C.c_float), ("n_instances", C.c_uint32),
                ("n_lights", C.c_uint32), ("n_meshes", C.c_uint32), ("n_tris", C.c_uint32)]


class TrayKernelTiming(C.Structure):
    _fields_ = [("render_ms", C.c_float), ("launches", C.c_uint32), ("samples", C.c_uint64), ("vertices", C.c_uint64),
                ("rays", C.c_uint64), ("retraced", C.c_uint64)]


class TrayScheduleInfo(C.Structure):
    _fields_ = [("wavefront", C.c_uint32), ("launched_wavefront", C.c_uint32), ("pool_slots", C.c_uint32), ("chunks", C.c_uint32),
                ("views", C.c_uint32), ("slices", C.c_uint32), ("n_moving", C.c_uint32), ("tile_workgroups", C.c_uint32),
                ("pool_bytes", C.c_uint64), ("schedule_bytes", C.c_uint64), ("xf_cache_bytes", C.c_uint64),
                ("transform_table", C.c_uint32), ("binned_stages", C.c_uint32), ("xf_table_bytes", C.c_uint64)]


class TrayRay(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("min_t", C.c_float), ("max_t", C.c_float), ("time", C.c_float)]


class TrayHit(C.Structure):
    _fields_ = [("t", C.c_float), ("inst", C.c_uint32), ("prim", C.c_uint32), ("p", C.c_float * 3), ("n", C.c_float * 3),
                ("ng", C.c_float * 3), ("u", C.c_float), ("v", C.c_float), ("dp_du", C.c_float * 3), ("dp_dv", C.c_float * 3)]


# every symbol include/trayhip.h declares, with its signature
SYMBOLS = {
    "tray_scene_load_file": (C.c_int, [C.c_char_p, _P(C.c_void_p)]),
    "tray_scene_load_string": (C.c_int, [C.c_char_p, C.c_char_p, _P(C.c_void_p)]),
    "tray_host_scene_info": (C.c_int, [C.c_void_p, _P(TraySceneInfo)]),
    "tray_host_scene_flatten": (C.c_int, [C.c_void_p, C.c_uint32, _P(_P(TrayFlatScene))]),
    "tray_host_scene_free": (None, [C.c_void_p]),
    "tray_block_queue": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_uint32), C.c_uint32, _P(C.c_uint32)]),
    "tray_round_spp": (C.c_uint32, [C.c_uint32]),
    "tray_resolve_srgb8": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "tray_init": (C.c_int, [C.c_int]),
    "tray_device_count": (C.c_int, [_P(C.c_int)]),
    "tray_scene_create": (C.c_int, [_P(TrayFlatScene), _P(C.c_void_p)]),
    "tray_scene_update_frame": (C.c_int, [C.c_void_p, _P(TrayFlatScene)]),
    "tray_scene_destroy": (None, [C.c_void_p]),
    "tray_scene_set_sampler": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "tray_multi_set_sampler": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "tray_scene_set_wavefront": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "tray_multi_set_wavefront": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "tray_last_schedule": (C.c_int, [C.c_void_p, _P(TrayScheduleInfo)]),
    "tray_scene_set_transform_table": (C.c_int, [C.c_void_p, C.c_int]),
    "tray_multi_set_transform_table": (C.c_int, [C.c_void_p, C.c_int]),
    "tray_debug_transform_table": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_uint32)]),
    "tray_adaptive_step": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "tray_render_tiles_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "tray_render_shard_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "tray_shard_tiles": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_uint32), C.c_uint32, _P(C.c_uint32)]),
    "tray_multi_create": (C.c_int, [_P(TrayFlatScene), C.c_int, _P(C.c_int), _P(C.c_void_p)]),
    "tray_render_frame_multi": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]),
    "tray_multi_timing": (C.c_int, [C.c_void_p, _P(TrayKernelTiming), _P(C.c_float)]),
    "tray_multi_update_frame": (C.c_int, [C.c_void_p, _P(TrayFlatScene)]),
    "tray_multi_destroy": (None, [C.c_void_p]),
    "tray_render_tiles": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]),
    "tray_last_timing": (C.c_int, [C.c_void_p, _P(TrayKernelTiming)]),
    "tray_debug_intersect": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "tray_debug_sample_radiance": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]),
    "tray_debug_bsdf": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tray_last_error": (C.c_char_p, []),
    "tray_version": (C.c_char_p, []),
    "tray_abi_sizeof": (C.c_uint32, [C.c_char_p]),
}

_lib = None


def lib():
    """Load libtrayhip.so (built by __graft_entry__.build() / make -C tray_rust_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the HIP path)")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)   # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != TRAY_OK:
        msg = lib().tray_last_error()
        raise TrayError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
