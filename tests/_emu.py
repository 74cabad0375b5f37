"""Host emulation of the DEVICE source (tests/emu): the files hipcc compiles for gfx950, compiled by g++ behind a shim with
one-lane waves. Test infrastructure only; lets the CPU suite run per-lane device code (traversal kernels, debug kernels)
against the oracle without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np

from tray_rust_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
HIP_DIR = os.path.join(ROOT, "tray_rust_amd", "csrc", "hip")
_libs = {}


def _stale(so, deps):
    return not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps)


def _target(defines):
    so = os.path.join(EMU_DIR, "libtrayemu" + "".join("_" + d.lower() for d in defines).replace("_tr_", "_") + ".so")
    deps = [os.path.join(EMU_DIR, f) for f in ("emu_kernels.cpp", "hip_emu.h")]
    deps += [os.path.join(HIP_DIR, f) for f in os.listdir(HIP_DIR) if f.endswith((".h", ".hip"))]
    deps += [os.path.join(ROOT, "tray_rust_amd", "csrc", "host", "gates.hpp"), os.path.join(ROOT, "include", "trayhip.h")]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-attributes", "-shared", "-o", so,
           os.path.join(EMU_DIR, "emu_kernels.cpp")] + ["-D" + d for d in defines]
    return so, deps, cmd


def _norm(defines):
    return tuple(sorted(set(defines)))


def prebuild(define_sets):
    """compile the stale ones of several builds side by side (each takes ~25 s; the suite uses five)"""
    procs = []
    for defines in define_sets:
        so, deps, cmd = _target(_norm(defines))
        if _stale(so, deps):
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("host emulation build failed: " + " ".join(cmd))


def emu(defines=()):
    """libtrayemu.so, or a build of the same sources with extra macros (defines=("TR_...",))"""
    defines = _norm(defines)
    key = defines
    if key not in _libs:
        so, deps, cmd = _target(defines)
        if _stale(so, deps):
            subprocess.run(cmd, check=True)
        h = C.CDLL(so)
        FS = C.POINTER(L.TrayFlatScene)
        h.emu_debug_intersect.restype = C.c_int
        h.emu_debug_intersect.argtypes = [FS, C.c_uint32, C.c_void_p, C.c_void_p]
        h.emu_wf_trace.restype = C.c_int
        h.emu_wf_trace.argtypes = [FS, C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
        h.emu_debug_sample_radiance.restype = C.c_int
        h.emu_debug_sample_radiance.argtypes = [FS, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        h.emu_debug_bsdf.restype = C.c_int
        h.emu_debug_bsdf.argtypes = [FS, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        h.emu_render_tiles.restype = C.c_int
        h.emu_render_tiles.argtypes = [FS, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        h.emu_render_wavefront.restype = C.c_int
        h.emu_render_wavefront.argtypes = [FS, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        h.emu_render_sampler.restype = C.c_int
        h.emu_render_sampler.argtypes = [FS, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
        h.emu_retraced.restype = C.c_uint
        h.emu_wf_bin.restype = C.c_uint32
        h.emu_wf_bin.argtypes = [C.c_uint32] + [C.c_void_p] * 6 + [C.c_int]
        _libs[key] = h
    return _libs[key]


def retraced(defines=()):
    """rays the flat instance loop handed to the reference's two-level traversal since the last call (tied candidates)"""
    return int(emu(defines=defines).emu_retraced())


def debug_intersect(flat, rays, hit_dtype):
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 9)
    hits = np.zeros(len(rays), dtype=hit_dtype)
    assert emu().emu_debug_intersect(flat, len(rays), rays.ctypes.data, hits.ctypes.data) == 0
    return hits


def wf_trace(flat, rays, stage, lds_depth=0, blocks=3):
    """k_wf_trace_dyn, stage 0 / 1 / 2 = A / B / C; returns (hit, t, inst, prim)"""
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 9)
    n = len(rays)
    hit = np.zeros(n, np.uint32); t = np.zeros(n, np.float32); inst = np.zeros(n, np.uint32); prim = np.zeros(n, np.uint32)
    b1 = np.zeros(n, np.float32); b2 = np.zeros(n, np.float32)
    rc = emu().emu_wf_trace(flat, stage, n, rays.ctypes.data, lds_depth, blocks, hit.ctypes.data, t.ctypes.data,
                                 inst.ctypes.data, prim.ctypes.data, b1.ctypes.data, b2.ctypes.data)
    assert rc == 0, rc
    return hit.astype(bool), t, inst, prim


def sample_radiance(flat, px, py, si, spp, seed, defines=()):
    px = np.ascontiguousarray(px, np.uint32); py = np.ascontiguousarray(py, np.uint32); si = np.ascontiguousarray(si, np.uint32)
    out = np.zeros((len(px), 8), np.float32)
    assert emu(defines=defines).emu_debug_sample_radiance(flat, len(px), px.ctypes.data, py.ctypes.data, si.ctypes.data, spp, seed, out.ctypes.data) == 0
    return out


def bsdf(flat, material_id, flags_sel, dirs, u3):
    dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 6); u3 = np.ascontiguousarray(u3, np.float32).reshape(-1, 3)
    out = np.zeros((len(dirs), 12), np.float32)
    assert emu().emu_debug_bsdf(flat, material_id, flags_sel, len(dirs), dirs.ctypes.data, u3.ctypes.data, out.ctypes.data) == 0
    return out


def render_tiles(flat, tiles_xy, spp, seed, blocks=1, coop=-1, film_rows=-1, defines=(), shard=(0, 0, 1)):
    """k_path_tiles over the given tiles as a SIMT emulation (fibers); returns (rgbw image, (samples, vertices, rays, feat))"""
    fs = flat.contents
    tiles_xy = np.ascontiguousarray(tiles_xy, np.uint32).reshape(-1, 2)
    img = np.zeros((fs.film.height, fs.film.width, 4), np.float32)
    stats = np.zeros(4, np.uint64)
    rc = emu(defines=defines).emu_render_tiles(flat, tiles_xy.ctypes.data, len(tiles_xy), spp, seed, img.ctypes.data, blocks, coop, film_rows, stats.ctypes.data, *shard)
    assert rc == 0, f"emu_render_tiles: {rc}"
    return img, tuple(int(x) for x in stats)


def render_sampler(flat, tiles_xy, kind, min_spp=1, max_spp=1, seed=1, batch_tiles=0):
    """launch_sampler's rounds of k_sampler_pass / k_sampler_decide over the given tiles; returns (rgbw image, (samples, vertices, rays))"""
    fs = flat.contents
    tiles_xy = np.ascontiguousarray(tiles_xy, np.uint32).reshape(-1, 2)
    img = np.zeros((fs.film.height, fs.film.width, 4), np.float32)
    stats = np.zeros(3, np.uint64)
    rc = emu().emu_render_sampler(flat, tiles_xy.ctypes.data, len(tiles_xy), kind, min_spp, max_spp, seed, img.ctypes.data, batch_tiles, stats.ctypes.data)
    assert rc == 0, f"emu_render_sampler: {rc}"
    return img, tuple(int(x) for x in stats)


def render_wavefront(flat, tiles_xy, spp, seed, trace=0, n_chunks=4, trace_blocks=2, lds_depth=0, defines=()):
    """the wavefront schedule (7 stage kernels per round) as SIMT emulations; returns (rgbw image, (samples, vertices, rays, rounds))"""
    fs = flat.contents
    tiles_xy = np.ascontiguousarray(tiles_xy, np.uint32).reshape(-1, 2)
    img = np.zeros((fs.film.height, fs.film.width, 4), np.float32)
    stats = np.zeros(4, np.uint64)
    rc = emu(defines).emu_render_wavefront(flat, tiles_xy.ctypes.data, len(tiles_xy), spp, seed, img.ctypes.data, trace, n_chunks, trace_blocks,
                                         lds_depth, stats.ctypes.data)
    assert rc == 0, f"emu_render_wavefront: {rc}"
    return img, tuple(int(x) for x in stats)


def wf_bin(n_chunks, counts, rays, bmin, bmax, stage=0):
    """the wavefront schedule's ray binning (k_wf_bin_hist + k_wf_bin_scatter) over a queue whose segment s holds counts[s] ray records:
    rays[s] = (counts[s], 8) uint32 words (slot, o, d, flags). Returns (sorted records per segment, their keys per segment)."""
    h = emu()
    counts = np.ascontiguousarray(counts, np.uint32)
    cap = int(h.emu_wf_bin(n_chunks, None, None, None, None, None, None, stage))
    assert len(counts) == 64 and counts.max() <= cap
    queue = np.zeros((64, cap, 8), np.uint32)
    for s_ in range(64):
        queue[s_, :counts[s_]] = rays[s_]
    out = np.zeros_like(queue)
    keys = np.zeros((64, cap), np.uint32)
    bmin = np.ascontiguousarray(bmin, np.float32); bmax = np.ascontiguousarray(bmax, np.float32)
    rc = h.emu_wf_bin(n_chunks, counts.ctypes.data, queue.ctypes.data, out.ctypes.data, keys.ctypes.data, bmin.ctypes.data, bmax.ctypes.data, stage)
    assert rc == cap, "emu_wf_bin failed"
    return [out[s_, :counts[s_]] for s_ in range(64)], [keys[s_, :counts[s_]] for s_ in range(64)]
