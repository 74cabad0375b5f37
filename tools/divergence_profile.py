"""Divergence profile of a kernel WITHOUT a GPU: the device source runs in the SIMT emulation of tests/emu (fibers, wave intrinsics
as rendezvous), compiled with -fno-inline -finstrument-functions. Between two rendezvous the lanes of a wave each run one segment;
a device function that several lanes call in the same segment is executed together on the GPU, as often as the lane that calls
it most. Per function the table gives
    wave calls / step     how often a wave executes it per wave step (sum over segments of the max over lanes; a step = one pass
                          of a wave through the three stages = one call of vertex_queries / k_wf_query's body)
    lanes                 average share of the 64 lanes that are in it when it runs (lane calls / (64 x wave calls))
    bytes                 size of its x86 body at -O1 without inlining -- a rough stand-in for its instruction count
    share                 wave calls x bytes, normalised: where the wave's issue slots go, to first order
It sees structure (who runs what together, how full the wave is), not timing.
usage: python tools/divergence_profile.py [cornell_box|smallpt|dragon] [--wavefront] [--define TR_MESH_TWO_CHILDREN]"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes

ap = argparse.ArgumentParser()
ap.add_argument("scene", nargs="?", default="cornell_box")
ap.add_argument("--wavefront", action="store_true", help="profile the wavefront schedule (k_wf_trace_dyn) instead of k_path_tiles")
ap.add_argument("--define", action="append", default=[])
ap.add_argument("--size", default="48x32x16")
ap.add_argument("--top", type=int, default=32)
ap.add_argument("--tiles", type=int, default=0, help="render only this many tiles, spread evenly over the Morton queue (a 1920x1080 film's tiles see coherent camera rays: what the bench runs)")
ap.add_argument("--grid", type=int, default=48, help="dragon: quads per side of the stand-in's grid (660 = the 871 200 triangles of C4)")
ap.add_argument("--extent", type=float, default=1.0, help="dragon: size of the mesh (0.2 = the bench workload)")
ap.add_argument("--min-bytes", type=int, default=250, help="hide helpers smaller than this (vector operators: their call overhead is not device work)")
args = ap.parse_args()
w, h, spp = (int(x) for x in args.size.split("x"))

emu_dir = os.path.join(ROOT, "tests", "emu")
so = os.path.join(emu_dir, "libtrayemu_prof" + "".join("_" + d.lower() for d in sorted(args.define)) + ".so")
subprocess.run(["g++", "-O1", "-fno-inline", "-finstrument-functions", "-finstrument-functions-exclude-file-list=hip_emu.h,/usr/include,/usr/lib",
                "-DTR_EMU_PROFILE", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-attributes", "-shared", "-o", so,
                os.path.join(emu_dir, "emu_kernels.cpp"), "-ldl", "-Wl,-Bsymbolic-functions"] + ["-D" + d for d in args.define], check=True)
lib = C.CDLL(so)
FS = C.POINTER(T._lib.TrayFlatScene)
lib.emu_render_tiles.restype = C.c_int
lib.emu_render_tiles.argtypes = [FS, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
lib.emu_render_wavefront.restype = C.c_int
lib.emu_render_wavefront.argtypes = [FS, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
lib.emu_profile_dump.argtypes = [C.c_char_p]

d = tempfile.mkdtemp(prefix="divprof")
scenes.write_assets(d, cornell=(w, h, spp), small=(w, h, spp))
if args.scene == "dragon":
    scenes.write_dragon_assets(d, film=(w, h, spp), grid=args.grid, extent=args.extent)
scene, *_ = T.Scene.load_file(os.path.join(d, args.scene + ".json"))
flat = scene.flatten(0)
tiles = np.array(T.BlockQueue((w, h), (8, 8)).blocks, np.uint32).reshape(-1, 2)
if args.tiles:
    tiles = np.ascontiguousarray(tiles[(np.arange(args.tiles) * 2 + 1) * len(tiles) // (2 * args.tiles)])
img = np.zeros((h, w, 4), np.float32)
stats = np.zeros(4, np.uint64)
lib.emu_profile_start()
if args.wavefront:
    rc = lib.emu_render_wavefront(flat, tiles.ctypes.data, len(tiles), spp, 1, img.ctypes.data, 0, 8, 2, 0, stats.ctypes.data)
else:
    rc = lib.emu_render_tiles(flat, tiles.ctypes.data, len(tiles), spp, 1, img.ctypes.data, 1, -1, -1, stats.ctypes.data, 0, 0, 1)
assert rc == 0, rc
dump = os.path.join(d, "profile.txt")
assert lib.emu_profile_dump(dump.encode()) > 0
samples, vertices = int(stats[0]), int(stats[1])

sizes = {}
for line in subprocess.run(["nm", "-S", "--defined-only", so], capture_output=True, text=True).stdout.splitlines():
    f = line.split()
    if len(f) == 4:
        sizes[f[3]] = int(f[1], 16)
rows = []
phase_rows = []
for line in open(dump):
    lane_calls, wave_calls, lanes, segments, name = line.split()
    if "@" in name:
        phase_rows.append((name, int(lane_calls), int(wave_calls)))
        continue
    rows.append((name, int(lane_calls), int(wave_calls), sizes.get(name, 0)))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
rows = [r for r in rows]
steps = max([r[2] for r, n in zip(rows, names) if "vertex_queries" in n] or [1])
big = [i for i in range(len(rows)) if rows[i][3] >= args.min_bytes]
total = sum(rows[i][2] * rows[i][3] for i in big) or 1
print(f"{args.scene} {w}x{h}x{spp}{' wavefront schedule' if args.wavefront else ' k_path_tiles'}{' ' + ' '.join(args.define) if args.define else ''}: "
      f"{samples} samples, {vertices} vertices, {steps} wave steps ({vertices / steps:.1f} of 64 lanes shade a vertex per step); "
      f"sum of wave calls x bytes per step: {total / steps:.0f}")
print(f"{'share':>6} {'wave calls/step':>16} {'lanes':>6} {'bytes':>6}  function")
order = sorted(big, key=lambda i: -rows[i][2] * rows[i][3])
for i in order[:args.top]:
    name, lane_calls, wave_calls, size = rows[i]
    short = names[i].split("(")[0].replace("tr::", "")
    print(f"{100 * wave_calls * size / total:5.1f}% {wave_calls / steps:16.2f} {100 * lane_calls / (64 * wave_calls):5.0f}% {size:6d}  {short}")

if phase_rows:   # TR_PROFILE_PHASES=1: the query passes one by one (1 = LIGHT, 2 = MIS, 3 = PATH)
    pn = subprocess.run(["c++filt"], input="\n".join(r[0].split("@")[0] for r in phase_rows), capture_output=True, text=True).stdout.splitlines()
    print("per query pass (phase 1 = LIGHT, 2 = MIS, 3 = PATH):")
    for (name, lane_calls, wave_calls), dn in sorted(zip(phase_rows, pn), key=lambda x: (x[1], x[0][0])):
        ph = name.split("@")[1]
        if ph != "0" and sizes.get(name.split("@")[0], 0) >= args.min_bytes:
            print(f"  pass {ph}  {wave_calls / steps:8.2f} wave calls/step  {100 * lane_calls / (64 * max(wave_calls, 1)):5.0f}% lanes  {dn.split('(')[0].replace('tr::', '')}")
