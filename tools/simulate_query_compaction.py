"""CPU-side schedule analysis for the tile kernel's BSDF query stage (Next / C2): replays the wave-synchronous schedule of
k_path_tiles on per-vertex query logs taken from the oracle and counts, per wave step, how many passes through the single BSDF
eval / pdf site are needed
    today           3 if any lane has a light-half query, else 2 (each lane runs its own LIGHT -> MIS -> PATH sequence)
    compacted       ceil(jobs / 64): (lane, query) jobs spread over the lanes of the wave
    + material sort sum over materials of ceil(jobs_m / 64)
    + kind sort     the same per material KIND (matte, plastic, ...: what selects the code path of the eval site)
and the share of lanes that do useful work in those passes.  usage: python tools/simulate_query_compaction.py [scene] [spp] [tiles]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 48
d = "/tmp/sim_qc"
scenes.write_assets(d, cornell=(1920, 1080, spp), small=(1920, 1080, spp))
scene, *_ = T.Scene.load_file(f"{d}/{name}.json")
flat = scene.flatten(0)
o = O.oracle()
o.oracle_path_profile.restype = C.c_int
o.oracle_path_profile.argtypes = [C.POINTER(T._lib.TrayFlatScene), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(1)
tiles = [(int(rng.integers(40, 200)), int(rng.integers(20, 115))) for _ in range(n_tiles)]   # tiles that see the box
tot = {"steps": 0, "units_today": 0, "units_aligned": 0, "passes_aligned": 0, "today": 0, "compact": 0, "sorted": 0, "by_kind": 0, "jobs": 0, "alive": 0, "mixed": 0, "mixed_kind": 0}
kind_of = [flat.contents.materials[m].kind for m in range(flat.contents.n_materials)]
per_kind = {}
per_mat = {}
for tx, ty in tiles:
    px = np.repeat(np.arange(64) % 8 + tx * 8, spp).astype(np.uint32)
    py = np.repeat(np.arange(64) // 8 + ty * 8, spp).astype(np.uint32)
    si = np.tile(np.arange(spp), 64).astype(np.uint32)
    codes = np.zeros((64 * spp, 16), np.uint8); counts = np.zeros(64 * spp, np.uint8)
    assert o.oracle_path_profile(flat, len(px), px.ctypes.data, py.ctypes.data, si.ctypes.data, spp, 1, codes.ctypes.data, counts.ctypes.data) == 0
    codes = codes.reshape(64, spp, 16); counts = counts.reshape(64, spp)
    for wave in range(4):   # wave w takes samples w, w + 4, ... of every pixel (lane = pixel)
        sample = np.full(64, wave); vertex = np.zeros(64, int)
        while True:
            live = sample < spp
            if not live.any():
                break
            jobs_by_mat = {}
            n_light = n_alive = 0
            kinds_in_pass = [set(), set(), set()]   # query kinds (LIGHT / MIS / PATH) that pass p of today's schedule has to run
            for lane in np.nonzero(live)[0]:
                s = sample[lane]
                if vertex[lane] < counts[lane, s]:
                    c = int(codes[lane, s, vertex[lane]])
                    j = 1 + (1 if c & 32 else 0) + (1 if c & 64 else 0)
                    jobs_by_mat[c & 31] = jobs_by_mat.get(c & 31, 0) + j
                    n_light += 1 if c & 32 else 0
                    seq = (["L"] if c & 32 else []) + (["M"] if c & 64 else []) + ["P"]   # the lane's own LIGHT -> MIS -> PATH sequence
                    for p_, k_ in enumerate(seq):
                        kinds_in_pass[p_].add(k_)
                    n_alive += 1
                    vertex[lane] += 1
                    if vertex[lane] >= counts[lane, s]:
                        sample[lane] += 4; vertex[lane] = 0
                else:   # camera miss: the lane only traced stage A this step
                    sample[lane] += 4; vertex[lane] = 0
            if n_alive == 0:
                continue
            jobs = sum(jobs_by_mat.values())
            tot["steps"] += 1; tot["alive"] += n_alive; tot["jobs"] += jobs
            tot["today"] += 3 if n_light else 2
            tot["units_today"] += sum(len(k) for k in kinds_in_pass)            # a pass runs the head / epilogue code of every kind present in it
            tot["units_aligned"] += len(set().union(*kinds_in_pass))           # aligned schedule: pass k serves kind k only
            tot["passes_aligned"] += len(set().union(*kinds_in_pass))
            tot["compact"] += -(-jobs // 64)
            tot["sorted"] += sum(-(-j // 64) for j in jobs_by_mat.values())
            tot["mixed"] += 1 if len(jobs_by_mat) > 1 else 0
            by_kind = {}
            for m, j in jobs_by_mat.items():
                by_kind[kind_of[m]] = by_kind.get(kind_of[m], 0) + j
            tot["by_kind"] += sum(-(-j // 64) for j in by_kind.values())
            tot["mixed_kind"] += 1 if len(by_kind) > 1 else 0
            for k, j in by_kind.items():
                a = per_kind.setdefault(k, [0, 0])
                a[0] += j; a[1] += -(-j // 64)
            for m, j in jobs_by_mat.items():
                a = per_mat.setdefault(m, [0, 0])
                a[0] += j; a[1] += -(-j // 64)
s = tot["steps"]
print(f"{name} {spp} spp, {n_tiles} tiles: {s} wave steps, {tot['alive'] / s:.1f} lanes with a vertex per step, {tot['jobs'] / s:.1f} query jobs per step")
print(f"  steps with more than one material among the live lanes: {100 * tot['mixed'] / s:.0f} %")
fs = flat.contents
kinds = {0: "matte", 1: "plastic", 2: "metal", 3: "glass", 4: "rough_glass", 5: "specular_metal", 6: "merl"}
for m, (j, p) in sorted(per_mat.items()):
    print(f"  material {m} ({kinds.get(fs.materials[m].kind, '?')}): {100 * j / tot['jobs']:.0f} % of the jobs, {p / s:.2f} material-pure passes per step")
print(f"  steps with more than one material KIND (= code path of the eval site): {100 * tot['mixed_kind'] / s:.0f} %")
for k, (j, p) in sorted(per_kind.items()):
    print(f"  kind {kinds.get(k, '?')}: {100 * j / tot['jobs']:.0f} % of the jobs, {p / s:.2f} kind-pure passes per step")
print(f"  query kinds run per step (each = that kind's sample head / epilogue once for the wave): today {tot['units_today'] / s:.2f} in {tot['today'] / s:.2f} passes, "
      f"aligned schedule (pass k serves query kind k; measured on the GPU in round 1: no effect) {tot['units_aligned'] / s:.2f} in {tot['passes_aligned'] / s:.2f} passes")
for k in ("today", "compact", "sorted", "by_kind"):
    print(f"  {k:8s} {tot[k] / s:.2f} passes per step, useful lanes {100 * tot['jobs'] / (64 * tot[k]):.0f} %")
