"""C4 stand-in at full mesh size (871 200 triangles + MERL): parity against the oracle on a reduced film, then throughput
at 1920x1080. Run on the GPU box: python tools/dragon_gpu.py [grid] [spp]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
import _oracle as O

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 660
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
extent = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
d = "/tmp/dragon_assets"
t = time.time(); path, ntri = scenes.write_dragon_assets(d, film=(480, 272, 32), grid=grid, extent=extent); t_write = time.time() - t
t = time.time(); scene, rt, _, fi = T.Scene.load_file(path); t_load = time.time() - t
flat = scene.flatten(0)
out = {"extent": extent, "tris": ntri, "write_s": round(t_write, 2), "load_s": round(t_load, 2), "mesh_nodes": int(flat.contents.n_mesh_nodes)}
rng = np.random.default_rng(1)
rays = O.camera_rays(flat, rng.uniform(0, [480, 272], (200000, 2)))
dev = scene.device_scene(0, 0)
hits = np.zeros(len(rays), dtype=O.HIT_DTYPE)
T.check(T.lib().tray_debug_intersect(dev, len(rays), rays.ctypes.data, hits.ctypes.data))
ref = O.intersect(flat, rays)
out["intersect_equal"] = bool((ref["inst"] == hits["inst"]).all() and (ref["prim"] == hits["prim"]).all() and (ref["t"] == hits["t"])[ref["inst"] != 0xffffffff].all())
m = ref["inst"] == 6
for f in ("p", "n", "ng", "u", "v", "dp_du", "dp_dv"):
    out["maxdiff_" + f] = float(np.abs(ref[f][m] - hits[f][m]).max())
out["mesh_hit_frac"] = float((ref["inst"] == 6).mean())
hip = T.Hip(0, seed=1)
rt.clear(); hip.render(scene, rt, T.Config(".", "s", 32, 1, fi, (0, 0)))
gpu = rt.get_renderf32().reshape(272, 480, 4).copy(); tim = hip.last_timing
t = time.time(); cpu, st = O.render_tiles(flat, 32, seed=1); t_cpu = time.time() - t
rgb = lambda i: i[..., :3] / np.maximum(i[..., 3:], 1e-20)
out["rmse_480x272x32"] = float(np.sqrt(np.mean((rgb(gpu) - rgb(cpu)) ** 2)))
out["vertices_gpu"], out["vertices_cpu"] = int(tim.vertices), int(st.vertices)
out["oracle_msamples_s"] = st.samples / t_cpu / 1e6
scene.release_device()
# throughput at the C4 film size
big = scenes.dragon_scene(1920, 1080, spp)
p2 = os.path.join(d, "dragon_big.json"); json.dump(big, open(p2, "w"))
scene2, rt2, _, fi2 = T.Scene.load_file(p2)
for mode in ("mega", "wave"):
    os.environ["TRAYHIP_MODE"] = mode
    scene2.release_device()
    hip = T.Hip(0, seed=1)
    best = 0
    for it in range(3):
        rt2.clear(); hip.render(scene2, rt2, T.Config(".", "s", spp, 1, fi2, (0, 0)))
        tm = hip.last_timing
        best = max(best, tm.samples / (tm.render_ms * 1e-3) / 1e6)
    out[f"{mode}_msamples_s_1080p_{spp}spp"] = best
    out[f"{mode}_vertices_per_sample"] = tm.vertices / tm.samples
    out[f"{mode}_rays_per_sample"] = tm.rays / tm.samples
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/dragon_gpu_{extent}.json", "w"), indent=1)
