"""Torch-free A/B of library builds in seconds: renders prepared scenes through the C ABI and prints Msamples/s from the library's
own HIP events (tray_last_timing). One process per library (TRAYHIP_LIB is read at import):
    python tools/mini_ab.py prepare <dir>                 # writes the scenes once
    TRAYHIP_LIB=... python tools/mini_ab.py run <dir> <label> cornell_box:64 dragon:32 tr15_like:16 ...
Also prints the pixel RMSE of each render against the first library's result stored in <dir> (parity of the variant on the GPU)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes

W, H = 1920, 1080
mode, d = sys.argv[1], sys.argv[2]
if mode == "prepare":
    scenes.write_assets(d, cornell=(W, H, 64), small=(W, H, 64))
    grid = int(os.environ.get("MINI_DRAGON_GRID", "220"))                          # 220: 96 800 triangles, loads in a second; 660: the 871 200 of C4
    scenes.write_dragon_assets(d, film=(W, H, 32), grid=grid, extent=0.2)   # (the bench workload: grid 660, extent 0.2)
    scenes.write_moving_box(d, width=W, height=H, samples=32)   # the moving test scene at film size (tile kernel, ANIM instantiation)
    scenes.write_tr15_like_assets(os.path.join(d, "tr15"), film=(W, H, 16), detail=float(os.environ.get("MINI_TR15_DETAIL", "0.15")))   # own directory: it brings its own models/
    sys.exit(0)
label = sys.argv[3]
hip = T.Hip(device=0, seed=1)
for item in sys.argv[4:]:
    name, spp = item.split(":")
    spp = int(spp)
    frame = int(os.environ.get("TR15_FRAME", "64")) if name == "tr15_like" else (3 if name == "moving_box" else 0)   # (64: inside BASELINE's 0..127, 11 instances move; rounds 2-4 profiled frame 330)
    t0 = time.time()
    scene, rt, _, fi = T.Scene.load_file(os.path.join(d, "tr15" if name == "tr15_like" else "", name + ".json"))
    if frame:
        fi = T.FrameInfo(fi.frames, fi.time, frame, frame)   # Config.current_frame = frame_info.start
    best = 0.0
    for rep in range(2):
        rt.clear()
        sys.stdout = open(os.devnull, "w")
        try:
            hip.render(scene, rt, T.Config(d, name, spp, 1, fi, (0, 0)))
        finally:
            sys.stdout = sys.__stdout__
        t = hip.last_timing
        best = max(best, t.samples / (t.render_ms * 1e-3) / 1e6)
    img = rt.get_renderf32().reshape(H, W, 4)
    rgb = img[..., :3] / np.maximum(img[..., 3:], 1e-20)
    ref_path = os.path.join(d, f"ref_{name}.npy")
    if os.path.exists(ref_path):
        ref = np.load(ref_path)
        r = float(np.sqrt(np.mean((rgb - ref) ** 2)))
    else:
        np.save(ref_path, rgb); r = 0.0
    try:   # the schedule the launch ran with (tools/pmc_workloads.py records it beside the counters)
        import json
        sch = hip.schedule(scene); sch["frame"] = frame; sch["spp"] = spp
        json.dump(sch, open(os.path.join(d, f"schedule_{name}.json"), "w"))
    except Exception as e:
        print("no schedule info:", e)
    print(f"{label:8s} {name:12s} {spp:3d} spp  {best:8.1f} Msamples/s  launches {t.launches:4d}  RMSE vs first {r:.2e}  (load+render {time.time() - t0:.1f}s)", flush=True)
