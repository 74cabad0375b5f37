"""Where a frame of configs[4] spends its wall clock beside the kernels: python tools/frame_overheads.py [spp] [frame ...] (GPU; tr15 stand-in at full detail).
Per frame: the host's flatten, tray_scene_update_frame (device_scene), rt.clear, hip.render's wall clock against the kernels' HIP-event time inside it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tray_rust_amd as T
from tray_rust_amd import scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = [int(a) for a in sys.argv[2:]] or [60, 61, 62, 63, 64, 65]
d = "/tmp/c5"
if not os.path.exists(os.path.join(d, "tr15_like.json")):
    scenes.write_tr15_like_assets(d, film=(1920, 1080, spp))
scene, rt, _, fi = T.Scene.load_file(os.path.join(d, "tr15_like.json"))
hip = T.Hip(0, seed=1)
print("# frame  flatten ms  update_frame ms  rt.clear ms  render wall ms  kernels ms  render - kernels ms")
for fr in frames:
    t0 = time.time(); scene.flatten(fr); t1 = time.time()
    scene.device_scene(fr, 0); t2 = time.time()
    rt.clear(); t3 = time.time()
    sys.stdout = open(os.devnull, "w")
    try:
        hip.render(scene, rt, T.Config(d, "tr15_like", spp, 1, T.FrameInfo(fi.frames, fi.time, fr, fr), (0, 0)))
    finally:
        sys.stdout = sys.__stdout__
    t4 = time.time()
    k = hip.last_timing.render_ms
    print(f"{fr:7d}  {1e3 * (t1 - t0):10.1f}  {1e3 * (t2 - t1):15.1f}  {1e3 * (t3 - t2):11.1f}  {1e3 * (t4 - t3):14.1f}  {k:10.1f}  {1e3 * (t4 - t3) - k:19.1f}", flush=True)
