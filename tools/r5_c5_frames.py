"""configs[4] on BASELINE's own frames: the tr15 stand-in at full detail, 1920x1080, 512 spp, frames 0, 16, ..., 112, 127 of the 128-frame sequence
(main.rs:91-106 loops them; scene.rs:152-176 moves the scene), one device scene walked from frame to frame with tray_scene_update_frame.
Per frame: instances that move within the shutter interval, Msamples/s of the frame's kernels (HIP events), launches, vertices per sample,
the schedule (pool slots / views / slices) and the per-path transform cache.  Torch-free:
    python tools/r5_c5_frames.py [spp] [frame ...]    -> gpurun_out/r05_c5_frames.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tray_rust_amd as T
from tray_rust_amd import scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
frames = [int(a) for a in sys.argv[2:]] or [0, 16, 32, 48, 64, 80, 96, 112, 127]
d = "/tmp/c5"
if not os.path.exists(os.path.join(d, "tr15_like.json")):
    scenes.write_tr15_like_assets(d, film=(1920, 1080, spp))
scene, rt, _, fi = T.Scene.load_file(os.path.join(d, "tr15_like.json"))
hip = T.Hip(0, seed=1)
out = open(os.path.join(ROOT, "gpurun_out", "r05_c5_frames.txt"), "w")


def say(line):
    print(line, flush=True); out.write(line + "\n"); out.flush()


say(f"# tr15 stand-in, full detail (59 instances, 3.1 M triangles), 1920x1080, {spp} spp; one device scene, tray_scene_update_frame between frames")
say("# frame  moving  Msamples/s  kernel ms  update+flatten s  launches  V       pool slots  views  slices  xf cache GB  transform table GB")
total_s, total_ms = 0, 0.0
t_wall0 = time.time()
for fr in frames:
    t0 = time.time()
    flat = scene.flatten(fr).contents
    moving = sum(1 for i in range(flat.n_instances) if flat.instances[i].animated)
    fi_f = T.FrameInfo(fi.frames, fi.time, fr, fr)
    rt.clear()
    scene.device_scene(fr, 0)
    t_up = time.time() - t0
    sys.stdout = open(os.devnull, "w")
    try:
        hip.render(scene, rt, T.Config(d, "tr15_like", spp, 1, fi_f, (0, 0)))
    finally:
        sys.stdout = sys.__stdout__
    t = hip.last_timing
    sch = hip.schedule(scene)
    total_s += t.samples; total_ms += t.render_ms
    say(f"{fr:7d}  {moving:6d}  {t.samples / t.render_ms / 1e3:10.1f}  {t.render_ms:9.1f}  {t_up:16.2f}  {t.launches:8d}  {t.vertices / t.samples:.4f}  "
        f"{sch['pool_slots']:10d}  {sch['views']:5d}  {sch['slices']:6d}  {sch['xf_cache_bytes'] / 2**30:11.1f}  {sch['xf_table_bytes'] / 2**30:18.1f}")
say(f"# all {len(frames)} frames: {total_s / total_ms / 1e3:.1f} Msamples/s over the frames' kernels")
wall = time.time() - t_wall0   # (flatten + tray_scene_update_frame + kernels + the copy of every film to the host: what main.rs:91-106 loops, without the PNG writes)
say(f"# wall clock of the loop over the frames (first frame's device-scene build included): {wall:.1f} s = {total_s / wall / 1e6:.1f} Msamples/s")
