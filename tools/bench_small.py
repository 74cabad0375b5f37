"""Short fixed workload for profiling: cornell 1920x1080 at --spp (default 64), prints Msamples/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tray_rust_amd as T
from tray_rust_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
scene_name = sys.argv[3] if len(sys.argv) > 3 else "cornell_box"
d = "/tmp/sc_small"
frame = 0
if scene_name == "tr15_like":
    scenes.write_tr15_like_assets(d, film=(1920, 1080, spp), detail=float(os.environ.get("TR15_DETAIL", "1.0")))
    frame = int(os.environ.get("TR15_FRAME", "330"))
    if "TR15_SHUTTER" in os.environ:   # 0 = closed shutter: nothing moves within a frame, the static kernels run
        import json
        desc = json.load(open(f"{d}/tr15_like.json")); desc["camera"]["shutter_size"] = float(os.environ["TR15_SHUTTER"])
        json.dump(desc, open(f"{d}/tr15_like.json", "w"))
elif scene_name == "moving_box":
    scenes.write_moving_box(d, width=1920, height=1080, samples=spp)
    frame = int(os.environ.get("MOVING_FRAME", "3"))
elif scene_name == "dragon":
    scenes.write_dragon_assets(d, film=(1920, 1080, spp), extent=float(os.environ.get("DRAGON_EXTENT", "0.2")))
else:
    scenes.write_assets(d, cornell=(1920, 1080, spp), small=(1920, 1080, spp))
scene, rt, spp, fi = T.Scene.load_file(f"{d}/{scene_name}.json")
hip = T.Hip(0, seed=1)
buf = torch.zeros(1080 * 1920 * 4, dtype=torch.float32, device="cuda")
for rep in range(reps):
    hip.render_device(scene, frame, (0, 0), spp, buf.data_ptr())
    tim = hip.timing(scene)
    print(f"{scene_name} 1080p {spp}spp: kernel ms {tim.render_ms:.2f} Msamples/s {tim.samples / tim.render_ms / 1e3:.2f} V {tim.vertices / tim.samples:.3f} rays/sample {tim.rays / tim.samples:.3f}", flush=True)
