"""Items per tile of the tile kernel's launch (k_path_tiles: progressive slices, level-major; TRAYHIP_TILE_SLICES) -- the whole frame on one GPU and
the eight shards of the frame dealt to 8 GPUs (16-tile chunks round-robin), torch-free:
    gpurun -- 'python tools/tile_slices_ab.py [workload ...]'
per setting: whole-frame ms, the eight shard times, efficiency = whole / 8 / slowest shard. `auto` = the library's own rule."""
import ctypes
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tray_rust_amd as T
from tray_rust_amd import scenes

W, H = 1920, 1080
hiprt = ctypes.CDLL("libamdhip64.so")
buf = ctypes.c_void_p()
nbytes = W * H * 4 * 4
assert hiprt.hipMalloc(ctypes.byref(buf), ctypes.c_size_t(nbytes)) == 0
hip = T.Hip(0, seed=1)
specs = {"cornell_box": 1024, "smallpt": 4096, "dragon": 2048}
for name in (sys.argv[1:] or ["dragon", "cornell_box"]):
    spp = specs[name]
    d = tempfile.mkdtemp(prefix="slices_")
    if name == "dragon": scenes.write_dragon_assets(d, film=(W, H, spp))
    else: scenes.write_assets(d, cornell=(W, H, spp), small=(W, H, spp))
    scene, rt, _, fi = T.Scene.load_file(os.path.join(d, name + ".json"))

    def ms(shard, n):
        assert hiprt.hipMemset(buf, 0, ctypes.c_size_t(nbytes)) == 0
        hip.render_shard_device(scene, 0, shard, n, spp, buf.value)
        hiprt.hipDeviceSynchronize()
        return hip.timing(scene).render_ms
    ms(0, 8)
    base = None
    for setting in ("auto", "1", "2", "3", "4", "5"):
        if setting == "auto": os.environ.pop("TRAYHIP_TILE_SLICES", None)
        else: os.environ["TRAYHIP_TILE_SLICES"] = setting
        whole = min(ms(0, 1), ms(0, 1))
        if setting == "1": base = whole
        shards = [ms(k, 8) for k in range(8)]
        ref = base if base else whole
        print(f"{name:12s} {spp} spp, items per tile {setting:>4s}: whole frame {whole:8.1f} ms ({W * H * spp / whole / 1e3:7.1f} Msamples/s) | 8 shards " +
              " ".join(f"{t:6.1f}" for t in shards) + f" ms | efficiency at 8 GPUs {whole / 8 / max(shards):.3f} (against this setting's whole frame), "
              f"{ref / 8 / max(shards):.3f} (against whole tiles)", flush=True)
