"""What ONE GPU does with its share of a frame when the frame is dealt to N GPUs -- the predicted ceiling of the tile-parallel efficiency, measured
on a 1-GPU box (torch-free): shard 0 of N through tray_render_shard_device (16-tile chunks round-robin, as bench.py --gpus N and
tray_render_frame_multi deal them) against the whole frame on the same GPU, for the four bench workloads at their own sample counts.
    gpurun -- 'python tools/eighth_rate.py [N ...]'          (default N = 2 4 8)
efficiency ceiling at N = (time of the whole frame / N) / (time of shard 0 of N); the RCCL sum-reduce (33 MB) and the slowest rank's tile
mix come on top on real hardware. configs[4] is sharded by FRAME first (frames are independent, main.rs:91-106): its tile figure matters only
when there are fewer frames than GPUs."""
import ctypes
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tray_rust_amd as T
from tray_rust_amd import scenes

W, H = 1920, 1080
ns = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
hiprt = ctypes.CDLL("libamdhip64.so")
buf = ctypes.c_void_p()
nbytes = W * H * 4 * 4
assert hiprt.hipMalloc(ctypes.byref(buf), ctypes.c_size_t(nbytes)) == 0
hip = T.Hip(0, seed=1)
print(f"{'workload':34s} {'N':>2s} {'whole frame ms':>15s} {'shard 0 of N ms':>16s} {'Msamples/s on this GPU':>23s} {'efficiency ceiling':>19s}")
for name, spp, frame in (("cornell_box", 1024, 0), ("smallpt", 4096, 0), ("dragon", 2048, 0), ("tr15_like", 512, 64)):
    d = tempfile.mkdtemp(prefix="eighth_")
    if name == "dragon": scenes.write_dragon_assets(d, film=(W, H, spp))
    elif name == "tr15_like": scenes.write_tr15_like_assets(d, film=(W, H, spp))
    else: scenes.write_assets(d, cornell=(W, H, spp), small=(W, H, spp))
    scene, rt, _, fi = T.Scene.load_file(os.path.join(d, name + ".json"))

    def shard_ms(n, reps=2, shard=0, chunk=16):
        best = 1e30
        for _ in range(reps):
            assert hiprt.hipMemset(buf, 0, ctypes.c_size_t(nbytes)) == 0
            hip.render_shard_device(scene, frame, shard, n, spp, buf.value, chunk_tiles=chunk)
            hiprt.hipDeviceSynchronize()
            t = hip.timing(scene)
            best = min(best, t.render_ms)
        return best, int(t.samples)
    shard_ms(8, reps=1)   # warm-up: pools, transform table
    whole, s1 = shard_ms(1)
    for n in ns:
        ms, sn = shard_ms(n)
        print(f"{name + ' ' + str(spp) + ' spp' + (' frame ' + str(frame) if frame else ''):34s} {n:2d} {whole:15.1f} {ms:16.1f} {sn / ms / 1e3:23.1f} {whole / n / ms * (sn * n / s1):19.3f}", flush=True)
    if os.environ.get("EIGHTH_ALL_SHARDS"):   # every shard of 8 (the slowest rank sets the frame's time), for several chunk sizes of the round-robin deal
        for chunk in (16, 4, 1):
            times = [shard_ms(8, reps=1, shard=k, chunk=chunk)[0] for k in range(8)]
            print(f"{name:12s} N = 8, chunks of {chunk:2d} tiles: shard times {' '.join(f'{t:7.1f}' for t in times)} ms -> efficiency {whole / 8 / max(times):.3f} (slowest shard), "
                  f"{whole / sum(times):.3f} (sum of the shards against the whole frame)", flush=True)
