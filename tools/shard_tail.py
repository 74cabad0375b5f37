"""What one GPU of an N-GPU node does (torch-free): shard 0 of N of a frame through tray_render_shard_device, timed by the library's
HIP events, with and without the tile slices of launch_tiles -- the tail of the persistent workgroups at small tile counts.
    gpurun -- 'python tools/shard_tail.py [scene] [spp]'"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 3:   # child: one measurement (the library reads TRAYHIP_TILE_SLICES per launch, a fresh process keeps things simple)
    import tray_rust_amd as T
    from tray_rust_amd import scenes
    name, spp, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    d = "/tmp/shard_tail"
    scenes.write_assets(d, cornell=(1920, 1080, spp), small=(1920, 1080, spp))
    scene, rt, _, fi = T.Scene.load_file(os.path.join(d, name + ".json"))
    hip = T.Hip(0, seed=1)
    buf = ctypes.c_void_p()
    hiprt = ctypes.CDLL("libamdhip64.so")
    nbytes = 1920 * 1080 * 4 * 4
    assert hiprt.hipMalloc(ctypes.byref(buf), ctypes.c_size_t(nbytes)) == 0 and hiprt.hipMemset(buf, 0, ctypes.c_size_t(nbytes)) == 0
    best = 1e9
    for rep in range(2):
        hip.render_shard_device(scene, 0, 0, n, spp, buf.value)
        hiprt.hipDeviceSynchronize()
        t = hip.timing(scene)
        best = min(best, t.render_ms)
    print(f"{name} {spp} spp, shard 0 of {n}, TRAYHIP_TILE_SLICES={os.environ.get('TRAYHIP_TILE_SLICES', 'auto'):4s}: {best:8.2f} ms  {t.samples / best / 1e3:7.1f} Msamples/s on this GPU -> x{n} = {n * t.samples / best / 1e3:7.1f}", flush=True)
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
spp = sys.argv[2] if len(sys.argv) > 2 else "1024"
for n in ("1", "2", "4", "8"):
    for slices in ("1", "2", "4", None):
        env = dict(os.environ)
        if slices: env["TRAYHIP_TILE_SLICES"] = slices
        else: env.pop("TRAYHIP_TILE_SLICES", None)
        subprocess.run([sys.executable, __file__, name, spp, n], env=env)
