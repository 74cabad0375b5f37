"""Multi-GPU readiness without the hardware: the library's in-process multi-device path -- tray_multi_create,
tray_render_frame_multi, tray_multi_update_frame, tray_multi_set_wavefront, tray_multi_destroy: one host thread and stream per device, ONE
grouped ncclReduce(sum, root = first device), the caller's current device restored -- executed in this container against stand-ins for the
HIP runtime (tests/stubs/fakehip.c, LD_PRELOAD) and for RCCL (tests/stubs/fakerccl.c, the dlopen target "librccl.so"). Kernel launches do
nothing but log their arguments and leave a mark per device in the film, the stand-in reduce really sums host buffers: this checks the
plumbing (which device is current at every call, shard arguments, group pairing, counts, root, error path), not pixels."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "stubs")

DRIVER = r'''
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import tray_rust_amd as T
from tray_rust_amd import scenes
d = %(tmp)r
scenes.write_assets(d, cornell=(%(w)d, %(h)d, 4))
scene, rt, spp, fi = T.Scene.load_file(os.path.join(d, "cornell_box.json"))
lib = T.lib()
T.check(lib.tray_init(1))                       # the caller's current device: must still be 1 afterwards
hip = T.Hip(device=1, seed=3)
n_dev = int(os.environ["FAKEHIP_DEVICES"])
cfg = T.Config(d, "cornell_box.json", spp, 1, fi, (0, 0))
try:
    per, ms = hip.render_multi(scene, rt, cfg, list(range(n_dev)))
    print("RENDER_OK", rt.pixels[0], len(per), ms)
    T.check(lib.tray_multi_set_wavefront(hip._multi, 1 << 20, 2, 4))
    per, ms = hip.render_multi(scene, rt, cfg, list(range(n_dev)))      # same frame again: no update, communicators kept
    print("RENDER_OK", rt.pixels[0], len(per), ms)
except T.TrayError as e:
    print("TRAY_ERROR", e.code, e.message)
cur = C.c_int(-1)
import ctypes.util
print("CURRENT_DEVICE", lib.tray_init and __import__("ctypes").CDLL(None).hipGetDevice(C.byref(cur)), cur.value)
hip.close_multi()
print("DONE")
'''


@pytest.fixture(scope="module")
def stubs(tmp_path_factory):
    d = tmp_path_factory.mktemp("stubs")
    hip = str(d / "libfakehip.so")
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", hip, os.path.join(STUBS, "fakehip.c"), "-lpthread"], check=True)
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", str(d / "librccl.so"), os.path.join(STUBS, "fakerccl.c"), "-ldl"], check=True)
    return d, hip


def run(stubs, tmp_path, n_dev, fail=None, w=64, h=48):
    d, hip = stubs
    log = str(tmp_path / "calls.log")
    env = dict(os.environ, LD_PRELOAD=hip, LD_LIBRARY_PATH=str(d) + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FAKEHIP_LOG=log,
               FAKEHIP_DEVICES=str(n_dev), FAKEHIP_TILE_KERNEL="1")
    if fail:
        env["FAKERCCL_FAIL"] = fail
    out = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT, "tmp": str(tmp_path), "w": w, "h": h}], env=env, capture_output=True, text=True, timeout=300)
    return out, open(log).read().splitlines() if os.path.exists(log) else []


@pytest.mark.parametrize("n_dev,w,h", [(2, 64, 48), (8, 256, 128), (8, 64, 48)])   # the last: 48 tiles = 3 chunks of 16 on 8 devices, five shards are empty
def test_in_process_multi_device_path_against_stub_runtimes(stubs, tmp_path, n_dev, w, h, built):
    out, log = run(stubs, tmp_path, n_dev, w=w, h=h)
    assert "DONE" in out.stdout, out.stdout + out.stderr
    renders = [l for l in out.stdout.splitlines() if l.startswith("RENDER_OK")]
    assert len(renders) == 2, out.stdout + out.stderr
    # every device's kernel left (device + 1) in word 0 of its film; ONE sum-reduce onto the first device, added into the caller's buffer:
    # 1 + 2 + ... + n after the first frame, twice that after the second (film::Image::add_blocks semantics: the call ADDS)
    n_tiles = (w // 8) * (h // 8)
    busy = min(n_dev, (n_tiles + 15) // 16)                                                  # devices whose shard holds a chunk of 16 tiles
    want = busy * (busy + 1) / 2
    assert float(renders[0].split()[1]) == want and float(renders[1].split()[1]) == 2 * want, renders
    assert re.search(r"CURRENT_DEVICE 0 1\b", out.stdout), out.stdout                      # the caller's device is current again
    launches = [l for l in log if l.startswith("launch")]
    assert len(launches) == 2 * busy
    for frame in range(2):
        seen = {}
        for l in launches[frame * busy:(frame + 1) * busy]:
            kv = dict(p.split("=") for p in l.split()[1:])
            seen[int(kv["dev"])] = kv
            assert int(kv["chunk"]) == 16 and int(kv["chunk_stride"]) == n_dev and int(kv["spp"]) == 4
        assert sorted(seen) == list(range(busy))                                            # one launch per device that has tiles, each on its own device
        assert sum(int(kv["tile_count"]) for kv in seen.values()) == n_tiles                # the shards partition the frame's tiles
        assert len({kv["stream"] for kv in seen.values()}) == busy and len({kv["film"] for kv in seen.values()}) == busy
    # the collective: ncclCommInitAll once, per frame ONE group of n reduces -- rank r issued while device r is current, on device r's
    # stream, count = W * H * 4 floats, float (7), sum (0), root 0
    assert sum(1 for l in log if l.startswith("nccl_comm_init_all")) == 1
    groups = "\n".join(l for l in log if l.startswith("nccl_")).split("nccl_group_start")[1:]
    assert len(groups) == 2
    streams = {int(dict(p.split("=") for p in l.split()[1:])["dev"]): dict(p.split("=") for p in l.split()[1:])["stream"] for l in log if l.startswith("stream_create")}
    for g in groups:
        reduces = [dict(p.split("=") for p in l.split()[1:]) for l in g.splitlines() if l.startswith("nccl_reduce")]
        assert [int(r["rank"]) for r in reduces] == list(range(n_dev)) and "nccl_group_end" in g
        for r in reduces:
            assert r["comm_dev"] == r["current_dev"] == r["rank"] and int(r["count"]) == w * h * 4
            assert (r["dtype"], r["op"], r["root"], r["in_group"]) == ("7", "0", "0", "1")
            assert r["stream"] == streams[int(r["rank"])]
    assert sum(1 for l in log if l.startswith("nccl_comm_destroy")) == n_dev


def test_a_failing_collective_is_reported_and_the_current_device_restored(stubs, tmp_path, built):
    out, log = run(stubs, tmp_path, 2, fail="reduce")
    assert "TRAY_ERROR -5" in out.stdout and "ncclReduce failed" in out.stdout, out.stdout + out.stderr   # TRAY_E_DEVICE
    assert re.search(r"CURRENT_DEVICE 0 1\b", out.stdout), out.stdout
    assert any(l.startswith("nccl_group_end") for l in log)                                  # the group is closed even when a reduce failed
    assert "DONE" in out.stdout
