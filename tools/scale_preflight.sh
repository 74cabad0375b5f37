#!/bin/bash
# Multi-GPU pre-flight: run the RCCL paths ONCE before a scaling run is the first thing that ever executes them.
#   tools/scale_preflight.sh            on the box that has the GPUs (an 8-GPU node; through gpurun on the 1-GPU pool it reports "1 device")
# With >= 2 visible devices: (1) bench.py's one-process-per-GPU path at N = 2 (torch.distributed "nccl" = RCCL, tray_render_shard_device per rank,
# ONE sum-reduce), two short steps; (2) the 2-rank GPU test of the same path against the oracle; (3) the in-library path
# (tray_multi_create / tray_render_frame_multi: one host thread, one stream per device, grouped ncclReduce) on all visible devices, checked
# against a single-device render. With one device: says so and exits 0 -- nothing to pre-flight.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=$(python - <<'PY'
import ctypes, sys
sys.path.insert(0, ".")
import tray_rust_amd as T
n = ctypes.c_int(0)
rc = T.lib().tray_device_count(ctypes.byref(n))
print(n.value if rc == 0 else 0)
PY
)
echo "scale_preflight: $N HIP device(s) visible"
if [ "${N:-0}" -lt 2 ]; then echo "scale_preflight: 1 device -- the RCCL paths need two; nothing to run (exit 0)"; exit 0; fi
set -e
PORT=$((29600 + $$ % 300))
echo "== bench.py --gpus 2 (one process per GPU, RCCL sum-reduce)"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 2 --steps 2 --warmup 1 --no-other-workloads --no-cpu-baseline | tee gpurun_out/preflight_bench_2gpu.json | cut -c1-400
echo "== the 2-rank GPU test (shards + reduce against the oracle)"
timeout 900 python -m pytest tests/test_multi_cpu.py -m gpu -q -k test_two_gpu_ranks_shard_and_rccl_reduce
echo "== the in-library path on all $N devices (tray_render_frame_multi) against one device"
timeout 900 python - <<PY
import sys
sys.path.insert(0, ".")
import numpy as np, tempfile, os
import tray_rust_amd as T
from tray_rust_amd import scenes
d = tempfile.mkdtemp()
scenes.write_assets(d, cornell=(320, 240, 64))
scene, rt, spp, fi = T.Scene.load_file(os.path.join(d, "cornell_box.json"))
cfg = T.Config(d, "cornell_box.json", spp, 1, fi, (0, 0))
hip = T.Hip(device=0, seed=5)
hip.render(scene, rt, cfg)
one = rt.get_renderf32().copy()
rt.clear()
tims, reduce_ms = hip.render_multi(scene, rt, cfg, list(range($N)))
many = rt.get_renderf32()
rgb = lambda i: i.reshape(-1, 4)[:, :3] / np.maximum(i.reshape(-1, 4)[:, 3:], 1e-20)
r = float(np.sqrt(np.mean((rgb(one) - rgb(many)) ** 2)))
print(f"tray_render_frame_multi on $N devices: samples per device {[int(t.samples) for t in tims]}, reduce {reduce_ms:.3f} ms, RMSE vs one device {r:.2e}")
assert sum(int(t.samples) for t in tims) == 320 * 240 * spp and r < 1e-5
PY
echo "scale_preflight: OK"
