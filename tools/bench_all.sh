#!/bin/bash
# The four bench lines of the round (C2 default, C3-C5 by --workload), each one JSON line under gpurun_out/bench_<tag>_*.json
#   gpurun --timeout 900 -- 'bash tools/bench_all.sh r02g'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; TAG=${1:-x}; mkdir -p gpurun_out
timeout 300 python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
for wl in smallpt dragon tr15_like; do
  timeout 400 python bench.py --workload $wl --steps 2 --warmup 1 > gpurun_out/bench_${TAG}_$wl.json 2> gpurun_out/bench_${TAG}_$wl.err
done
python - <<PY
import json, glob
for p in sorted(glob.glob("gpurun_out/bench_${TAG}*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(p.split("/")[-1], d["value"], d["unit"], "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"), "ms/step", d["ms_per_step"])
    except Exception as e:
        print(p, "unreadable", e)
PY
