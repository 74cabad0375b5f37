#!/bin/bash
# round 5, GPU call 8: the bench line and configs[4] per frame with the final host defaults (same device code as cycle b: hash-gated counters stay valid)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r05_c_bench.json 2> gpurun_out/r05_c_bench.err; cut -c1-300 gpurun_out/r05_c_bench.json
python tools/r5_c5_frames.py 512; cp gpurun_out/r05_c5_frames.txt gpurun_out/r05_c_c5_frames.txt
python tools/mini_ab.py prepare /tmp/mini_ab > /dev/null 2>&1; python tools/mini_ab.py run /tmp/mini_ab final cornell_box:64 smallpt:64 dragon:32 moving_box:32 moving_box:256 2>&1 | grep Msamples
