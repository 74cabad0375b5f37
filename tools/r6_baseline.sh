#!/bin/bash
# Round 6, first GPU call: the split build (ten kernel translation units) through the quick check, the GPU suite and the driver's bench command.
TAG=${1:-a}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
P=r06_${TAG}
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -4
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^Frame [0-9]*: rendering took" > gpurun_out/${P}_gpu_suite.log; tail -3 gpurun_out/${P}_gpu_suite.log
timeout 900 python bench.py > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err; cut -c1-600 gpurun_out/${P}_bench.json
