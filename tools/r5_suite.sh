#!/bin/bash
# round 5: the GPU suite of the build in tray_rust_amd/libtrayhip.so -> gpurun_out/r05_<tag>_gpu_suite.log
TAG=${1:-a}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^Frame [0-9]*: rendering took" > gpurun_out/r05_${TAG}_gpu_suite.log; tail -30 gpurun_out/r05_${TAG}_gpu_suite.log | cut -c1-400
