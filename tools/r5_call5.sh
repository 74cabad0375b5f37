#!/bin/bash
# round 5, GPU call 5: per-kernel counters of the C5 stand-in in the benched schedule (current build)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 700 python tools/pmc_workloads.py r05_c tr15_like:128 2>&1 | tail -2 | cut -c1-600
