#!/bin/bash
# Round 4 measurement cycle for the build in tray_rust_amd/libtrayhip.so:  gpurun --timeout 2400 -- 'bash tools/r4_measure.sh <tag>'
#   GPU suite -> bench line (the driver's command) -> rocprofv3 kernel stats of that command -> counter passes of all four workloads
#   -> per-kernel times of the C5 stand-in.  Everything lands in gpurun_out/; copy what is to be judged into profiles/.
TAG=${1:-a}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 1300 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/r04_${TAG}_gpu_suite.log; tail -3 gpurun_out/r04_${TAG}_gpu_suite.log
timeout 600 python bench.py > gpurun_out/r04_${TAG}_bench.json 2> gpurun_out/r04_${TAG}_bench.err; cut -c1-400 gpurun_out/r04_${TAG}_bench.json
bash tools/kernel_stats_bench.sh r04_${TAG} --no-other-workloads
timeout 900 python tools/pmc_workloads.py r04_${TAG} 2>&1 | tail -8
bash tools/c5_libs.sh 32 libtrayhip.so 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_${TAG}_c5_kstats -- python /tmp/c5_run.py > /dev/null 2>&1
cd $ROOT; python tools/kstats_table.py gpurun_out/r04_${TAG}_c5_kstats > gpurun_out/r04_${TAG}_c5_kernel_times.txt 2>&1; head -8 gpurun_out/r04_${TAG}_c5_kernel_times.txt
