#!/bin/bash
# Round 4 measurement cycle for the build in tray_rust_amd/libtrayhip.so:  gpurun --timeout 2400 -- 'bash tools/r4_measure.sh <tag>'
#   GPU suite -> counter passes of all four workloads + static registers (so that the bench line finds hash-matched `latest` files) -> bench
#   line (the driver's command) -> rocprofv3 kernel stats of that command -> per-kernel times of the C5 stand-in -> rates of the side paths.
# Everything lands in gpurun_out/ (incl. the two *_latest.json); copy what is to be judged into profiles/.
TAG=${1:-a}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 1300 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/r04_${TAG}_gpu_suite.log; tail -3 gpurun_out/r04_${TAG}_gpu_suite.log
timeout 900 python tools/pmc_workloads.py r04_${TAG} 2>&1 | tail -6 | cut -c1-400
cp gpurun_out/summary_r04_${TAG}/pmc_latest.json profiles/pmc_latest.json
bash tools/static_registers.sh > gpurun_out/r04_${TAG}_static_registers.txt 2>&1; cp profiles/static_registers_latest.json gpurun_out/r04_${TAG}_static_registers_latest.json
timeout 600 python bench.py > gpurun_out/r04_${TAG}_bench.json 2> gpurun_out/r04_${TAG}_bench.err; cut -c1-400 gpurun_out/r04_${TAG}_bench.json
bash tools/kernel_stats_bench.sh r04_${TAG} --no-other-workloads
bash tools/c5_libs.sh 32 libtrayhip.so 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_${TAG}_c5_kstats -- python /tmp/c5_run.py > /dev/null 2>&1
cd $ROOT; python tools/kstats_table.py gpurun_out/r04_${TAG}_c5_kstats > gpurun_out/r04_${TAG}_c5_kernel_times.txt 2>&1; head -8 gpurun_out/r04_${TAG}_c5_kernel_times.txt
python tools/r4_side_paths.py > gpurun_out/r04_${TAG}_side_paths.txt 2>&1; cat gpurun_out/r04_${TAG}_side_paths.txt
