#!/bin/bash
# Round 6 measurement cycle for the build in tray_rust_amd/libtrayhip.so:  gpurun --timeout 2700 -- 'bash tools/r6_measure.sh <tag>'
#   GPU suite -> counter passes of all four workloads in the benched schedules (tr15_like at 128 spp, frame 64: 4 slices, 33 M slots, one view) +
#   static registers (so that the bench line finds hash-matched `latest` files) -> the driver's bench command -> rocprofv3 kernel stats of that
#   command -> per-kernel times and the per-kernel traffic table of C5 -> configs[4] per frame -> rates of the side paths -> moving_box counters.
# Everything lands in gpurun_out/ (incl. the two *_latest.json); copy what is to be judged into profiles/.
TAG=${1:-a}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
P=r06_${TAG}
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^Frame [0-9]*: rendering took" > gpurun_out/${P}_gpu_suite.log; tail -3 gpurun_out/${P}_gpu_suite.log
timeout 1200 python tools/pmc_workloads.py ${P} 2>&1 | tail -6 | cut -c1-300
cp gpurun_out/summary_${P}/pmc_latest.json profiles/pmc_latest.json
python tools/c5_traffic_table.py gpurun_out/summary_${P} > gpurun_out/${P}_c5_traffic_by_kernel.txt 2>&1; head -8 gpurun_out/${P}_c5_traffic_by_kernel.txt | cut -c1-200
bash tools/static_registers.sh > gpurun_out/${P}_static_registers.txt 2>&1; cp profiles/static_registers_latest.json gpurun_out/${P}_static_registers_latest.json
timeout 900 python bench.py > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err; cut -c1-400 gpurun_out/${P}_bench.json
bash tools/kernel_stats_bench.sh ${P} --no-other-workloads
cd /tmp; export TMPDIR=/tmp
C5_FRAME=64 bash $ROOT/tools/c5_libs.sh 256 libtrayhip.so 2>&1 | tail -1
C5_FRAME=64 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${P}_c5_kstats -- python /tmp/c5_run.py > /dev/null 2>&1
cd $ROOT; python tools/kstats_table.py gpurun_out/${P}_c5_kstats > gpurun_out/${P}_c5_kernel_times.txt 2>&1; head -8 gpurun_out/${P}_c5_kernel_times.txt
python tools/r5_c5_frames.py 512; cp gpurun_out/r05_c5_frames.txt gpurun_out/${P}_c5_frames.txt
python tools/r4_side_paths.py > gpurun_out/${P}_side_paths.txt 2>&1; cat gpurun_out/${P}_side_paths.txt
PMC_SETS=1 timeout 300 python tools/pmc_ab.py moving_box:32 libtrayhip.so 2>&1 | tail -1; PMC_SETS=1 timeout 300 python tools/pmc_ab.py cornell_box:64 libtrayhip.so 2>&1 | tail -1; cp gpurun_out/pmc_ab.txt gpurun_out/${P}_pmc_moving_box.txt
