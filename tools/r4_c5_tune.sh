#!/bin/bash
# Round 4, GPU call 2: VALU / packed-f32 microbenchmark with measured clock; per-ray step counts and phase clocks of the quad traversal;
# node-phase / refill thresholds re-swept for the four-slot records; per-kernel times.
#   gpurun --timeout 900 -- 'bash tools/r4_c5_tune.sh'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 120 tools/ubench_valu > gpurun_out/r04_ubench_valu.txt 2>&1; head -40 gpurun_out/r04_ubench_valu.txt
{
echo "== thresholds"; bash tools/c5_libs.sh 32 libtrayhip.so libtrayhip_ns4.so libtrayhip_ns16.so libtrayhip_nm8.so libtrayhip_nm28.so libtrayhip_rf16.so libtrayhip_rf40.so libtrayhip.so
echo "== steps per ray"; TRAYHIP_STATS=1 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_stats.so LABEL=stats timeout 200 python /tmp/c5_run.py 2>&1 | grep "stage\|full detail"
echo "== phase clocks"; TRAYHIP_STATS=1 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_clocks.so LABEL=clocks timeout 200 python /tmp/c5_run.py 2>&1 | grep "wave cycles\|full detail"
echo "== old phase clocks / steps: see profiles/r03_c5_traversal_steps_ab.txt"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4_kstats -- python /tmp/c5_run.py > $ROOT/gpurun_out/r4_kstats.log 2>&1
cd $ROOT; python tools/kstats_table.py gpurun_out/r4_kstats 2>&1 | head -24
} 2>&1 | tee gpurun_out/r4_c5_tune.log
