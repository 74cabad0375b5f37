#!/bin/bash
# Round 6, GPU call 3: (1) C5 with the continuation rays queued by the shading kernels (k_wf_advance no longer re-reads them) against the previous
# schedule (libtrayhip_bin3.so with TRAYHIP_WF_BIN=0 = the committed build before the change), frames 64 / 127 at 128 spp, + per-kernel times of both;
# (2) the tile kernel's OWN per-sample radiance against the oracle for the default flags and for the build with the SLP vectoriser on
# (-DTR_SAMPLE_DUMP builds, tools/tile_sample_dump.py); packed f32 against scalar on subnormals (tools/pk_denorm_check); (3) where the dragon's
# wave cycles go (-DTR_STAGE_CLOCKS build, full-size mesh); (4) one GPU's share of a frame at N = 2 / 4 / 8 (tools/eighth_rate.py).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -2
OLD="TRAYHIP_WF_BIN=0 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_bin3.so"
{
for fr in 64 127; do
  C5_FRAME=$fr bash tools/c5_env.sh 128 "old_f$fr=$OLD" "new_f$fr=TRAYHIP_WF_BIN=0" "old_f$fr=$OLD" "new_f$fr=TRAYHIP_WF_BIN=0"
done
} 2>&1 | tee gpurun_out/r06_c5_query_enqueues_ab.txt
cd /tmp; export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_bin3.so; else unset TRAYHIP_LIB; fi
  C5_FRAME=64 TRAYHIP_WF_BIN=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r06_c5_kstats_$v -- python /tmp/c5_run.py > /dev/null 2>&1
  python $ROOT/tools/kstats_table.py $ROOT/gpurun_out/r06_c5_kstats_$v > $ROOT/gpurun_out/r06_c5_kernel_times_$v.txt 2>&1; echo "== $v"; head -9 $ROOT/gpurun_out/r06_c5_kernel_times_$v.txt
done
unset TRAYHIP_LIB; cd $ROOT
{
echo "== packed f32 against scalar (tools/pk_denorm_check)"; timeout 60 tools/pk_denorm_check 2>&1 | tail -12
for sc in cornell_box smallpt dragon; do for v in dump slpdump; do TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_$v.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 $v 2>&1 | grep -v "^Frame"; done; done
} 2>&1 | tee gpurun_out/r06_slp_tile_samples.txt
{
[ -f /tmp/mini_full/cornell_box.json ] || MINI_DRAGON_GRID=660 MINI_TR15_DETAIL=0.15 python tools/mini_ab.py prepare /tmp/mini_full > /dev/null 2>&1
TRAYHIP_STATS=1 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_clk.so python tools/mini_ab.py run /tmp/mini_full clk dragon:32 cornell_box:64 2>&1 | grep -v "^Frame"
} 2>&1 | tee gpurun_out/r06_c4_stage_clocks.txt
timeout 600 python tools/eighth_rate.py 2 4 8 2>&1 | grep -v "^Frame" | tee gpurun_out/r06_eighth_rate.txt
# (5) the flat instance loop with the per-lane pass over simple instances (TR_FLAT_PEND, dev_geom.h: trace_flat) against the uniform loop (nopend build)
{
echo "== bit check, default build (per-lane pass)"; python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame" | head -4
AB_WORKLOADS="cornell_box:64 cornell_box:256 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r06_pend libtrayhip.so libtrayhip_nopend.so libtrayhip.so libtrayhip_nopend.so
rm -f gpurun_out/pmc_ab.txt; PMC_SETS=1 timeout 300 python tools/pmc_ab.py cornell_box:64 libtrayhip.so libtrayhip_nopend.so 2>&1 | tail -2; PMC_SETS=1 timeout 300 python tools/pmc_ab.py smallpt:64 libtrayhip.so libtrayhip_nopend.so 2>&1 | tail -2; cat gpurun_out/pmc_ab.txt
} 2>&1 | tee gpurun_out/r06_c2_per_lane_pass_ab.txt
