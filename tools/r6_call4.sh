#!/bin/bash
# Round 6, GPU call 4: (1) the flat loop's per-lane pass restricted to flat rectangles / disks against the uniform loop (nopend build); (2) the SLP
# vectoriser's wrong samples, narrowed: final throughput against radiance, with the cooperative small-mesh test off (TRAYHIP_NO_COOP), without
# horizontal reductions (-mllvm -slp-vectorize-hor=false); (3) every shard of 8 for chunk sizes 16 / 4 / 1 of the round-robin deal.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
{
echo "== bit check, default build"; python tools/r5_bitcheck.py /tmp/mini_ab 40000 2>&1 | grep -v "^Frame" | head -4
AB_WORKLOADS="cornell_box:64 cornell_box:256 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r06_pend2 libtrayhip.so libtrayhip_nopend.so libtrayhip.so libtrayhip_nopend.so
rm -f gpurun_out/pmc_ab.txt; for w in cornell_box:64 smallpt:64; do PMC_SETS=1 timeout 300 python tools/pmc_ab.py $w libtrayhip.so libtrayhip_nopend.so > /dev/null 2>&1; done; cat gpurun_out/pmc_ab.txt
} 2>&1 | tee gpurun_out/r06_c2_per_lane_pass_ab2.txt
{
for sc in cornell_box dragon; do
  DUMP_SAVE=/tmp/dump_$sc.npy TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_dump.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 default_flags 2>&1 | grep -v "^Frame" | head -2
  DUMP_COMPARE=/tmp/dump_$sc.npy TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_slpdump.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 slp_on 2>&1 | grep -v "^Frame" | head -3
  DUMP_COMPARE=/tmp/dump_$sc.npy TRAYHIP_NO_COOP=1 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_slpdump.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 slp_on_no_coop 2>&1 | grep -v "^Frame" | head -3
  DUMP_COMPARE=/tmp/dump_$sc.npy TRAYHIP_FEAT_ALL=1 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_slpdump.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 slp_on_every_lobe_kernel 2>&1 | grep -v "^Frame" | head -3
  DUMP_COMPARE=/tmp/dump_$sc.npy TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_slphor.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 slp_on_no_horizontal_reductions 2>&1 | grep -v "^Frame" | head -3
done
} 2>&1 | tee gpurun_out/r06_slp_narrowing.txt
EIGHTH_ALL_SHARDS=1 timeout 900 python tools/eighth_rate.py 8 2>&1 | grep -v "^Frame" | tee gpurun_out/r06_eighth_rate_all_shards.txt
