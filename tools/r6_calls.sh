#!/bin/bash
# The GPU calls of round 6, one script:  gpurun --timeout <s> -- 'bash tools/r6_calls.sh <call>'   (<call> = baseline | 2 .. 6 | 8 .. 13; tools/README.md says what each one
# measured and where its record lies under profiles/). Variant libraries (libtrayhip_<name>.so) are built beforehand with tools/variant.sh and travel with the push.
# tools/r6_measure.sh <tag> is the round's measurement cycle.
CALL=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
case "$CALL" in
baseline)
  # Round 6, first GPU call: the split build (ten kernel translation units) through the quick check, the GPU suite and the driver's bench command.
  TAG=${1:-a}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; mkdir -p gpurun_out
  P=r06_${TAG}
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -4
  timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^Frame [0-9]*: rendering took" > gpurun_out/${P}_gpu_suite.log; tail -3 gpurun_out/${P}_gpu_suite.log
  timeout 900 python bench.py > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err; cut -c1-600 gpurun_out/${P}_bench.json

  ;;
2)
  # Round 6, GPU call 2: (1) ray binning before the traversal stages (TRAYHIP_WF_BIN: 0 off, 1 stage A, 2 stage B, 3 both; the bin3 build has 8 cells
  # per axis instead of 4) on C5's frames 64 and 127 at 128 spp; (2) the SLP vectoriser / strict-aliasing builds on the cut-down tile workloads
  # (RMSE against the default build's render + rate); (3) C5's film against the oracle with binning on (the GPU suite's full-size C5 test).
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -2
  {
  for fr in 64 127; do
    C5_FRAME=$fr bash tools/c5_env.sh 128 "bin0_f$fr=TRAYHIP_WF_BIN=0" "bin1_f$fr=TRAYHIP_WF_BIN=1" "bin3_f$fr=TRAYHIP_WF_BIN=3" "bin0_f$fr=TRAYHIP_WF_BIN=0" "bin3_f$fr=TRAYHIP_WF_BIN=3"
    [ -f tray_rust_amd/libtrayhip_bin3.so ] && C5_FRAME=$fr bash tools/c5_env.sh 128 "cells8_bin3_f$fr=TRAYHIP_WF_BIN=3 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_bin3.so" "cells8_bin1_f$fr=TRAYHIP_WF_BIN=1 TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_bin3.so"
  done
  } 2>&1 | tee gpurun_out/r06_c5_binning_ab.txt
  LIBS="libtrayhip.so"; for v in slp slpnsa nsa; do [ -f tray_rust_amd/libtrayhip_$v.so ] && LIBS="$LIBS libtrayhip_$v.so"; done
  bash tools/ab.sh r06_slp $LIBS $LIBS 2>&1 | tail -30
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "c5" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^Frame [0-9]*: rendering took" | tail -15 | tee gpurun_out/r06_c5_fullsize_binned.txt

  ;;
3)
  # Round 6, GPU call 3: (1) C5 with the continuation rays queued by the shading kernels (k_wf_advance no longer re-reads them) against the previous
  # schedule (libtrayhip_bin3.so with TRAYHIP_WF_BIN=0 = the committed build before the change), frames 64 / 127 at 128 spp, + per-kernel times of both;
  # (2) the tile kernel's OWN per-sample radiance against the oracle for the default flags and for the build with the SLP vectoriser on
  # (-DTR_SAMPLE_DUMP builds, tools/tile_sample_dump.py); packed f32 against scalar on subnormals (tools/pk_denorm_check); (3) where the dragon's
  # wave cycles go (-DTR_STAGE_CLOCKS build, full-size mesh); (4) one GPU's share of a frame at N = 2 / 4 / 8 (tools/eighth_rate.py).
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

  ;;
4)
  # Round 6, GPU call 4: (1) the flat loop's per-lane pass restricted to flat rectangles / disks against the uniform loop (nopend build); (2) the SLP
  # vectoriser's wrong samples, narrowed: final throughput against radiance, with the cooperative small-mesh test off (TRAYHIP_NO_COOP), without
  # horizontal reductions (-mllvm -slp-vectorize-hor=false); (3) every shard of 8 for chunk sizes 16 / 4 / 1 of the round-robin deal.
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

  ;;
5)
  # Round 6, GPU call 5: progressive tile slices (whole frame and the 8 shards), the SLP reduction program, the SLP build's samples against the default's
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
  timeout 900 python tools/tile_slices_ab.py dragon cornell_box 2>&1 | grep -v "^Frame" | tee gpurun_out/r06_tile_slices_progressive_ab.txt
  {
  for v in on off; do /opt/rocm/bin/hipcc -O3 -ffp-contract=off $([ $v = off ] && echo -fno-slp-vectorize) --offload-arch=gfx950 tools/experiments/slp_rect_test.hip -o /tmp/slp_rect_$v 2>/dev/null; echo "== slp_rect_test, SLP $v"; /tmp/slp_rect_$v; done
  for sc in cornell_box dragon; do
    DUMP_SAVE=/tmp/dump_$sc.npy TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_dump.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 default_flags 2>&1 | grep -v "^Frame" | head -1
    DUMP_COMPARE=/tmp/dump_$sc.npy TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_slpdump.so timeout 300 python tools/tile_sample_dump.py $sc 160x120x32 slp_on 2>&1 | grep -v "^Frame" | grep -v "   px"
  done
  } 2>&1 | tee gpurun_out/r06_slp_reduction.txt

  ;;
6)
  # Round 6, GPU call 6: k_wf_begin fetching its records together with the flags word (against the build of measurement cycle a), the persistent
  # k_sampler_pass (rates of the side paths + their GPU parity tests)
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
  {
  for fr in 64 127; do C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_prev.so libtrayhip.so libtrayhip_prev.so libtrayhip.so; done
  } 2>&1 | tee gpurun_out/r06_c5_begin_prefetch_ab.txt
  python tools/r4_side_paths.py 2>&1 | grep -v "^Frame" | tee gpurun_out/r06_side_paths_persistent.txt
  TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_prev.so python tools/r4_side_paths.py 2>&1 | grep -v "^Frame" | grep "Uniform\|Adaptive\|flag" | sed 's/^/previous build: /' | tee -a gpurun_out/r06_side_paths_persistent.txt
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "sampler or adaptive or uniform or rank_4 or animated_mesh or whitted or c2 or c3 or c4" 2>&1 | tail -5

  ;;
8)
  # Round 6, GPU call 8 (moving scenes on the tile kernel): the per-lane pass of the flat loop taking the MOVING spheres / rectangles / disks
  # (FlatInst::lane_pass = 2), the first pending mover's transform requested inside the uniform loop (prefetch), the fill of the cache columns from the
  # frame's table dealt out to the whole wave (coop), and the two ceilings: cache columns that stay in L2 (colmask), columns + table in L2 (colkmask; WRONG pictures both).
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
  AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r06_moving_box libtrayhip_nomov.so libtrayhip.so libtrayhip_prefetch.so libtrayhip_coop.so libtrayhip_cooppre.so libtrayhip_colmask.so libtrayhip_colkmask.so \
      libtrayhip_nomov.so libtrayhip.so libtrayhip_prefetch.so libtrayhip_coop.so libtrayhip_cooppre.so
  for lib in libtrayhip.so libtrayhip_cooppre.so; do
    echo "== parity of moving scenes, $lib"
    TRAYHIP_LIB=$ROOT/tray_rust_amd/$lib timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "moving or anim" 2>&1 | tail -3
  done 2>&1 | tee gpurun_out/r06_moving_box_parity.txt

  ;;
9)
  # Round 6, GPU call 9: the cooperative table fill handing the moving CAMERA's record over as well (one trip to the table per regeneration step instead of two;
  # camoff_ = fill only), against the build before (nomov_) and the all-in-L2 ceiling (colkmask_, WRONG pictures); table-mode parity tests on the new default.
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
  LIBS=${CALL9_LIBS:-"libtrayhip_nomov.so libtrayhip_camoff.so libtrayhip.so libtrayhip_colkmask.so"}
  AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r06_moving_box_cam $LIBS $LIBS
  {
  echo "== parity of moving scenes and of table mode, libtrayhip.so"
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "moving or anim or transform_table or whitted or sampler" 2>&1 | tail -3
  } 2>&1 | tee gpurun_out/r06_moving_box_cam_parity.txt

  ;;
10)
  # Round 6, GPU call 10: the transform table's records at 128 bytes (one cache line each, TR_XF_REC 32) against 112 (camoff_: the cooperative fill at 112),
  # two records per lane and trip in the fill (two_), the camera's record requested before the fill (early_, earlytwo_); moving_box on the tile kernel and the C5 stand-in (whose stage kernels gather the records directly).
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
  LIBS="libtrayhip_camoff.so libtrayhip.so libtrayhip_two.so libtrayhip_early.so libtrayhip_earlytwo.so"
  AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r06_moving_box_rec128 $LIBS $LIBS
  {
  for fr in 64 127; do C5_FRAME=$fr bash tools/c5_libs.sh 128 libtrayhip_camoff.so libtrayhip.so libtrayhip_camoff.so libtrayhip.so; done
  } 2>&1 | tee gpurun_out/r06_c5_rec128_ab.txt
  {
  echo "== parity of moving scenes and of table mode, libtrayhip.so"
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "moving or anim or transform_table or c5 or tr15" 2>&1 | tail -3
  echo "== libtrayhip_two.so"
  TRAYHIP_LIB=$ROOT/tray_rust_amd/libtrayhip_two.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "moving or transform_table" 2>&1 | tail -3
  } 2>&1 | tee gpurun_out/r06_moving_box_rec128_parity.txt

  ;;
11)
  # Round 6, GPU call 11: 128-byte table records on the tile kernel's moving scenes (camoff_ = the same build at 112 bytes), then measurement cycle d of the final build
  AB_WORKLOADS="moving_box:32 moving_box:128" bash tools/ab.sh r06_moving_box_rec128_only libtrayhip_camoff.so libtrayhip.so libtrayhip_camoff.so libtrayhip.so
  bash tools/r6_measure.sh d

  ;;
12)
  # Round 6, GPU call 12 (C4): the workgroup gathers the rays that enter the dragon's mesh and walks the tree with full waves (TRAYHIP_WG_COMPACT=1;
  # dev_geom.h: DevScene::wg_compact). prev_ = the build before the code existed, _ = with the code, switched off / on. Full-size mesh (871 200 triangles).
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
  export AB_DIR=/tmp/mini_full MINI_DRAGON_GRID=660
  WL="cornell_box:64 smallpt:64 dragon:32 dragon:256"
  {
  echo "== switched off"
  AB_WORKLOADS="$WL" bash tools/ab.sh r06_wg_off libtrayhip_prev.so libtrayhip.so libtrayhip_prev.so libtrayhip.so
  echo "== TRAYHIP_WG_COMPACT=1 (only the dragon has a large mesh)"
  TRAYHIP_WG_COMPACT=1 AB_WORKLOADS="dragon:32 dragon:256 cornell_box:64" bash tools/ab.sh r06_wg_on libtrayhip.so libtrayhip_prev.so libtrayhip.so
  } 2>&1 | tee gpurun_out/r06_c4_wg_compact_ab.txt
  TRAYHIP_WG_COMPACT=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "dragon or c4 or mesh" 2>&1 | grep "passed\|failed" | tail -2 | tee -a gpurun_out/r06_c4_wg_compact_ab.txt

  ;;
13)
  # Round 6, GPU call 13: the AMDGPU machine scheduler's other strategies (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause / iterative-minreg / iterative-ilp)
  # against the default (max occupancy) on the four workloads at full size. Scheduling moves no f32 operation: the bits are the same (RMSE column, parity tests below).
  timeout 300 python tools/quick_gpu_check.py 2>&1 | tail -1
  export AB_DIR=/tmp/mini_full MINI_DRAGON_GRID=660
  LIBS=${CALL13_LIBS:-"libtrayhip.so libtrayhip_ilp.so libtrayhip_mclause.so libtrayhip_iminreg.so libtrayhip_iilp.so"}
  {
  AB_WORKLOADS="cornell_box:64 smallpt:64 dragon:32 moving_box:32" bash tools/ab.sh r06_sched $LIBS $LIBS
  C5_FRAME=64 bash tools/c5_libs.sh 128 $LIBS $LIBS
  } 2>&1 | grep -v "^Frame" | tee gpurun_out/r06_sched_strategy_ab.txt

  ;;
*) echo "usage: tools/r6_calls.sh baseline|2|3|4|5|6|8|9|10|11|12|13"; exit 2 ;;
esac
