cd $GRAFT_REPO_ROOT
bash tools/c5_libs.sh 32 libtrayhip.so > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_kstats3 -- python /tmp/c5_run.py > $GRAFT_REPO_ROOT/gpurun_out/r4_kstats3.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/kstats_table.py gpurun_out/r4_kstats3 2>&1 | head -16
