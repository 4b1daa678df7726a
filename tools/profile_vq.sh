cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/cmd_prof
rocprofv3 --kernel-trace -d $R/gpurun_out/cmd_prof -o t -- python $R/tools/vq_probe.py 96 f16x3 5 > $R/gpurun_out/vqprof.log 2>&1
DB=$(find $R/gpurun_out/cmd_prof -name "*.db" | head -1)
python $R/tools/rocpd_by_grid.py $DB 40 > $R/gpurun_out/r05_vq_by_grid.txt
cat $R/gpurun_out/r05_vq_by_grid.txt
rm -rf $R/gpurun_out/cmd_prof
