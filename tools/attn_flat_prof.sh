cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for b in 1 2 3; do for v in 0 -1; do
  if [ $v = 0 ]; then export BEVGEN_ATTN_FLAT=0; else unset BEVGEN_ATTN_FLAT; fi
  rm -rf $R/gpurun_out/cmd_prof
  BEVGEN_BENCH_NO_PMC=1 rocprofv3 --kernel-trace -d $R/gpurun_out/cmd_prof -o t -- python $R/bench.py --steps 1 --warmup 1 --batch $b --no-decode-leg --no-extra-legs --no-cpu-baseline --no-exact-leg > /dev/null 2>&1
  DB=$(find $R/gpurun_out/cmd_prof -name "*.db" | head -1)
  echo "== batch $b flat=$v"; python $R/tools/rocpd_by_grid.py $DB 30 | grep -E "attention" | cut -c1-200
done; done 2>&1 | tee $R/gpurun_out/r05_attn_flat_kernels.txt
rm -rf $R/gpurun_out/cmd_prof
