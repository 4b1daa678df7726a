cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 2 0; do
export BEVGEN_GLDS_CONFIG=$c
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq_$c -o sq --output-format csv -- python $R/tools/gemm_probe.py 3 3 > $R/gpurun_out/pmc_sq_$c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $R/gpurun_out/pmc_tcc_$c -o tcc --output-format csv -- python $R/tools/gemm_probe.py 3 3 > $R/gpurun_out/pmc_tcc_$c.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/pmc_tcc2_$c -o tcc --output-format csv -- python $R/tools/gemm_probe.py 3 3 > $R/gpurun_out/pmc_tcc2_$c.log 2>&1
done
ls -R $R/gpurun_out | head -50
