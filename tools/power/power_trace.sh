#!/bin/bash
# Socket power + clocks sampled while (a) the register-only MFMA loop and (b) the LDS-DMA split-precision GEMM probe run.
# usage on the GPU box: bash tools/power/power_trace.sh <tag>     -> gpurun_out/<tag>_power_{idle,mfma,gemm}.csv + logs
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
OUT=$R/gpurun_out
mkdir -p $OUT
( timeout 10 amd-smi metric -g 0 --power --clock 2>&1 | head -80 ) > $OUT/${TAG}_amdsmi_sample.txt
( timeout 10 amd-smi static -g 0 --limit 2>&1 | head -40 ) >> $OUT/${TAG}_amdsmi_sample.txt
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $OUT/${TAG}_hwmon_ls.txt 2>&1
python $R/tools/power/sampler.py $OUT/${TAG}_power_idle.csv & sleep 2; touch $OUT/${TAG}_power_idle.csv.stop; wait
python $R/tools/power/sampler.py $OUT/${TAG}_power_mfma.csv &
sleep 1
timeout 60 $R/tools/power/mfma_loop 6 > $OUT/${TAG}_mfma_loop.log 2>&1
sleep 1; touch $OUT/${TAG}_power_mfma.csv.stop; wait
python $R/tools/power/sampler.py $OUT/${TAG}_power_mfma_fresh.csv &
sleep 1
timeout 60 $R/tools/power/mfma_loop 6 4 2 16 > $OUT/${TAG}_mfma_loop_fresh.log 2>&1
sleep 1; touch $OUT/${TAG}_power_mfma_fresh.csv.stop; wait
python $R/tools/power/sampler.py $OUT/${TAG}_power_gemm.csv &
sleep 1
timeout 120 python $R/tools/gemm_probe.py 3 25000 24576,1024,1024 > $OUT/${TAG}_gemm_probe.log 2>&1
sleep 1; touch $OUT/${TAG}_power_gemm.csv.stop; wait
cat $OUT/${TAG}_mfma_loop.log; cat $OUT/${TAG}_mfma_loop_fresh.log; cat $OUT/${TAG}_gemm_probe.log
wc -l $OUT/${TAG}_power_*.csv
python $R/tools/power/summarise.py $OUT/${TAG}_power_idle.csv $OUT/${TAG}_power_mfma.csv $OUT/${TAG}_power_mfma_fresh.csv $OUT/${TAG}_power_gemm.csv
