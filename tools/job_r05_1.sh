cd $GRAFT_REPO_ROOT; O=gpurun_out
./tools/gridbar/xcd_xchg_probe 200 2>&1 | tee $O/r05_xcd_xchg_probe.txt
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "route_m" 2>&1 | tail -3 | tee $O/r05_routem_tests.txt
ROUNDS=2 bash tools/ab.sh m env BEVGEN_QKV_MERGE=0,1 2>&1 | tee $O/r05_ab_qkv_merge.txt
ROUNDS=2 bash tools/ab.sh m env BEVGEN_QKV_MERGE=0,1 --batch 1 2>&1 | tee $O/r05_ab_qkv_merge_b1.txt
ROUNDS=1 bash tools/ab.sh m env BEVGEN_GEMM_BAND=2,4,8 2>&1 | tee $O/r05_ab_gemm_band.txt
( time python tools/ktrace_inrun.py traffic16 ) 2>&1 | tail -8 | tee $O/r05_decode_traffic.txt
