cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "mlp_fused or ln_gemm or ar_attn" 2>&1 | tail -5 | tee $O/r05_mlpf_ops.txt
timeout 1500 python -m pytest tests/test_models_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "route_a or ar_ or gpt or config4 or config5" 2>&1 | tail -6 | tee $O/r05_mlpf_models.txt
ROUNDS=2 timeout 900 bash tools/ab.sh decode env BEVGEN_MLP_FUSE=0,1 2>&1 | tee $O/r05_ab_mlp_fuse.txt
ROUNDS=2 timeout 600 bash tools/ab.sh m env BEVGEN_GEMM_ROWSPLIT_SIDE=0,1 --batch 1 2>&1 | tee $O/r05_ab_rowsplit_side_b1.txt
ROUNDS=1 timeout 600 bash tools/ab.sh m env BEVGEN_GEMM_ROWSPLIT_SIDE=0,2 2>&1 | tee $O/r05_ab_rowsplit_side_b16.txt
