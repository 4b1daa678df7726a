cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "mlp_fused" 2>&1 | tail -2 | tee $O/r05_mlpf_barrier_ops.txt
ROUNDS=3 timeout 1200 bash tools/ab.sh decode lib "base nosleep" 2100 "f16:f16" 2>&1 | tee $O/r05_ab_mlpf_barrier.txt
