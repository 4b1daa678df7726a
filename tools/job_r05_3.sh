cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "coop or mlp_fused or ar_attn" 2>&1 | tail -5 | tee $O/r05_coop_ops.txt
timeout 1500 python -m pytest tests/test_models_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "route_a or ar_ or gpt or config4 or config5" 2>&1 | tail -6 | tee $O/r05_coop_models.txt
ROUNDS=2 timeout 900 bash tools/ab.sh decode env BEVGEN_QKV_COOP=0,1 2>&1 | tee $O/r05_ab_qkv_coop.txt
timeout 300 python tools/decode_trace.py 16 1044 f16 1 f16 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/r05_decode_trace_coop.txt
BEVGEN_QKV_COOP=0 timeout 300 python tools/decode_trace.py 16 1044 f16 1 f16 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/r05_decode_trace_nocoop.txt
