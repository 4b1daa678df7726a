cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "coop" 2>&1 | tail -3 | tee $O/r05_coop_ops2.txt
ROUNDS=2 timeout 900 bash tools/ab.sh decode env BEVGEN_QKV_COOP=0,1 2100 "f16:f16 f32:f32" 2>&1 | tee $O/r05_ab_qkv_coop2.txt
timeout 300 python tools/decode_trace.py 16 1044 f16 1 f16 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/r05_decode_trace_coop2.txt
