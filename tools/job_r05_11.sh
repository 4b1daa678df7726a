cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "mlp_fused" 2>&1 | tail -2 | tee $O/r05_mlpf64_ops.txt
timeout 1200 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "config5 or config4 or two_route_a" 2>&1 | tail -3 | tee $O/r05_mlpf64_models.txt
: > $O/r05_ab_mlpf64.txt
for i in 1 2; do for lib in base new; do
  if [ $lib = new ]; then unset BEVGEN_LIB_PATH; else export BEVGEN_LIB_PATH=$GRAFT_REPO_ROOT/.ab/libbase.so; fi
  python tools/decode_probe.py 16 2100 fused f16 1 f16 2>/dev/null | grep "ms/step" | sed "s/^/$lib /; s/tokens equal.*//" | tee -a $O/r05_ab_mlpf64.txt
  python tools/decode_probe.py 64 1000 fused f32 4 f32 2>/dev/null | grep "ms/step" | sed "s/^/$lib /; s/tokens equal.*//" | tee -a $O/r05_ab_mlpf64.txt
done; done
