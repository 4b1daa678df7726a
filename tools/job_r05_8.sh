cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_ops_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "vq or groupnorm or net2net" 2>&1 | tail -3 | tee $O/r05_gn_tests.txt
: > $O/r05_ab_gn_apply.txt
for i in 1 2; do
  BEVGEN_LIB_PATH=$GRAFT_REPO_ROOT/.ab/libbase.so python tools/vq_probe.py 96 2>/dev/null | sed 's/^/base /' | tee -a $O/r05_ab_gn_apply.txt
  python tools/vq_probe.py 96 2>/dev/null | sed 's/^/new  /' | tee -a $O/r05_ab_gn_apply.txt
done
