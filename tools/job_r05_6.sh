cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "vq or groupnorm or conv" 2>&1 | tail -3 | tee $O/r05_vq_tests.txt
for i in 1 2; do
  BEVGEN_LIB_PATH=$GRAFT_REPO_ROOT/.ab/libhead.so python tools/vq_probe.py 96 2>/dev/null | sed 's/^/head /' | tee -a $O/r05_ab_vq.txt
  python tools/vq_probe.py 96 2>/dev/null | sed 's/^/new  /' | tee -a $O/r05_ab_vq.txt
done
for c in 6 12 24; do BEVGEN_VQ_CHUNK=$c python tools/vq_probe.py 96 2>/dev/null | sed 's/^/new  /' | tee -a $O/r05_ab_vq.txt; done
bash tools/profile_b1.sh r05 1 2>&1 | tail -45 | tee $O/r05_b1_profile.txt
