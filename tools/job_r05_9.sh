cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_ops_gpu.py tests/test_dropin_gpu.py tests/test_generate_gpu.py -x -q -m gpu -k "vq or groupnorm or net2net or conv or generate" 2>&1 | tail -3 | tee $O/r05_gn_epi_tests.txt
: > $O/r05_ab_vq_up_planes.txt
for i in 1 2; do for v in 0 1; do
  BEVGEN_VQ_UP_PLANES=$v python tools/vq_probe.py 96 2>/dev/null | sed "s/^/BEVGEN_VQ_UP_PLANES=$v /" | tee -a $O/r05_ab_vq_up_planes.txt
done; done
