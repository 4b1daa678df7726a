cd $GRAFT_REPO_ROOT; O=gpurun_out
: > $O/r05_auto_threshold.txt
for B in 1 2 3 4 6 8; do
  python tools/decode_probe.py $B 2100 fused,split f32 1 f32 2>/dev/null | grep "ms/step" | tee -a $O/r05_auto_threshold.txt
  python tools/decode_probe.py $B 2100 fused,split f16 1 f16 2>/dev/null | grep "ms/step" | tee -a $O/r05_auto_threshold.txt
done
