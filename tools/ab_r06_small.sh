#!/bin/bash
# small-batch A/Bs on one box: the row-split rest's block shape (q | k | v of two scenes), key-split self-attention at two / four scenes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "row_split or independent_of_batch" > $O/ab_small_tests.txt 2>&1; tail -3 $O/ab_small_tests.txt
: > $O/ab_small.txt
for i in 1 2 3; do for v in 4 0; do
  echo -n "ROWSPLIT_BOT_WM=$v " >> $O/ab_small.txt; BEVGEN_ROWSPLIT_BOT_WM=$v python tools/ab_ln_fold.py 2 5 2>/dev/null | tail -1 >> $O/ab_small.txt
done; done
for i in 1 2; do for b in 2 4; do for v in 0 2 4; do
  echo -n "ATTN_KSPLIT=$v " >> $O/ab_small.txt; BEVGEN_ATTN_KSPLIT=$v python tools/ab_ln_fold.py $b 5 2>/dev/null | tail -1 >> $O/ab_small.txt
done; done; done
cat $O/ab_small.txt
