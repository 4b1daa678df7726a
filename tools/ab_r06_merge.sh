#!/bin/bash
# merged q | k | v projection beyond two scenes?  Measured on the binary BEFORE the default changed (tools/ab_ln_fold.py BATCH 4): $BEVGEN_QKV_MERGE = 1 meant "up to 3072 token rows", 2 "always".  Today: 3 = up to 3072 rows, 1 (default) = always
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab_merge.txt
for i in 1 2; do for b in ${BATCHES:-3 4 8}; do for v in 1 2; do
  echo -n "QKV_MERGE=$v " >> $O/ab_merge.txt; BEVGEN_QKV_MERGE=$v python tools/ab_ln_fold.py $b 4 2>/dev/null | tail -1 >> $O/ab_merge.txt
done; done; done
cat $O/ab_merge.txt
