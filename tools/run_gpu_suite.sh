#!/bin/bash
# full GPU suite + smoke; tail of the result into gpurun_out/gpu_suite.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/gpu_suite.txt
