#!/bin/bash
# SQ instruction-mix / stall counters of attention_split_kernel at the bench shape (three --pmc passes, kernel trace only).  usage: bash tools/pmc_attn_sq.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
OUT=$R/gpurun_out/${TAG}_attn_split_sq.txt
: > $OUT
pass() {
  rm -rf $R/gpurun_out/pmc_attn
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_attn -o a --output-format csv -- python $R/tools/attn_probe.py 4 > $R/gpurun_out/pmc_attn.log 2>&1
  python - "$@" <<PY >> $OUT
import csv, glob, collections, sys
f = glob.glob("$R/gpurun_out/pmc_attn/**/*counter_collection.csv", recursive=True)
t = glob.glob("$R/gpurun_out/pmc_attn/**/*kernel_trace.csv", recursive=True)
if not f: print("pass failed:", sys.argv[1:]); sys.exit(0)
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0]))}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "attention_split_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in acc.items():
    vals = [sum(v) for v in d.values()]
    ns = [dur[i] for i in d]
    print(f"{c:28s} per launch {sum(vals)/len(vals):16.0f}   launches {len(vals)}   avg kernel {sum(ns)/len(ns)/1e3:8.1f} us (serialised under counter collection)")
PY
}
pass SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES
pass SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
rm -rf $R/gpurun_out/pmc_attn
cat $OUT
