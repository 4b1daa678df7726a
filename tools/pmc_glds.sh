#!/bin/bash
# SQ-level PMC counters of the LDS-DMA GEMM over tools/gemm_probe.py (one --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- python $R/tools/gemm_probe.py 3 3 24576,1024,4096 > $R/gpurun_out/pmc_sq.log 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$R/gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "glds" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
wc = m["SQ_WAVE_CYCLES"]
print({k: f"{v:.4g}" for k, v in m.items()})
print("wait_any %.3f  wait_inst %.3f (lds %.3f)  active %.3f | mfma busy / (GUI_ACTIVE/8 * 1024 SIMDs) = %.3f" % (m["SQ_WAIT_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc, m["SQ_WAIT_INST_LDS"] / wc, m["SQ_ACTIVE_INST_ANY"] / wc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
rm -rf $R/gpurun_out/pmc_sq
