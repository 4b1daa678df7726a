#!/bin/bash
# SQ-level PMC counters of the LDS-DMA GEMM on the three headline projection shapes of the bench workload (one --pmc pass, kernel trace only):
# effective clock (GRBM_GUI_ACTIVE / kernel time), matrix-pipe busy fraction, wave state split.  usage on the GPU box: bash tools/pmc_glds.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
rm -rf $R/gpurun_out/pmc_sq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- python $R/tools/gemm_probe.py 3 3 > $R/gpurun_out/pmc_sq.log 2>&1
python - <<PY > $R/gpurun_out/${TAG}_pmc_glds_sq.json
import csv, collections, glob, json
f = glob.glob("$R/gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
t = glob.glob("$R/gpurun_out/pmc_sq/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(t)):
    if "glds" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "glds" in r["Kernel_Name"] and r["Dispatch_Id"] in dur:
        agg[dur[r["Dispatch_Id"]][1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[dur[r["Dispatch_Id"]][1]]["_ns"].append(dur[r["Dispatch_Id"]][0])
out = {"note": "rocprofv3 --kernel-trace --pmc (one pass) over tools/gemm_probe.py 3 3 on MI355X: gemm_split_glds_kernel on the three projection shapes of the headline workload "
               "(M=24576; N x K = 1024x1024, 5460x1024, 1024x2752), keyed by launch grid.  effective_clock_GHz = GRBM_GUI_ACTIVE / kernel time; mfma_busy = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 per XCD... x 1024 SIMDs); wave state fractions over SQ_WAVE_CYCLES.", "by_grid": {}}
for grid, m in agg.items():
    mm = {k: sum(v) / len(v) for k, v in m.items()}
    wc = mm["SQ_WAVE_CYCLES"]
    out["by_grid"][str(grid)] = {"kernel_us": mm["_ns"] / 1e3, "effective_clock_GHz": mm["GRBM_GUI_ACTIVE"] / 8 / mm["_ns"], "mfma_busy": mm["SQ_VALU_MFMA_BUSY_CYCLES"] / (mm["GRBM_GUI_ACTIVE"] / 8 * 1024),
                                 "wait_any": mm["SQ_WAIT_ANY"] / wc, "wait_inst": mm["SQ_WAIT_INST_ANY"] / wc, "wait_inst_lds": mm["SQ_WAIT_INST_LDS"] / wc, "active": mm["SQ_ACTIVE_INST_ANY"] / wc,
                                 "raw": {k: v for k, v in mm.items() if k != "_ns"}}
print(json.dumps(out, indent=1))
PY
cat $R/gpurun_out/${TAG}_pmc_glds_sq.json | head -60
rm -rf $R/gpurun_out/pmc_sq
