#!/bin/bash
# SQ / GRBM counters per kernel for an arbitrary probe (one --pmc pass, kernel trace only, bounded).  usage: bash tools/pmc_sq.sh <tag> <python args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
rm -rf $R/gpurun_out/pmc_sq2
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq2 -o sq --output-format csv -- python "$@" > $R/gpurun_out/pmc_sq2.log 2>&1
python - <<PY > $R/gpurun_out/${TAG}_pmc_sq_by_kernel.json
import csv, collections, glob, json, re
f = glob.glob("$R/gpurun_out/pmc_sq2/**/*counter_collection.csv", recursive=True)[0]
t = glob.glob("$R/gpurun_out/pmc_sq2/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t))}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void bevgen::", "").replace("bevgen::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k] += 1; agg[k]["_ns"] += dur.get(r["Dispatch_Id"], 0)
out = {"note": "rocprofv3 --kernel-trace --pmc SQ_* GRBM_GUI_ACTIVE (one pass; kernels run serialised under counter collection). effective_clock_GHz = GRBM_GUI_ACTIVE/8 per ns of kernel time; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); valu/wait/active = fractions of SQ_WAVE_CYCLES (quad-cycles)", "kernels": {}}
for k, m in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"])[:14]:
    wc = m["SQ_WAVE_CYCLES"] or 1.0; gui = m["GRBM_GUI_ACTIVE"] / 8 or 1.0
    out["kernels"][k] = {"launches": cnt[k], "total_us": m["_ns"] / 1e3, "effective_clock_GHz": gui / (m["_ns"] or 1), "mfma_busy": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024),
                         "valu_active": m["SQ_ACTIVE_INST_VALU"] / wc, "wait_any": m["SQ_WAIT_ANY"] / wc, "wait_inst": m["SQ_WAIT_INST_ANY"] / wc, "active_any": m["SQ_ACTIVE_INST_ANY"] / wc}
print(json.dumps(out, indent=1))
PY
python -c "
import json; d=json.load(open('$R/gpurun_out/${TAG}_pmc_sq_by_kernel.json'))
for k,v in d['kernels'].items(): print(f\"{k[:60]:60s} n={v['launches']:4d} {v['total_us']:9.0f}us clk {v['effective_clock_GHz']:.2f} mfma {v['mfma_busy']:.2f} valu {v['valu_active']:.2f} wait_any {v['wait_any']:.2f} wait_inst {v['wait_inst']:.2f} act {v['active_any']:.2f}\")"
rm -rf $R/gpurun_out/pmc_sq2
