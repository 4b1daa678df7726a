#!/bin/bash
# Cache-path counters per kernel (vector L1 <-> L2 <-> fabric) for an arbitrary probe: four separate --pmc passes (kernel trace only, bounded), merged per kernel.
# usage on the GPU box: bash tools/pmc_cache.sh <tag> <python args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
P1="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
P2="TCC_HIT_sum TCC_MISS_sum"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P4="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
PASSES=${PMC_CACHE_PASSES:-"1 2 3 4"}
for i in $PASSES; do
  eval P=\$P$i
  rm -rf $R/gpurun_out/pmc_c$i
  timeout 400 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/pmc_c$i -o c --output-format csv -- python "$@" > $R/gpurun_out/pmc_c$i.log 2>&1
done
python - <<PY > $R/gpurun_out/${TAG}_pmc_cache_by_kernel.json
import csv, collections, glob, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for i in (1, 2, 3, 4):
    fs = glob.glob("$R/gpurun_out/pmc_c%d/**/*counter_collection.csv" % i, recursive=True)
    ts = glob.glob("$R/gpurun_out/pmc_c%d/**/*kernel_trace.csv" % i, recursive=True)
    if not fs: continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(ts[0]))} if ts else {}
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void bevgen::", "").replace("bevgen::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], i) not in seen:
            seen.add((r["Dispatch_Id"], i)); cnt[k][i] += 1
            if i == 1: agg[k]["_ns"] += dur.get(r["Dispatch_Id"], 0)
out = {"note": "rocprofv3 --kernel-trace --pmc, four passes (TCP->TCC requests + latency | TCC hit / miss | TCP stall cycles | TCC EA reads); per-launch means; kernels run serialised "
               "under counter collection.  l1_to_l2_read_MB prices a TCP_TCC_READ_REQ at 64 bytes; calibrated on layernorm_vec_kernel (whose read bytes are known: 142.5 MB per launch on average) a request is closer to 128 bytes - double the MB and TB/s columns; TCC_EA0_RDREQ = L2 -> fabric read requests "
               "(32 B or 64 B each: bytes = 32 * RDREQ_32B + 64 * (RDREQ - RDREQ_32B)); read latency = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ in L1 clocks", "kernels": {}}
for k, m in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"])[:10]:
    n1, n2, n3, n4 = (cnt[k][1] or 1), (cnt[k][2] or 1), (cnt[k][3] or 1), (cnt[k][4] or 1)
    rd = m["TCP_TCC_READ_REQ_sum"] / n1
    ea = m["TCC_EA0_RDREQ_sum"] / n4; ea32 = m["TCC_EA0_RDREQ_32B_sum"] / n4
    us = m["_ns"] / n1 / 1e3
    out["kernels"][k] = {"launches": n1, "avg_us_under_counters": us, "l1_to_l2_read_MB": rd * 64 / 1e6, "l1_to_l2_read_TBs": rd * 64 / (us * 1e-6) / 1e12 if us else 0,
                         "l1_read_latency_clk": m["TCP_TCC_READ_REQ_LATENCY_sum"] / (m["TCP_TCC_READ_REQ_sum"] or 1),
                         "l2_hit_rate": m["TCC_HIT_sum"] / ((m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) or 1), 
                         "l2_to_fabric_read_MB": (32 * ea32 + 64 * (ea - ea32)) / 1e6,
                         "tcp_pending_stall_per_gui_clk": m["TCP_PENDING_STALL_CYCLES_sum"] / ((m["GRBM_GUI_ACTIVE"] / 8) or 1) / 256,
                         "tcp_tcr_stall_per_gui_clk": m["TCP_TCR_TCP_STALL_CYCLES_sum"] / ((m["GRBM_GUI_ACTIVE"] / 8) or 1) / 256,
                         "tcp_ta_data_stall_per_gui_clk": m["TCP_TCP_TA_DATA_STALL_CYCLES_sum"] / ((m["GRBM_GUI_ACTIVE"] / 8) or 1) / 256}
print(json.dumps(out, indent=1))
PY
python -c "
import json; d=json.load(open('$R/gpurun_out/${TAG}_pmc_cache_by_kernel.json'))
for k,v in d['kernels'].items(): print(f\"{k[:56]:56s} n={v['launches']:4d} {v['avg_us_under_counters']:8.1f}us L1<-L2 {v['l1_to_l2_read_MB']:8.1f} MB {v['l1_to_l2_read_TBs']:5.2f} TB/s lat {v['l1_read_latency_clk']:6.0f} hit {v['l2_hit_rate']:.3f} fabric {v['l2_to_fabric_read_MB']:8.1f} MB stalls {v['tcp_pending_stall_per_gui_clk']:.2f} {v['tcp_tcr_stall_per_gui_clk']:.2f} {v['tcp_ta_data_stall_per_gui_clk']:.2f}\")"
tail -3 $R/gpurun_out/pmc_c1.log | cut -c1-200
rm -rf $R/gpurun_out/pmc_c1 $R/gpurun_out/pmc_c2 $R/gpurun_out/pmc_c3 $R/gpurun_out/pmc_c4
