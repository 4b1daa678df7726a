#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE counter CSVs (rocprofv3 --pmc, one counter per pass) -> per-kernel HBM traffic JSON (profiles/*_pmc_hbm_traffic.json).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read."""
import csv, json, sys, collections

import os
STEPS = int(os.environ.get("PMC_DECODE_STEPS", "40"))
MEAN_N = 256 + 1 + (STEPS - 2) / 2.0
try:   # block-sparse layouts (PMC_DENSITY < 1): tools/pmc_probe.py records the fraction of the causal keys that sit in present blocks
    VISIBLE = float(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pmc_visible.txt")).read())
except Exception:
    VISIBLE = 1.0
SHAPES = {
    "gemm_f32_kernel": ("M=24576 N=1024 K=1024 fp32", (24576 * 1024 + 1024 * 1024 + 24576 * 1024) * 4),
    "gemm_split_glds_kernel": ("M=24576 N=1024 K=1024, A and B as interleaved hi/lo f16 planes (4 B/element), fp32 C", (24576 * 1024 + 1024 * 1024 + 24576 * 1024) * 4),
    "ar_attn_fused_kernel": (f"Route A config 4, B=16 H=16 fp32 KV, mean over {STEPS} decode steps x 24 layers (mean context {MEAN_N:.0f}, visible fraction {VISIBLE:.3f}): K/V rows of present blocks once + the layer's q/k/v weight (12.6 MB, read through L2 by the 16 workgroups of a head)",
                             2 * 16 * 16 * MEAN_N * VISIBLE * 64 * 4 + 3 * 1024 * 1024 * 4),
    "ar_mlp_fused_kernel<0>": ("ln2 (folded) + MLP up + GELU + MLP down in one launch (round 5), M=16 D=1024, fp32 weights: both matrices once + rows in, hidden out and back (XCD-local exchange), 8 partial planes out",
                               (2 * 4096 * 1024 + 16 * 1024 + 2 * 16 * 4096 + 8 * 16 * 1024) * 4),
    "skinny_fused_kernel<true, 0, false, true>": ("ln2 (folded) + MLP up-projection M=16 N=4096 K=1024, fp32 weights", (4096 * 1024 + 16 * 1024 + 16 * 4096) * 4),
    "skinny_fused_kernel<true, 0, false, false>": ("ln_f + head M=16 N=1024 K=1024, fp32 weights", (1024 * 1024 + 16 * 1024 + 16 * 1024) * 4),
    "skinny_fused_kernel<false": ("MLP down-projection M=16 N=1024 K=4096, split over K (4 partial sums written)", (1024 * 4096 + 16 * 4096 + 4 * 16 * 1024) * 4),
}


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = {}
    for k, v in acc.items():
        v.sort()
        if "gemm_" in k:   # the probe launches its bench-sized GEMMs first (5 each); later launches of the same kernels belong to the Route A prefill
            v = v[:5]
        out[k] = sum(x for _, x in v) / len(v)
    return out


def main(fetch_csv, write_csv):
    f = mean_by_kernel(fetch_csv, "FETCH_SIZE")
    w = mean_by_kernel(write_csv, "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over tools/pmc_probe.py on MI355X; counters are KiB per "
                   "dispatch. gfx950 correction (MI355X_MICROARCH.md HBM section): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> "
                   "read_bytes = 2*FETCH_SIZE*1024; WRITE_SIZE uncorrected.", "kernels": {}}
    for name, fk in f.items():
        for key, (shape, alg) in SHAPES.items():
            if key in name:
                wk = w.get(name, 0.0)
                rd, wr = 2 * fk * 1024, wk * 1024
                short = name.split("(")[0].replace("void bevgen::", "").replace("bevgen::", "")
                out["kernels"][short] = {"shape": shape, "algorithmic_bytes": alg, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "read_bytes_corrected": rd,
                                         "write_bytes": wr, "traffic_bytes": rd + wr, "traffic_over_algorithmic": (rd + wr) / alg}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
