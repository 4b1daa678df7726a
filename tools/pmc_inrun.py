#!/usr/bin/env python
"""In-run HBM traffic of the headline workload's dominant kernel (bench.py `roofline.traffic`).

Two uses:
  python tools/pmc_inrun.py probe            the child: ONE transformer forward of the bench workload (Route M, 6 x 256 x 256, 16 scenes) -
                                             98 launches of the LDS-DMA split-precision GEMM with the real mix of shapes and epilogues
  pmc_inrun.measure(kernel_substr)           the parent (called by bench.py on rank 0 at N = 1, outside the timed region): runs the child under
                                             `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes, kernel trace only:
                                             MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots": the two counters do not fit one pass) and returns the mean bytes per
                                             launch of the kernels whose name contains `kernel_substr`, with the guide's gfx950 correction (FETCH_SIZE counts half of
                                             a wide coalesced read: x2; WRITE_SIZE as reported; both in KiB).
"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAL_M, CAL_N, CAL_K = 24576, 128, 1024   # calibration GEMM: one column tile


def probe():
    sys.path.insert(0, ROOT)
    import torch
    from bevgen_amd import presets, synthetic
    from bevgen_amd.runtime import Context
    from bevgen_amd.weights import maskgit_state_dict

    cams, batch = int(os.environ.get("PMC_CAMS", "6")), int(os.environ.get("PMC_BATCH", "16"))
    cfg = presets.config2(cams)
    ctx = Context(cfg, route="maskgit", device=0, max_batch=batch, precision="f16x3")
    ctx.load_state_dict(maskgit_state_dict(cfg, 1234))
    ctx.set_tables()
    ctx.finalize()
    bt = {k: v.to(ctx.device) for k, v in synthetic.make_batch(cfg, batch, seed=1000).items()}
    ids = torch.full((batch * cams, cfg.num_cam_tokens), cfg.vocab_size, dtype=torch.long, device=ctx.device)
    ctx.muse_forward(ids, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], want_logits=True, want_embed=False)
    torch.cuda.synchronize()
    # calibration launch (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"): the same LDS-DMA GEMM with ONE column tile
    # (N = 128): every A panel can only be read once, so its memory-side reads are (M K + N K) 4 bytes whatever the tile order - the LAST gemm_split_glds dispatch of the run
    from bevgen_amd.runtime import _ptr, _stream
    a = torch.randn(CAL_M, CAL_K, device=ctx.device)
    w = torch.randn(CAL_N, CAL_K, device=ctx.device)
    out = torch.empty(CAL_M, CAL_N, device=ctx.device)
    ctx._check(ctx.lib.bevgen_op_gemm(ctx._h, _ptr(a), _ptr(w), None, None, _ptr(out), CAL_M, CAL_N, CAL_K, 0, 3, _stream()))
    torch.cuda.synchronize()
    ctx.close()


def _mean_counter(csv_path, counter, kernel_substr):
    """-> (mean over the launches of the kernel, their number, value of the calibration launch = the last gemm_split_glds dispatch)"""
    vals, cal = [], (-1, None)
    for r in csv.DictReader(open(csv_path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r.get("Kernel_Name", "")
        if "gemm_split_glds_kernel" in name and int(r["Dispatch_Id"]) > cal[0]:
            cal = (int(r["Dispatch_Id"]), float(r["Counter_Value"]))
        if kernel_substr in name:
            vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    vals = [v for d, v in vals if d != cal[0]]
    return (sum(vals) / len(vals), len(vals), cal[1]) if vals else (None, 0, cal[1])


def measure(kernel_substr, timeout=240):
    """-> dict(bytes_per_launch, read_bytes_corrected, write_bytes, launches, ...) or dict(error=...)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bevgen_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "probe"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(files)} csv; {r.stderr[-300:]}"}
            mean, n, cal = _mean_counter(files[0], counter, kernel_substr)
            if mean is None:
                return {"error": f"no {counter} rows for a kernel containing '{kernel_substr}'"}
            out[counter] = (mean, n, cal)
        except subprocess.TimeoutExpired:
            return {"error": f"rocprofv3 --pmc {counter}: timeout after {timeout} s"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = 2.0 * out["FETCH_SIZE"][0] * 1024.0, out["WRITE_SIZE"][0] * 1024.0
    res = {}
    if out["FETCH_SIZE"][2] and out["WRITE_SIZE"][2] is not None:
        cal_rd_alg, cal_wr_alg = (CAL_M * CAL_K + CAL_N * CAL_K) * 4.0, CAL_M * CAL_N * 4.0
        res["calibration"] = {"shape": f"same kernel family, M={CAL_M} N={CAL_N} K={CAL_K} (one column tile: every A panel is read exactly once)",
                              "read_bytes_corrected_over_algorithmic": 2.0 * out["FETCH_SIZE"][2] * 1024.0 / cal_rd_alg,
                              "write_bytes_over_algorithmic": out["WRITE_SIZE"][2] * 1024.0 / cal_wr_alg,
                              "note": "a ratio near 2 on the read side means the guide's x2 correction of FETCH_SIZE does not apply to this kernel's 128-byte LDS-DMA lines"}
    return {**res, "bytes_per_launch": rd + wr, "read_bytes_corrected": rd, "write_bytes": wr, "launches": out["FETCH_SIZE"][1],
            "FETCH_SIZE_KiB": out["FETCH_SIZE"][0], "WRITE_SIZE_KiB": out["WRITE_SIZE"][0],
            "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child passes) over one transformer forward of this workload; "
                      "read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB; mean over the launches of the kernel"}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "probe":
        probe()
    else:
        print(measure(sys.argv[1] if len(sys.argv) > 1 else "gemm_split_glds_kernel<0, 4, 3"))
