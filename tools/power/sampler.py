#!/usr/bin/env python
"""Sample socket power and clocks of GPU 0 at ~20 Hz until <out>.stop appears.  Sources, in order: amdgpu hwmon sysfs (power1_average / power1_input in uW,
freq1_input = sclk in Hz, freq2_input = mclk), else `amd-smi metric --power --clock --json`.   usage: sampler.py out.csv"""
import glob, json, os, subprocess, sys, time

out = sys.argv[1]
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))


def rd(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return ""


def sysfs_sample():
    best = None
    for h in hw:
        p = rd(h + "/power1_average") or rd(h + "/power1_input")
        s = rd(h + "/freq1_input")
        m = rd(h + "/freq2_input")
        t = rd(h + "/temp1_input")
        if p or s:
            row = (float(p) / 1e6 if p else float("nan"), float(s) / 1e6 if s else float("nan"), float(m) / 1e6 if m else float("nan"), float(t) / 1e3 if t else float("nan"))
            if best is None or (row[0] == row[0] and row[0] > (best[0] if best[0] == best[0] else -1)):
                best = row
    return best


def smi_sample():
    try:
        r = subprocess.run(["amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=3)
        j = json.loads(r.stdout)
        j = j[0] if isinstance(j, list) else j
        pw = j.get("power", {})
        p = pw.get("socket_power", pw.get("current_socket_power", {}))
        p = p.get("value", float("nan")) if isinstance(p, dict) else p
        ck = j.get("clock", {})
        g = ck.get("gfx_0", ck.get("gfx", {}))
        s = g.get("clk", {}).get("value", float("nan")) if isinstance(g, dict) else float("nan")
        m = ck.get("mem_0", {}).get("clk", {}).get("value", float("nan"))
        return (float(p), float(s), float(m), float("nan"))
    except Exception:
        return None


def all_cards():
    """(power W, sclk MHz) of every amdgpu hwmon the host exposes: the box is one GPU of a multi-GPU node, the container's sysfs shows all of them;
    OUR card is the one whose columns move with our load (and the one amd-smi reports)."""
    row = []
    for h in hw:
        p = rd(h + "/power1_average") or rd(h + "/power1_input")
        s = rd(h + "/freq1_input")
        row.append((float(p) / 1e6 if p else float("nan"), float(s) / 1e6 if s else float("nan")))
    return row


use_sysfs = sysfs_sample() is not None
with open(out, "w") as f:
    f.write(f"# source: {'amdgpu hwmon sysfs, one (power W, sclk MHz) pair per card: ' + ' '.join(hw) if use_sysfs else 'amd-smi metric --json'}\n")
    caps = [rd(h + "/power1_cap") for h in hw]
    f.write("# power1_cap (uW) per card: " + " ".join(caps) + "\n")
    f.write("t_s," + (",".join(f"p{i}_w,sclk{i}_mhz" for i in range(len(hw))) if use_sysfs else "power_w,sclk_mhz,mclk_mhz,temp_c") + "\n")
    t0 = time.time()
    while not os.path.exists(out + ".stop"):
        if use_sysfs:
            f.write(f"{time.time() - t0:.3f}," + ",".join(f"{p:.0f},{s:.0f}" for p, s in all_cards()) + "\n")
        else:
            s = smi_sample()
            if s is not None:
                f.write(f"{time.time() - t0:.3f},{s[0]:.1f},{s[1]:.0f},{s[2]:.0f},{s[3]:.1f}\n")
        f.flush()
        time.sleep(0.04 if use_sysfs else 0.0)
os.remove(out + ".stop")
