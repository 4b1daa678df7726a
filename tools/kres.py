#!/usr/bin/env python
"""Register / spill / LDS usage of the kernels of one source file (cross-compile, no GPU): python tools/kres.py decode_fused.hip [name-filter]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bevgen_amd", "csrc")
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-c", src, "-o", os.devnull,
                    "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], cwd=csrc, capture_output=True, text=True)
if r.returncode: print(r.stderr[-3000:]); sys.exit(1)
name, cur = None, {}
def flush():
    if name and flt in name:
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("bevgen::", "").split("(")[0]
        print(f"{short:<70} vgpr {cur.get('VGPRs','?'):>4} agpr {cur.get('AGPRs','?'):>3} spill {cur.get('VGPRs Spill','?'):>3} sgpr {cur.get('TotalSGPRs','?'):>4} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4} lds {cur.get('LDS Size [bytes/block]','?'):>6} occ {cur.get('Occupancy [waves/SIMD]','?')}")
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        flush(); name, cur = m.group(1), {}
    m = re.search(r"(VGPRs Spill|AGPRs|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and name: cur[m.group(1)] = m.group(2)
flush()
