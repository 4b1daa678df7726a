#!/bin/bash
# hipcc cross-compiles here without a GPU; the binary travels with the snapshot (git-ignored)
cd "$(dirname "$0")"
C=../../bevgen_amd/csrc
rm -f attn_lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBEVGEN_ATTN_LAB -I$C -o attn_lab attn_lab.cpp $C/attention_split.hip $C/profiler.cpp
