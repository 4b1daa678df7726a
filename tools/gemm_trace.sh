#!/bin/bash
# builds the trace variant of the library (phase stamps in gemm_split_glds_kernel) next to the product one and runs tools/gemm_trace.py on the bench's projection shapes
cd bevgen_amd/csrc && mkdir -p build_trace && make -j16 > /dev/null && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DBEVGEN_GEMM_TRACE -c gemm_split_glds.hip -o build_trace/gemm_split_glds.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libbevgen_hip_trace.so $(ls build/*.o | grep -v gemm_split_glds.o) build_trace/gemm_split_glds.o && cd ../..
