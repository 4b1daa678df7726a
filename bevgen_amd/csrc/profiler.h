// Optional per-launch timing of the hot kernels with HIP events recorded on the launch stream (bench.py's roofline leg).
// Disabled by default: when off, the wrappers cost one branch.
#pragma once
#include <hip/hip_runtime.h>

namespace bevgen {

enum ProfKind { PROF_GEMM = 0, PROF_CONV3 = 1, PROF_ATTN = 2, PROF_DECODE_ATTN = 3, PROF_GEMM_SKINNY = 4, PROF_KINDS = 5 };

struct ProfScope {
    // records an event pair around the enclosed launch(es) when profiling is enabled
    ProfScope(int kind, double work, hipStream_t s);
    ~ProfScope();
    int idx;
    hipStream_t stream;
};

void prof_begin();
// out[kind*3 + {0,1,2}] = launches, total milliseconds, total work (flops or bytes); synchronises the device
void prof_end(double* out);
bool prof_enabled();

}  // namespace bevgen
