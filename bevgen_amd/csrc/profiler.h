// Optional per-launch timing of the hot kernels with HIP events recorded on the launch stream (bench.py's roofline leg).
// Disabled by default: when off, the wrappers cost one branch.  The state is PER CONTEXT (Ctx::prof): the C-ABI entry points make their context's profiler
// the calling thread's current one for the duration of the call, so two contexts in one process never interleave their records.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace bevgen {

// PROF_GEMM_SMALL: launches of the LDS-DMA GEMM that are NOT its 256-row throughput instantiation (small-problem blocks, the short last part of a row-split
// launch): kept apart so that PROF_GEMM's launches / milliseconds are those of ONE kernel and can be checked against a rocprof trace
enum ProfKind { PROF_GEMM = 0, PROF_CONV3 = 1, PROF_ATTN = 2, PROF_DECODE_ATTN = 3, PROF_GEMM_SKINNY = 4, PROF_GEMM_SMALL = 5, PROF_KINDS = 6 };

struct Profiler {
    struct Rec { int kind; double work; hipEvent_t a, b; };
    bool on = false;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    ~Profiler();
    void begin();
    // out[kind*3 + {0,1,2}] = launches, total milliseconds, total work (flops or bytes); synchronises the device
    void end(double* out);
};

// the profiler the launch wrappers of THIS thread report to (null = none); returns the previous one
Profiler* prof_set_current(Profiler* p);

struct ProfScope {
    // records an event pair around the enclosed launch(es) when the calling thread's current profiler is enabled.
    // attach = true: the pair is NOT recorded on the stream; the launcher hands ev_a() / ev_b() to hipExtLaunchKernelGGL instead, which stamps them with the kernel's own
    // begin / end (what rocprofv3 --kernel-trace reports), without the two marker packets whose processing a stream-recorded pair adds to every short launch (~2 us)
    ProfScope(int kind, double work, hipStream_t s, bool attach = false);
    ~ProfScope();
    bool attached() const { return idx >= 0 && attach; }
    hipEvent_t ev_a() const;
    hipEvent_t ev_b() const;
    bool attach = false;
    Profiler* p;
    int idx;
    hipStream_t stream;
    int uncaught;   // std::uncaught_exceptions() at construction: a destructor that runs during unwinding drops its record
};

bool prof_enabled();

}  // namespace bevgen
