#include "profiler.h"

#include <vector>

#include "common.h"

namespace bevgen {

namespace {
struct Rec { int kind; double work; hipEvent_t a, b; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_used = 0;

hipEvent_t get_event() {
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        g_pool.push_back(e);
    }
    return g_pool[g_pool_used++];
}
}  // namespace

bool prof_enabled() { return g_on; }

ProfScope::ProfScope(int kind, double work, hipStream_t s) : idx(-1), stream(s) {
    if (!g_on) return;
    Rec r{kind, work, get_event(), get_event()};
    HIP_CHECK(hipEventRecord(r.a, s));
    idx = (int)g_recs.size();
    g_recs.push_back(r);
}

ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_recs[idx].b, stream);
}

void prof_begin() {
    g_recs.clear();
    g_pool_used = 0;
    g_on = true;
}

void prof_end(double* out) {
    g_on = false;
    HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < PROF_KINDS * 3; ++i) out[i] = 0.0;
    for (const Rec& r : g_recs) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        out[r.kind * 3 + 0] += 1.0;
        out[r.kind * 3 + 1] += ms;
        out[r.kind * 3 + 2] += r.work;
    }
    g_recs.clear();
    g_pool_used = 0;
}

}  // namespace bevgen
