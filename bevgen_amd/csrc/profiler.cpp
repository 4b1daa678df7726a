#include "profiler.h"

#include "common.h"

#include <exception>

namespace bevgen {

namespace {
thread_local Profiler* t_current = nullptr;
thread_local unsigned* t_status = nullptr;   // device address of the executing context's status word (common.h)

hipEvent_t get_event(Profiler& p) {
    if (p.pool_used == p.pool.size()) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        p.pool.push_back(e);
    }
    return p.pool[p.pool_used++];
}
}  // namespace

Profiler* prof_set_current(Profiler* p) {
    Profiler* old = t_current;
    t_current = p;
    return old;
}

bool prof_enabled() { return t_current && t_current->on; }

unsigned* status_current() { return t_status; }
unsigned* status_set_current(unsigned* dev) {
    unsigned* old = t_status;
    t_status = dev;
    return old;
}

ProfScope::ProfScope(int kind, double work, hipStream_t s, bool attach_) : attach(attach_), p(nullptr), idx(-1), stream(s), uncaught(std::uncaught_exceptions()) {
    if (!t_current || !t_current->on) return;
    p = t_current;
    Profiler::Rec r{kind, work, get_event(*p), get_event(*p)};
    if (!attach) HIP_CHECK(hipEventRecord(r.a, s));
    idx = (int)p->recs.size();
    p->recs.push_back(r);
}

ProfScope::~ProfScope() {
    if (idx < 0) return;
    if (std::uncaught_exceptions() > uncaught) {
        // unwinding: the enclosed launch did not happen (or failed) - forget the record instead of leaving a pair whose events were never recorded
        // (bevgen_profile_end would fail in hipEventElapsedTime and mask the original error).  Records are appended in order, so it is the last one.
        if ((size_t)idx + 1 == p->recs.size()) { p->recs.pop_back(); p->pool_used -= 2; }
        return;
    }
    if (!attach) (void)hipEventRecord(p->recs[idx].b, stream);
}

hipEvent_t ProfScope::ev_a() const { return p->recs[idx].a; }
hipEvent_t ProfScope::ev_b() const { return p->recs[idx].b; }

Profiler::~Profiler() {
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
}

void Profiler::begin() {
    recs.clear();
    pool_used = 0;
    on = true;
}

void Profiler::end(double* out) {
    on = false;
    HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < PROF_KINDS * 3; ++i) out[i] = 0.0;
    for (const Rec& r : recs) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        out[r.kind * 3 + 0] += 1.0;
        out[r.kind * 3 + 1] += ms;
        out[r.kind * 3 + 2] += r.work;
    }
    recs.clear();
    pool_used = 0;
}

}  // namespace bevgen
