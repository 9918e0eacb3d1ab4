// Optional per-launch HIP-event timing, recorded on the stream the kernel is launched on.
// Used only by bench.py's roofline leg (omgsr_timing_*); off by default and free when off.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

enum { OMGSR_TK_IGEMM = 1, OMGSR_TK_ATTN = 2, OMGSR_TK_GN = 3, OMGSR_TK_LN = 4, OMGSR_TK_ELT = 5, OMGSR_TK_SOFTMAX = 6 };

namespace omgsr {
struct TimingRec { int kind; double flops, bytes; long long m, n, k; hipEvent_t e0, e1; int variant = 0; int stage = 0; };
struct TimingState {
    bool on = false;
    int stage = 0;
    std::vector<TimingRec> recs;
};
TimingState& timing_state();

struct TimingScope {
    hipStream_t st; bool active; TimingRec rec;
    TimingScope(int kind, double flops, double bytes, hipStream_t s, long long m = 0, long long n = 0, long long k = 0)
        : st(s), active(timing_state().on) {
        if (!active) return;
        rec.kind = kind; rec.stage = timing_state().stage; rec.flops = flops; rec.bytes = bytes; rec.m = m; rec.n = n; rec.k = k;
        (void)hipEventCreate(&rec.e0); (void)hipEventCreate(&rec.e1);
        (void)hipEventRecord(rec.e0, st);
    }
    ~TimingScope() {
        if (!active) return;
        (void)hipEventRecord(rec.e1, st);
        timing_state().recs.push_back(rec);
    }
};
}  // namespace omgsr
