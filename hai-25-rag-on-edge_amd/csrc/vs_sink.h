// vs_sink.h -- where the streaming scans and the wide IVF scan put what they find (CandSink, vs_kernels.h): a wave bins
// its private candidate buffer into the per-query lists at the end of its kernel.
#pragma once
#include "vs_kernels.h"
#include "vs_dev.h"

namespace vs {

// A wave bins the candidates it has collected (its private buffer `wb`, `n` entries) into the per-query lists.  Called at
// the very end of a streaming scan, outside the tile loop: the returning atomics cost nothing there.
struct SinkEntryAsIs {
    __device__ __forceinline__ int4 operator()(const int4& c) const { return c; }
};
template <typename Fix = SinkEntryAsIs>
__device__ __forceinline__ void sink_bin_wave(const CandSink& p, int wb, int n, int lane, const Fix fix = Fix(), int diag = 0) {
    if (n > p.wcap) {
        if (lane == 0) p.overflow[0] = 1;  // entries were dropped: the fallback kernels behind take over
        return;
    }
    if (n == 0) return;
    // The wave reads back what it stored itself.  Its stores are complete after the wait; the lines were never read in
    // this launch before (so no stale copy can sit in this CU's vector cache), and the loads below are agent-scope
    // atomic loads anyway, which do not hit that cache.  (An acquire fence here invalidates the whole vector cache
    // under the CU's other 15 waves: measured 11 us of a 45 us kernel.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int4* src = p.wbuf + (int64_t)wb * p.wcap;
    // (a hash: neighbouring buffers hold neighbouring units of one list, whose candidates belong to the same queries)
    const unsigned hash = ((unsigned)wb * 2654435761u) >> 16;
    int sub = (int)(hash % (unsigned)p.nsub);
    if (p.xcd_subs) sub = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7) * p.xcd_subs + (int)(hash % (unsigned)p.xcd_subs);  // XCC_ID
    for (int e = lane; e < n; e += 64) {
        const unsigned long long* s64 = reinterpret_cast<const unsigned long long*>(src + e);
        const unsigned long long lo = __hip_atomic_load(s64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(s64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int4 c = fix(make_int4((int)lo, (int)(lo >> 32), (int)hi, (int)(hi >> 32)));  // (query, distance bits, id, -)
        const int64_t lst = (int64_t)c.x * p.nsub + sub;
#ifdef VS_STAMPS
        if (diag & 32) {
            p.cand_d[lst * p.cap + (e & 63)] = __builtin_bit_cast(float, c.y);
            continue;
        }
        if (diag & 64) continue;
#endif
        const int pos = atomicAdd(p.cnt + (p.xcd_subs ? (int64_t)sub * p.cnt_sub_stride + c.x : lst), 1);
        if (pos < p.cap) {
            p.cand_d[lst * p.cap + pos] = __builtin_bit_cast(float, c.y);
            p.cand_i[lst * p.cap + pos] = c.z;
        } else if (p.slow) {
            p.slow[c.x] = 1;    // more rows under this query's bound than its lists hold: the exact slow path takes it
        } else {
            p.overflow[0] = 1;  // ... or, where there is no per-query slow path, the launch's fallback kernels
        }
    }
}

}  // namespace vs
