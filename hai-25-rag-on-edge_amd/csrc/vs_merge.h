// vs_merge.h -- the compacting merge as a device function: ranking of candidate / partial lists (merge_compact_kernel,
// vs_seed_merge.hip) and the ranking stage of the wide IVF pipeline (ivf_wide_rank_kernel, vs_ivf.hip).
#pragma once
#include "vs_kernels.h"
#include "vs_dev.h"

namespace vs {

constexpr int kMergeTrack = 256;  // leading outputs kept on chip for the tie flag / seed threshold
struct MergeLayout {
    int64_t stride_g, stride_q;
};

constexpr int kCompactCap = 4096;

// kout rounds of wave-wide (dist, id) argmin over M <= 64*EPL candidates parked in LDS; lane-local
// candidates live in registers, the wave reduction is DPP only (no LDS traffic, no barriers).
template <int EPL>
__device__ __forceinline__ void wave_rank_and_emit(const MergeParams& p, int q, const float* cd, const int* ci, int M,
                                                   float* outd, int lane) {
    float d[EPL];
    int id[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int idx = e * 64 + lane;
        d[e] = idx < M ? cd[idx] : VS_INF;
        id[e] = idx < M ? ci[idx] : 0x7fffffff;
    }
    // lane (round % 64) keeps the round's winner; the wave writes 64 rounds at a time (the id map is read once per output
    // there, all lanes at once: read inside the rounds it costs a cache round trip per round)
    float keep_d = VS_INF;
    int keep_i = -1;
    auto flush = [&](int first, int n) {
        if (lane < n) {
            const int round = first + lane;
            if (p.out_d) p.out_d[(int64_t)q * p.kout + round] = keep_d;
            if (p.out_i) p.out_i[(int64_t)q * p.kout + round] = (keep_i >= 0 && p.id_map) ? p.id_map[keep_i] : keep_i;
        }
    };
    for (int round = 0; round < p.kout; ++round) {
        float md = d[0];
        int mi = id[0];
#pragma unroll
        for (int e = 1; e < EPL; ++e)
            if (lex_lt(d[e], id[e], md, mi)) {
                md = d[e];
                mi = id[e];
            }
        float bd;
        int bi;
        wave_lexmin(md, mi, bd, bi);
        const bool none = bi == 0x7fffffff;
        if (lane == 0 && round < kMergeTrack) outd[round] = none ? VS_INF : bd;
        if (lane == (round & 63)) {
            keep_d = none ? VS_INF : bd;
            keep_i = none ? -1 : bi;
        }
        if ((round & 63) == 63) flush(round - 63, 64);
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (id[e] == bi && d[e] == bd) {
                d[e] = VS_INF;
                id[e] = 0x7fffffff;
            }
    }
    if (p.kout & 63) flush(p.kout & ~63, p.kout & 63);
}

// (cd, ci: kCompactCap words of LDS each, from the kernel)
__device__ __forceinline__ void merge_compact_body(const MergeParams& p, const MergeLayout& L, float* const cd, int* const ci, const int q) {
    __shared__ int cnt;
    __shared__ float wbd[4];
    __shared__ int wbi[4];
    __shared__ int wbp[4];
    __shared__ float outd[kMergeTrack];
    // (q: the output query; input lists may be grouped in padded batches)
    const int q_in = p.q_group_out > 0 ? (q / p.q_group_out) * p.q_group_in + (q % p.q_group_out) : q;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
#ifdef VS_STAMPS
#define MRG_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[(int)blockIdx.x * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff); } while (0)
#else
#define MRG_STAMP(i)
#endif
    MRG_STAMP(0);
    if (p.run_if && ((p.run_mode == 1 && !p.run_if[0]) || (p.run_mode == 2 && p.run_if[0]))) return;
    if (p.invalid && p.q_group_out > 0 && p.invalid[q / p.q_group_out]) {
        // the int8 scan skipped this batch (a query was not an integer in [0, 255]): tell the caller to rerun it
        for (int t = tid; t < p.kout; t += 256) {
            if (p.out_d) p.out_d[(int64_t)q * p.kout + t] = VS_INF;
            if (p.out_i) p.out_i[(int64_t)q * p.kout + t] = -1;
        }
        if (tid == 0 && p.flags) p.flags[q] = 2;
        return;
    }
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (p.flat_len) {
        // G unsorted candidate lists per query (streaming scans): list g holds flat_len[q_in * G + g] <= kin entries
        // (the lengths are fetched together: one after the other they would cost G cache round trips)
        __shared__ int s_len[64], s_off[65];
        if (tid < p.G && tid < 64)
            s_len[tid] = min(p.flat_len[p.flat_len_sub_stride ? (int64_t)tid * p.flat_len_sub_stride + q_in : (int64_t)q_in * p.G + tid], p.kin);
        __syncthreads();
        const int G = min(p.G, 64);
        if (tid == 0) {
            int o = 0;
            for (int g = 0; g < G; ++g) {
                s_off[g] = o;
                o += s_len[g];
            }
            s_off[G] = o;
        }
        __syncthreads();
        const int off = s_off[G];
        // all lists in one pass over the entries (list by list the copies are G cache round trips one after the other)
        constexpr int GU = kCompactCap / 256;  // every entry of the longest possible set in one go: one cache round trip
        float vd[GU];
        int vi[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int e = 256 * u + tid;
            vd[u] = 0.f;
            vi[u] = 0;
            if (256 * u < off) {  // workgroup-uniform
                int g = 0;
                for (int t = 1; t < G; ++t) g += s_off[t] <= e ? 1 : 0;  // the list entry e belongs to
                const int64_t src = ((int64_t)q_in * p.G + g) * p.kin + (e - s_off[g]);
                if (e < off) {
                    vd[u] = p.part_d[src];
                    vi[u] = p.part_i[src];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int e = 256 * u + tid;
            if (e < off) {
                cd[e] = vd[u];
                ci[e] = vi[u];
            }
        }
        if (tid == 0) cnt = off;
    } else
    for (int g = tid; g < p.G; g += 256) {
        const int64_t off = (int64_t)g * L.stride_g + (int64_t)q_in * L.stride_q;
        for (int j = 0; j < p.kin; ++j) {
            const float d = p.part_d[off + j];
            const int id = p.part_i ? p.part_i[off + j] : (g * p.kin + j);
            if (!(d < VS_INF) || id < 0) {
                if (p.part_i) break;  // sorted list: the rest is padding
                continue;
            }
            const int pos = atomicAdd(&cnt, 1);
            cd[pos] = d;
            ci[pos] = id;
        }
    }
    __syncthreads();
    const int M = cnt;
    MRG_STAMP(1);
#ifdef VS_STAMPS
    if (p.dbg && tid == 0) p.dbg[(int)blockIdx.x * 16 + 8] = M;
#endif
    const int n_track = p.kout < kMergeTrack ? p.kout : kMergeTrack;

    // Many candidates (a loose bound on a few queries): the kout-th smallest of the first 256 bounds the answer; whatever
    // is not above it (usually a few dozen entries) is copied aside and ranked by one wave like a short list.
    constexpr int kKeep = 512;  // (with the 32 KB of cd/ci this keeps the kernel at 4 workgroups per CU: 1024 queries resident at once)
    __shared__ float cd2[kKeep];
    __shared__ int ci2[kKeep];
    __shared__ float s_thr_d;
    __shared__ int s_thr_i, s_keep;
    bool filtered = false;
    if (M > 256 && p.kout <= 64) {  // workgroup-uniform
        if (wave == 0) {
            float d[4];
            int id[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[e] = cd[e * 64 + lane];
                id[e] = ci[e * 64 + lane];
            }
            float bd = VS_INF;
            int bi = 0x7fffffff;
            for (int round = 0; round < p.kout; ++round) {
                float md = d[0];
                int mi = id[0];
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (lex_lt(d[e], id[e], md, mi)) {
                        md = d[e];
                        mi = id[e];
                    }
                wave_lexmin(md, mi, bd, bi);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (id[e] == bi && d[e] == bd) {
                        d[e] = VS_INF;
                        id[e] = 0x7fffffff;
                    }
            }
            if (lane == 0) {
                s_thr_d = bd;
                s_thr_i = bi;
                s_keep = 0;
            }
        }
        __syncthreads();
        const float td = s_thr_d;
        const int ti = s_thr_i;
        for (int e = tid; e < M; e += 256) {
            const float d = cd[e];
            const int id = ci[e];
            if (!lex_lt(td, ti, d, id)) {  // (d, id) <= (td, ti)
                const int pos = atomicAdd(&s_keep, 1);
                if (pos < kKeep) {
                    cd2[pos] = d;
                    ci2[pos] = id;
                }
            }
        }
        __syncthreads();
        filtered = s_keep <= kKeep;  // else: masses of ties at the threshold -> the workgroup-wide rounds below
    }
    if (filtered) {
        const int S = s_keep;
        if (wave == 0) {
            if (S <= 64) wave_rank_and_emit<1>(p, q, cd2, ci2, S, outd, lane);
            else if (S <= 256) wave_rank_and_emit<4>(p, q, cd2, ci2, S, outd, lane);
            else wave_rank_and_emit<16>(p, q, cd2, ci2, S, outd, lane);
        }
    } else if (M <= 1024) {  // (between 257 and 1024 only when kout > 64)
        if (wave == 0) {
            if (M <= 64) wave_rank_and_emit<1>(p, q, cd, ci, M, outd, lane);
            else if (M <= 256) wave_rank_and_emit<4>(p, q, cd, ci, M, outd, lane);
            else wave_rank_and_emit<16>(p, q, cd, ci, M, outd, lane);
        }
    } else {
        for (int round = 0; round < p.kout; ++round) {
            float bd = VS_INF;
            int bi = -1, bp = -1;
            for (int e = tid; e < M; e += 256) {
                const float d = cd[e];
                const int id = ci[e];
                if (id >= 0 && lex_lt(d, id, bd, bi < 0 ? 0x7fffffff : bi)) {
                    bd = d;
                    bi = id;
                    bp = e;
                }
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const float od2 = __shfl_xor(bd, m);
                const int oi2 = __shfl_xor(bi, m);
                const int op2 = __shfl_xor(bp, m);
                if (oi2 >= 0 && (bi < 0 || lex_lt(od2, oi2, bd, bi))) {
                    bd = od2;
                    bi = oi2;
                    bp = op2;
                }
            }
            if (lane == 0) {
                wbd[wave] = bd;
                wbi[wave] = bi;
                wbp[wave] = bp;
            }
            __syncthreads();
            bd = wbd[0];
            bi = wbi[0];
            bp = wbp[0];
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (wbi[w] >= 0 && (bi < 0 || lex_lt(wbd[w], wbi[w], bd, bi))) {
                    bd = wbd[w];
                    bi = wbi[w];
                    bp = wbp[w];
                }
            if (bi < 0) bd = VS_INF;
            __syncthreads();
            if (tid == 0) {
                if (bp >= 0) ci[bp] = -1;  // consumed
                if (round < kMergeTrack) outd[round] = bd;
                if (p.out_d) p.out_d[(int64_t)q * p.kout + round] = bd;
                if (p.out_i) p.out_i[(int64_t)q * p.kout + round] = (bi >= 0 && p.id_map) ? p.id_map[bi] : bi;
            }
            __syncthreads();
        }
    }
    MRG_STAMP(2);
    if (tid == 0) {
        if (p.flags) {
            int f = 0;
            for (int i = 0; i + 1 < n_track; ++i)
                if (outd[i] == outd[i + 1] && outd[i] < VS_INF) f = 1;
            if (p.flag_empty && !(outd[0] < VS_INF)) f = 2;
            if (p.shard_flags)
                for (int g = 0; g < p.G; ++g)
                    if (p.shard_flags[(int64_t)g * p.shard_flags_stride + q] == 2) f = 2;
            p.flags[q] = f;
        }
        if (p.tau_out) {
            const float kth = outd[n_track - 1];
            p.tau_out[q] = kth < VS_INF ? next_up(kth) : VS_INF;
        }
    }
}

}  // namespace vs
