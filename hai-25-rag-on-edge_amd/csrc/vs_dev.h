// vs_dev.h -- device-side helpers shared by the kernel translation units (wave-wide DPP reductions, lane-private
// sorted lists, ordered-float keys).  Wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VS_INF __builtin_huge_valf()

// next representable float above x (x finite): the seed threshold must admit ties at the k-th value
__device__ __forceinline__ float next_up(float x) {
    if (x == 0.f) return __builtin_bit_cast(float, 1);
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b + (b >= 0 ? 1 : -1));
}

__device__ __forceinline__ bool lex_lt(float d0, int i0, float d1, int i1) {
    return (d0 < d1) || (d0 == d1 && i0 < i1);
}

// Insert (d, id) into a per-lane ascending list kept in registers.  Ordering is
// (dist, id) so that the result does not depend on which lane saw which row.
template <int KCAP>
__device__ __forceinline__ void list_insert(float (&ld)[KCAP], int (&li)[KCAP], float d, int id) {
    float cd = d;
    int ci = id;
#pragma unroll
    for (int j = 0; j < KCAP; ++j) {
        const bool lt = lex_lt(cd, ci, ld[j], li[j]);
        const float td = lt ? ld[j] : cd;
        const int ti = lt ? li[j] : ci;
        ld[j] = lt ? cd : ld[j];
        li[j] = lt ? ci : li[j];
        cd = td;
        ci = ti;
    }
}

// ------------------------------------------------------------------------------------------------
// DPP helpers: wave-wide reductions without LDS round trips (ds_bpermute chains cost ~10x more
// latency than these when a single wave runs them back to back).
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float x) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int x) {
    return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ float rdlane_f(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
// min over the 64 lanes, result uniform
__device__ __forceinline__ float wave_min_f32(float x) {
    x = fminf(x, dpp_mov_f<0xB1>(x));   // quad_perm [1,0,3,2]
    x = fminf(x, dpp_mov_f<0x4E>(x));   // quad_perm [2,3,0,1]
    x = fminf(x, dpp_mov_f<0x141>(x));  // row_half_mirror
    x = fminf(x, dpp_mov_f<0x140>(x));  // row_mirror
    return fminf(fminf(rdlane_f(x, 0), rdlane_f(x, 16)), fminf(rdlane_f(x, 32), rdlane_f(x, 48)));
}
__device__ __forceinline__ int wave_min_i32(int x) {
    x = min(x, dpp_mov_i<0xB1>(x));
    x = min(x, dpp_mov_i<0x4E>(x));
    x = min(x, dpp_mov_i<0x141>(x));
    x = min(x, dpp_mov_i<0x140>(x));
    return min(min(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)),
               min(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}
// lexicographic (dist, id) argmin over the wave; ids are unique or negative
__device__ __forceinline__ void wave_lexmin(float d, int id, float& bd, int& bi) {
    bd = wave_min_f32(d);
    bi = wave_min_i32(d == bd ? id : 0x7fffffff);
}

// k-th smallest (1-based) of the 256 values held 4 per lane across the wave; +inf if fewer are finite
__device__ __forceinline__ float wave_kth_smallest(float a0, float a1, float a2, float a3, int k, int lane) {
    float res = VS_INF;
    for (int round = 0; round < k; ++round) {
        const float m = fminf(fminf(a0, a1), fminf(a2, a3));
        const float wm = wave_min_f32(m);
        res = wm;
        if (!(wm < VS_INF)) break;
        const unsigned long long mask = __ballot(m == wm);
        if (lane == __builtin_ctzll(mask)) {  // drop exactly one instance
            if (a0 == wm) a0 = VS_INF;
            else if (a1 == wm) a1 = VS_INF;
            else if (a2 == wm) a2 = VS_INF;
            else a3 = VS_INF;
        }
    }
    return res;
}

// kout rounds of wave-wide (dist, id) argmin over M <= 64*EPL candidates parked in LDS (cd/ci);
// lane-local candidates live in registers, the reduction is DPP only.  Writes kout (dist, id)
// pairs (padded with +inf / -1) and, if flag != nullptr, whether two emitted distances are equal.
template <int EPL>
__device__ __forceinline__ void wave_rank_emit(const float* cd, const int* ci, int M, int kout, float* out_d,
                                               int32_t* out_i, int32_t* flag, const int32_t* id_map, int lane) {
    float d[EPL];
    int id[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int idx = e * 64 + lane;
        d[e] = idx < M ? cd[idx] : VS_INF;
        id[e] = idx < M ? ci[idx] : 0x7fffffff;
    }
    float prev = VS_INF;
    int tie = 0;
    for (int round = 0; round < kout; ++round) {
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
        if (!none && round > 0 && bd == prev) tie = 1;
        prev = none ? VS_INF : bd;
        if (lane == 0) {
            if (out_d) out_d[round] = none ? VS_INF : bd;
            if (out_i) out_i[round] = none ? -1 : (id_map ? id_map[bi] : bi);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (id[e] == bi && d[e] == bd) {
                d[e] = VS_INF;
                id[e] = 0x7fffffff;
            }
    }
    if (flag && lane == 0) *flag = tie;
}

// Rank by counting: lane e < M holds pair e; returns the number of pairs before it in (dist, id) order (ids are unique).
__device__ __forceinline__ int wave_rank_count(float d, int id, int M) {
    int rank = 0;
    for (int j = 0; j < M; ++j) {
        const float dj = rdlane_f(d, j);
        const int ij = __builtin_amdgcn_readlane(id, j);
        rank += lex_lt(dj, ij, d, id) ? 1 : 0;
    }
    return rank;
}

// partial sums across lanes 1, 2 and 8 apart (DPP: no LDS round trip)
__device__ __forceinline__ float dpp_add_xor1(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_add_xor2(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_add_half_mirror(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
}

// wave-resident sorted list: lane j (< KCAP <= 16) holds entry j
// order-preserving map float -> unsigned (for integer min/max on distances of either sign)
__device__ __forceinline__ unsigned f32_ordered(float x) {
    const unsigned b = __builtin_bit_cast(unsigned, x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(unsigned u) {
    const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, b);
}

// minimum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the four rows of a column in an MFMA 16x16 result), in every one
// of them.  gfx950's row swaps: v_permlane32_swap(a, b) exchanges a's upper half with b's lower half -- with a = b = x that
// leaves (lo, lo) and (hi, hi); v_permlane16_swap does the same with odd / even rows of 16.  Plain VALU, no LDS round trip.
__device__ __forceinline__ unsigned col4_min_u32(unsigned x) {
    const auto h = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    const unsigned m = min(h[0], h[1]);
    const auto q = __builtin_amdgcn_permlane16_swap(m, m, false, false);
    return min(q[0], q[1]);
}

}  // namespace vs

// time stamps of diagnostic builds (-DVS_STAMPS): p.dbg[workgroup][16]
#ifdef VS_STAMPS
#define VS_STAMP(i)                                                                                    \
    do {                                                                                               \
        if (p.dbg && threadIdx.x == 0)                                                                 \
            p.dbg[blockIdx.x * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);       \
    } while (0)
#define VS_STAMPC(i)                                                                                   \
    do {                                                                                               \
        if (p.dbg && threadIdx.x == 0)                                                                 \
            p.dbg[blockIdx.x * 16 + (i)] = (int)(__builtin_amdgcn_s_memtime() & 0x7fffffff);           \
    } while (0)
#else
#define VS_STAMP(i)
#define VS_STAMPC(i)
#endif
