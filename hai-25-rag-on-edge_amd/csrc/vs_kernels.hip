// vs_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the distance + top-k hot path.
//
//   scan_kernel        : Q[<=32 x 128] x base^T on v_mfma_f32_16x16x4_f32 with the L2 epilogue
//                        (cpu_baseline.cpp:229-242) and the top-k (cpu_baseline.cpp:127-153) fused in;
//                        the B x N score matrix of QnnRunner::executeBatchRaw is never written
//                        unless asked for (kModeStore: IVF coarse stage, tie fallback).
//   merge_kernel       : cross-workgroup / cross-list / cross-GPU merge of sorted partial lists.
//   row_sqnorm_kernel  : compute_norms (cpu_baseline.cpp:95-125) in the reference's summation order.
//   ivf_scan_kernel    : computeDotProductsContiguous + heap top-k (IVFIndex.cpp:270-358, 738-767)
//                        as a coalesced list scan with a wave-resident sorted list.
//
// Wavefront = 64 lanes everywhere; nothing here is written for 32-wide warps.
#include "vs_kernels.h"

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
// Brute-force scan, register-direct variant.
//
// Workgroup = 8 waves (2 per SIMD).  Each wave owns whole 16-row base tiles: lane (r = l & 15,
// g = l >> 4) loads base[row0 + r][16 t + 4 g .. +3] for t = 0..7 straight into VGPRs (eight
// global_load_dwordx4, 64 contiguous bytes per row per instruction) and feeds element i of chunk t
// to MFMA step (t, i) as the A operand; the query fragments use the same k permutation as the B
// operand, so D[row][query] accumulates the full 128-long dot product.  In the 16x16 result a lane
// holds one query column (l & 15) and four base rows (4 g + reg): the top-k state of a query is
// therefore lane-private and needs no cross-lane traffic until the workgroup is done.
//
// Two tiles are kept in flight per wave (one being multiplied, one landing), i.e. 8 waves x 8 KB
// per CU, which is what it takes to cover HBM latency at ~10 B/clk/CU.
// ------------------------------------------------------------------------------------------------
template <int NQH>
struct TileRegs {
    f32x4 a[8];
    f32x4 bn;
};

template <int NQH, int KCAP, int MODE>
__global__ __launch_bounds__(kScanThreads, 2) void scan_kernel(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform
    const int r = lane & 15;
    const int g = lane >> 4;

    // query fragments (B operand): qf[h][t][i] = Q[16 h + r][16 t + 4 g + i]
    f32x4 qf[NQH][8];
    float qn[NQH];
    float tau[NQH];
#pragma unroll
    for (int h = 0; h < NQH; ++h) {
        const float* qrow = p.q + (h * 16 + r) * kDim + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; ++t) qf[h][t] = *reinterpret_cast<const f32x4*>(qrow + 16 * t);
        qn[h] = p.qnorm[h * 16 + r];
        tau[h] = p.tau0 ? p.tau0[h * 16 + r] : VS_INF;
    }

    float ld[NQH][KCAP];
    int li[NQH][KCAP];
#pragma unroll
    for (int h = 0; h < NQH; ++h)
#pragma unroll
        for (int j = 0; j < KCAP; ++j) {
            ld[h][j] = VS_INF;
            li[h][j] = -1;
        }

    const int64_t n_rows = p.row_end - p.row_begin;
    const int tiles_total = (int)((n_rows + kTileRows - 1) / kTileRows);
    const int tile0 = blockIdx.x * p.tiles_per_wg;
    const int tile1 = min(tile0 + p.tiles_per_wg, tiles_total);
    const int64_t last_row = p.row_end - 1;

    auto load_tile = [&](int t, TileRegs<NQH>& T) {
        const int64_t row0 = p.row_begin + (int64_t)t * kTileRows;
        const int64_t row = min(row0 + r, last_row);  // tail rows re-read the last row, masked below
        const float* src = p.base + row * kDim + 4 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) T.a[c] = *reinterpret_cast<const f32x4*>(src + 16 * c);
        T.bn = *reinterpret_cast<const f32x4*>(p.bnorm + row0 + 4 * g);  // bnorm is padded by 16
    };

    auto compute_tile = [&](int t, const TileRegs<NQH>& T) {
        f32x4 acc[NQH];
#pragma unroll
        for (int h = 0; h < NQH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < NQH; ++h)
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(T.a[c][i], qf[h][c][i], acc[h], 0, 0, 0);

        const int64_t rbase = p.row_begin + (int64_t)t * kTileRows + 4 * g;  // this lane's first row
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            float d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // cpu_baseline.cpp:241  dist = qn + bn - 2*dot  (gcc contracts to fnmadd(2, dot, qn+bn))
                const float l2 = fmaf(-2.0f, acc[h][j], qn[h] + T.bn[j]);
                const float v = p.metric ? -acc[h][j] : l2;
                d[j] = (rbase + j <= last_row) ? v : VS_INF;
            }
            if (MODE == kModeTopK) {
                const float dmin = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
                if (dmin < tau[h]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (d[j] < tau[h]) {
                            list_insert<KCAP>(ld[h], li[h], d[j], (int)(rbase + j) + p.id_offset);
                            tau[h] = fminf(tau[h], ld[h][KCAP - 1]);
                        }
                }
            } else {
                const int qidx = h * 16 + r;
                if (qidx < p.nq_valid) {
                    float* dst = p.store + (int64_t)qidx * p.store_ld + (rbase - p.row_begin);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (rbase + j <= last_row) dst[j] = d[j];
                }
            }
        }
    };

    // Two register tile buffers, statically named (runtime-indexed vector arrays would spill).
    // Loads are issued unconditionally (the tile index is clamped, so a wave's last one or two
    // prefetches re-read a valid tile and are discarded): with a load under a branch hipcc can no
    // longer count the queue and falls back to s_waitcnt vmcnt(0), which serialises load and MFMA.
    TileRegs<NQH> TA, TB;
    const int tlast = max(tile1 - 1, 0);
    int t = tile0 + wave;
    load_tile(min(t, tlast), TA);
    for (; t < tile1; t += 2 * kScanWaves) {
        load_tile(min(t + kScanWaves, tlast), TB);
        compute_tile(t, TA);
        load_tile(min(t + 2 * kScanWaves, tlast), TA);
        if (t + kScanWaves < tile1) compute_tile(t + kScanWaves, TB);
    }

    if (MODE != kModeTopK) return;

    // ---- workgroup merge: 32 lane lists per query (8 waves x 4 lane groups) -> one sorted list ----
    // LDS image: [query][list s = wave*4+g][KCAP] for dist and id.
    float* sd = reinterpret_cast<float*>(smem);
    int* si = reinterpret_cast<int*>(smem + (size_t)NQH * 16 * 32 * KCAP * sizeof(float));
#pragma unroll
    for (int h = 0; h < NQH; ++h) {
        const int base_off = ((h * 16 + r) * 32 + (wave * 4 + g)) * KCAP;
#pragma unroll
        for (int j = 0; j < KCAP; ++j) {
            sd[base_off + j] = ld[h][j];
            si[base_off + j] = li[h][j];
        }
    }
    __syncthreads();

    // each half-wave merges one query: lane s of the half walks list s
    const int half = lane >> 5;
    const int s = lane & 31;
    constexpr int NQ = NQH * 16;
    for (int qq = wave * 2 + half; qq < NQ; qq += 2 * kScanWaves) {
        const int lo = (qq * 32 + s) * KCAP;
        int ptr = 0;
        float hd = sd[lo];
        int hi = si[lo];
        float* od = p.part_d + ((int64_t)blockIdx.x * kMaxBatch + qq) * KCAP;
        int32_t* oi = p.part_i + ((int64_t)blockIdx.x * kMaxBatch + qq) * KCAP;
        for (int round = 0; round < KCAP; ++round) {
            float bd = hd;
            int bi = hi;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) {
                const float od2 = __shfl_xor(bd, m);
                const int oi2 = __shfl_xor(bi, m);
                if (lex_lt(od2, oi2, bd, bi)) {
                    bd = od2;
                    bi = oi2;
                }
            }
            if (s == 0) {
                od[round] = bd;
                oi[round] = bi;
            }
            if (bi >= 0 && hi == bi && hd == bd) {  // the owner of the winner advances
                ++ptr;
                hd = ptr < KCAP ? sd[lo + ptr] : VS_INF;
                hi = ptr < KCAP ? si[lo + ptr] : -1;
            }
        }
    }
}

template <int NQH, int KCAP, int MODE>
static hipError_t launch_scan_t(const ScanParams& p, int grid, hipStream_t s) {
    size_t lds = (MODE == kModeTopK) ? (size_t)NQH * 16 * 32 * KCAP * 8 : 0;
    auto kfn = scan_kernel<NQH, KCAP, MODE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(kScanThreads), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanParams& p, int grid, int kcap, int nqh, int mode, hipStream_t s) {
    if (mode == kModeStore) {
        return nqh == 1 ? launch_scan_t<1, 8, kModeStore>(p, grid, s) : launch_scan_t<2, 8, kModeStore>(p, grid, s);
    }
    if (kcap == 8) {
        return nqh == 1 ? launch_scan_t<1, 8, kModeTopK>(p, grid, s) : launch_scan_t<2, 8, kModeTopK>(p, grid, s);
    }
    if (kcap == 16) {
        return nqh == 1 ? launch_scan_t<1, 16, kModeTopK>(p, grid, s) : launch_scan_t<2, 16, kModeTopK>(p, grid, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// Merge of sorted lists.  One 256-thread workgroup per query; thread t owns lists t, t+256, ...
// (LPT of them).  kout rounds of a workgroup-wide lexicographic argmin over the list heads.
// part_i == nullptr means "the id of entry (g, j) is g*kin + j" (used to pick probes out of a
// score matrix: G = nlist lists of length 1).
// ------------------------------------------------------------------------------------------------
constexpr int kMergeTrack = 256;  // leading outputs kept on chip for the tie flag / seed threshold
struct MergeLayout {
    int64_t stride_g, stride_q;
};

template <int LPT>
__global__ __launch_bounds__(256) void merge_kernel(const MergeParams p, const MergeLayout L) {
    __shared__ float wbd[4];
    __shared__ int wbi[4];
    __shared__ float outd[kMergeTrack];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    float hd[LPT];
    int hi[LPT];
    int ptr[LPT];
    auto fetch = [&](int g, int j, float& d, int& id) {
        if (g < p.G && j < p.kin) {
            const int64_t off = (int64_t)g * L.stride_g + (int64_t)q * L.stride_q + j;
            d = p.part_d[off];
            id = p.part_i ? p.part_i[off] : (g * p.kin + j);
            if (d != d) { d = VS_INF; id = -1; }  // NaN never wins
        } else {
            d = VS_INF;
            id = -1;
        }
    };
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
        ptr[u] = 0;
        fetch(tid + 256 * u, 0, hd[u], hi[u]);
    }

    for (int round = 0; round < p.kout; ++round) {
        float bd = hd[0];
        int bi = hi[0];
#pragma unroll
        for (int u = 1; u < LPT; ++u)
            if (lex_lt(hd[u], hi[u], bd, bi)) {
                bd = hd[u];
                bi = hi[u];
            }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const float od2 = __shfl_xor(bd, m);
            const int oi2 = __shfl_xor(bi, m);
            if (lex_lt(od2, oi2, bd, bi)) {
                bd = od2;
                bi = oi2;
            }
        }
        if (lane == 0) {
            wbd[wave] = bd;
            wbi[wave] = bi;
        }
        __syncthreads();
        bd = wbd[0];
        bi = wbi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (lex_lt(wbd[w], wbi[w], bd, bi)) {
                bd = wbd[w];
                bi = wbi[w];
            }
        __syncthreads();
        if (tid == 0) {
            if (round < kMergeTrack) outd[round] = bd;
            if (p.out_d) p.out_d[(int64_t)q * p.kout + round] = bd;
            if (p.out_i) p.out_i[(int64_t)q * p.kout + round] = (bi >= 0 && p.id_map) ? p.id_map[bi] : bi;
        }
        if (bi >= 0) {
#pragma unroll
            for (int u = 0; u < LPT; ++u)
                if (hi[u] == bi && hd[u] == bd) {
                    ++ptr[u];
                    fetch(tid + 256 * u, ptr[u], hd[u], hi[u]);
                }
        }
    }
    if (tid == 0) {
        const int n = p.kout < kMergeTrack ? p.kout : kMergeTrack;
        if (p.flags) {
            int f = 0;
            for (int i = 0; i + 1 < n; ++i)
                if (outd[i] == outd[i + 1] && outd[i] < VS_INF) f = 1;
            p.flags[q] = f;
        }
        if (p.tau_out) {
            const float kth = outd[n - 1];
            p.tau_out[q] = kth < VS_INF ? next_up(kth) : VS_INF;
        }
    }
}

hipError_t launch_merge_layout(const MergeParams& p, int64_t stride_g, int64_t stride_q, hipStream_t s) {
    if (p.kout < 1 || p.G < 1 || p.nq < 1) return hipErrorInvalidValue;
    MergeLayout L{stride_g, stride_q};
    const int lpt = (p.G + 255) / 256;
    if (lpt <= 1) hipLaunchKernelGGL(merge_kernel<1>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 2) hipLaunchKernelGGL(merge_kernel<2>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 4) hipLaunchKernelGGL(merge_kernel<4>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 8) hipLaunchKernelGGL(merge_kernel<8>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 16) hipLaunchKernelGGL(merge_kernel<16>, dim3(p.nq), dim3(256), 0, s, p, L);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_merge(const MergeParams& p, hipStream_t s) {
    // scan partial layout: [G][nq_stride][kin]
    return launch_merge_layout(p, (int64_t)p.nq_stride * p.kin, p.kin, s);
}

hipError_t launch_pick_probes(const float* scores, int64_t ld, int B, int nlist, int nprobe,
                              int32_t* probes, hipStream_t s) {
    MergeParams p{};
    p.part_d = scores;
    p.part_i = nullptr;
    p.G = nlist;
    p.kin = 1;
    p.nq = B;
    p.kout = nprobe;
    p.out_d = nullptr;
    p.out_i = probes;
    return launch_merge_layout(p, 1, ld, s);
}

// ------------------------------------------------------------------------------------------------
// Row norms in the reference's order: 8 FMA lanes over v[8 i + j], then r0+r1+...+r7 left to
// right, then the scalar tail (cpu_baseline.cpp:95-114).  8 threads per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ v, int64_t rows, int dim,
                                                         float* __restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = gid >> 3;
    const int j = (int)(gid & 7);
    const bool ok = row < rows;
    const float* src = v + (ok ? row : 0) * dim;
    float acc = 0.f;
    const int d8 = dim & ~7;
    for (int i = 0; i < d8; i += 8) {
        const float x = src[i + j];
        acc = fmaf(x, x, acc);
    }
    const int lane = threadIdx.x & 63;
    const int b = lane & ~7;
    float sum = __shfl(acc, b);
#pragma unroll
    for (int u = 1; u < 8; ++u) sum = sum + __shfl(acc, b + u);
    for (int i = d8; i < dim; ++i) sum = fmaf(src[i], src[i], sum);
    if (ok && j == 0) out[row] = sum;
}

hipError_t launch_row_sqnorm(const float* v, int64_t rows, int dim, float* out, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t threads = rows * 8;
    const int grid = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3(grid), dim3(256), 0, s, v, rows, dim, out);
    return hipGetLastError();
}

// Pad to 32 x 128 with zero rows (main.cpp:206-211, main_ivf.cpp:149-153) + norms, one workgroup.
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, int B,
                                                           float* __restrict__ qpad, float* __restrict__ qnorm) {
    const int row = threadIdx.x >> 3;  // 32 rows x 8 lanes
    const int j = threadIdx.x & 7;
    float acc = 0.f;
    for (int i = 0; i < kDim; i += 8) {
        const float x = row < B ? q[row * kDim + i + j] : 0.f;
        qpad[row * kDim + i + j] = x;
        acc = fmaf(x, x, acc);
    }
    const int lane = threadIdx.x & 63;
    const int b = lane & ~7;
    float sum = __shfl(acc, b);
#pragma unroll
    for (int u = 1; u < 8; ++u) sum = sum + __shfl(acc, b + u);
    if (j == 0) qnorm[row] = sum;
}

hipError_t launch_prep_queries(const float* q, int B, float* qpad, float* qnorm, hipStream_t s) {
    hipLaunchKernelGGL(prep_queries_kernel, dim3(1), dim3(256), 0, s, q, B, qpad, qnorm);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// IVF list scan.  One 256-thread workgroup per (query, probe) item; the four waves take
// alternating groups of 8 rows.  8 lanes share a row: every wave-instruction reads 8 rows x 128
// contiguous bytes (whole cache lines), four instructions cover the 512-byte rows, nothing is
// staged through LDS because no byte is used twice.  The 8 partial sums are folded with DPP
// (quad_perm xor 1, xor 2, row_half_mirror).  The running top-k of a wave is one sorted list with
// entry j living in lane j; inserting is a ballot + popcount + row_shr:1 shift.
// ------------------------------------------------------------------------------------------------
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
template <int KCAP>
__device__ __forceinline__ void wave_list_insert(float& ld, int& li, float cd, int ci, int lane) {
    const bool before = lane < KCAP && lex_lt(ld, li, cd, ci);
    const int pos = __popcll(__ballot(before));
    const float sd = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ld), 0x111, 0xF, 0xF, false));
    const int si = __builtin_amdgcn_update_dpp(0, li, 0x111, 0xF, 0xF, false);
    if (lane < KCAP) {
        if (lane == pos) { ld = cd; li = ci; }
        else if (lane > pos) { ld = sd; li = si; }
    }
}

template <int KCAP>
__global__ __launch_bounds__(256) void ivf_scan_kernel(const IvfScanParams p) {
    __shared__ float sld[4][KCAP];
    __shared__ int sli[4][KCAP];
    const int item = blockIdx.x;
    const int b = item / p.nprobe;
    const int pr = item - b * p.nprobe;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = p.probes[b * p.nprobe + pr];
    int start = 0, end = 0;
    if (c >= 0 && (!p.owned || p.owned[c])) {
        start = p.offsets[c];
        end = p.offsets[c + 1];
    }
    if (threadIdx.x == 0 && p.cand_count && end > start)
        atomicAdd(p.cand_count, (unsigned long long)(end - start));

    const int rr = lane >> 3, s8 = lane & 7;
    f32x4 qf[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) qf[m] = *reinterpret_cast<const f32x4*>(p.q + b * kDim + 4 * (s8 + 8 * m));
    const float qn = p.qnorm[b];

    float ld = VS_INF;
    int li = -1;
    float tau = VS_INF;

    for (int row0 = start + wave * 8; row0 < end; row0 += 32) {
        const int row = row0 + rr;
        const bool valid = row < end;
        const int rowc = valid ? row : end - 1;
        const float* src = p.vecs + (int64_t)rowc * kDim + 4 * s8;
        f32x4 v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = *reinterpret_cast<const f32x4*>(src + 32 * m);
        const float vn = p.vnorm[rowc];
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = fmaf(v[m][i], qf[m][i], acc);
        acc = dpp_add_xor1(acc);
        acc = dpp_add_xor2(acc);
        acc = dpp_add_half_mirror(acc);
        const float d = p.metric ? -acc : fmaf(-2.0f, acc, qn + vn);
        bool pass = valid && s8 == 0 && d < tau;
        unsigned long long mask = __ballot(pass);
        while (mask) {
            const int src_lane = __builtin_ctzll(mask);
            const float cd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), src_lane));
            const int ci = __builtin_amdgcn_readlane(row, src_lane);
            wave_list_insert<KCAP>(ld, li, cd, ci, lane);
            tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ld), KCAP - 1));
            pass = pass && lane != src_lane && d < tau;
            mask = __ballot(pass);
        }
    }

    if (lane < KCAP) {
        sld[wave][lane] = ld;
        sli[wave][lane] = li;
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < 4; ++w)
            for (int j = 0; j < KCAP; ++j) {
                const float cd = sld[w][j];
                const int ci = sli[w][j];
                if (!(cd < tau) && !(cd == tau)) break;  // lists are sorted; NaN never stored
                if (ci >= 0 && lex_lt(cd, ci,
                                      __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ld), KCAP - 1)),
                                      __builtin_amdgcn_readlane(li, KCAP - 1))) {
                    wave_list_insert<KCAP>(ld, li, cd, ci, lane);
                    tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ld), KCAP - 1));
                }
            }
        if (lane < KCAP) {
            const int64_t o = ((int64_t)b * p.nprobe + pr) * KCAP + lane;
            p.part_d[o] = ld;
            p.part_i[o] = li;
        }
    }
}

hipError_t launch_ivf_scan(const IvfScanParams& p, hipStream_t s) {
    const int grid = p.B * p.nprobe;
    if (grid <= 0) return hipSuccess;
    if (p.kcap == 8) hipLaunchKernelGGL(ivf_scan_kernel<8>, dim3(grid), dim3(256), 0, s, p);
    else if (p.kcap == 16) hipLaunchKernelGGL(ivf_scan_kernel<16>, dim3(grid), dim3(256), 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace vs
