// vs_ivf.hip -- IVFIndex::searchBatch (IVFIndex.cpp:640-859) on the GPU: the wide list-major pipeline over launch groups
// (coarse MFMA -> pick -> bounds || plan -> scan -> rank) and the query-major fallback (pick_probes + ivf_scan_kernel); see vs_kernels.h.
#include "vs_kernels.h"
#include "vs_dev.h"
#include "vs_sink.h"
#include "vs_merge.h"
#include <type_traits>
#include <algorithm>

namespace vs {

// ------------------------------------------------------------------------------------------------
// Probe selection (std::nth_element at IVFIndex.cpp:711, made deterministic): one wave per query.
// Each lane sorts its 16 strided scores in registers (bitonic network, static indices), parks the
// sorted run in LDS and the wave then pops nprobe winners with shuffle argmin rounds.
// ------------------------------------------------------------------------------------------------
constexpr int kPickVPL = 16;  // values per lane -> nlist <= 1024

__global__ __launch_bounds__(256) void pick_probes_kernel(const float* __restrict__ scores, int64_t ld, int B, int nlist,
                                                          int nprobe, int32_t* __restrict__ probes) {
    __shared__ float sv[4][64][kPickVPL + 1];
    __shared__ int si[4][64][kPickVPL + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= B) return;
    float v[kPickVPL];
    int id[kPickVPL];
#pragma unroll
    for (int i = 0; i < kPickVPL; ++i) {
        const int c = i * 64 + lane;
        float x = c < nlist ? scores[(int64_t)q * ld + c] : VS_INF;
        const bool ok = c < nlist && x == x;
        v[i] = ok ? x : VS_INF;
        id[i] = ok ? c : 0x7fffffff;
    }
#pragma unroll
    for (int k = 2; k <= kPickVPL; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < kPickVPL; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const bool sw = up ? lex_lt(v[l], id[l], v[i], id[i]) : lex_lt(v[i], id[i], v[l], id[l]);
                    const float tv = sw ? v[l] : v[i];
                    const int ti = sw ? id[l] : id[i];
                    v[l] = sw ? v[i] : v[l];
                    id[l] = sw ? id[i] : id[l];
                    v[i] = tv;
                    id[i] = ti;
                }
            }
#pragma unroll
    for (int i = 0; i < kPickVPL; ++i) {
        sv[wave][lane][i] = v[i];
        si[wave][lane][i] = id[i];
    }
    int ptr = 0;
    float hd = v[0];
    int hi = id[0];
    for (int round = 0; round < nprobe; ++round) {
        float bd;
        int bi;
        wave_lexmin(hd, hi, bd, bi);
        if (lane == 0) probes[(int64_t)q * nprobe + round] = bi == 0x7fffffff ? -1 : bi;
        if (hi == bi && bi != 0x7fffffff) {
            ++ptr;
            hd = ptr < kPickVPL ? sv[wave][lane][ptr] : VS_INF;
            hi = ptr < kPickVPL ? si[wave][lane][ptr] : 0x7fffffff;
        }
    }
}

hipError_t launch_pick_probes(const float* scores, int64_t ld, int B, int nlist, int nprobe,
                              int32_t* probes, hipStream_t s) {
    if (nlist <= 64 * kPickVPL) {
        hipLaunchKernelGGL(pick_probes_kernel, dim3((B + 3) / 4), dim3(256), 0, s, scores, ld, B, nlist, nprobe, probes);
        return hipGetLastError();
    }
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
// IVF coarse stage + probe selection in one launch (IVFIndex.cpp:654-666 centroid scores, :697-723
// top-nprobe): one 256-thread workgroup per query.  Distances to all centroids with the same
// 8-lanes-per-row dot product as the list scan (centroids are L2 resident), then selection without
// sorting rounds: the nprobe-th smallest of the 256 per-thread minima bounds the answer, the few
// scores under that bound are compacted and ranked by counting (every candidate counts how many
// others precede it in (dist, id) order and writes itself to that slot).  Deterministic.
// ------------------------------------------------------------------------------------------------

// multi-batch launches: advance a per-batch pointer to batch blockIdx.y's copy (see IvfMulti)
template <class T>
__device__ __forceinline__ T* mb_adv(T* ptr, long long bytes) {
    return ptr ? reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(ptr)) + bytes) : ptr;
}

// Coarse stage, part 1 (IVFIndex.cpp:654-666, the reference's NPU matmul): scores[q][c] = ||q||^2 + ||c||^2 - 2 q.c for
// ALL queries of a launch group against all centroids, as one MFMA contraction: grid (nlist / 64, n_batches), four
// waves per workgroup, wave w owns the 16-centroid tile 4 blockIdx.x + w as the A operand (fragments straight from
// global memory: the centroids are L2 resident) and the batch's <= 32 queries as two 16-column B operands -- the same
// v_mfma_f32_16x16x4_f32 chain and epilogue as the brute-force scan, so a centroid score is the number that scan
// would produce.  1024 x 1024 x 128 per group of 32 batches: a few microseconds.
__global__ __launch_bounds__(256) void ivf_coarse_mfma_kernel(const float* __restrict__ q_all, int B, const float* __restrict__ cents,
                                                             const float* __restrict__ cnorm, int nlist, int metric,
                                                             float* __restrict__ scores_all, int ld, IvfMulti mb, IvfGroup grp, int prep_only,
                                                             int nct) {
    // A workgroup keeps its batch's queries (LDS, then both 16-column B operands in registers, with their norms) for `nct`
    // tiles of 64 centroids in a row: blockIdx.x = group of nct tiles.  One tile per workgroup read 48 KB for half a MFLOP
    // of MFMA work and worked out the same 32 norms in every one of the 16 workgroups of a batch: 8192 queries cost 44 us,
    // three times what the arithmetic takes.  The next tile's centroids are in flight while this one is scored.
    const int y = blockIdx.y;
    const float* q = mb_adv(q_all, (long long)y * mb.q);
    float* scores = mb_adv(scores_all, (long long)y * mb.slab);
    const int n_tiles = (nlist + 63) >> 6;
    const int t0 = (int)blockIdx.x * nct, t1 = min(n_tiles, t0 + nct);
#ifdef VS_STAMPS
#define CO_STAMP(i) do { if (grp.dbg && threadIdx.x == 0 && !prep_only) grp.dbg[(20480 + (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff); } while (0)
#else
#define CO_STAMP(i)
#endif
    CO_STAMP(0);
    __shared__ float qn_s[kMaxBatch];
    // Both operands go through LDS: the 64 centroids of a tile and the batch's queries are read from global memory as
    // whole rows (a wave instruction = 1 KB in one piece) and the MFMA fragments are cut out of LDS.  Read as fragments
    // straight from memory, every load instruction touched 64 separate 16-byte pieces 512 bytes apart, and the address
    // unit, not the arithmetic, set the kernel's time (13 us).  Rows are 132 floats apart in LDS: fragment reads of
    // 8 neighbouring rows then fall into different banks.
    constexpr int LD = kDim + 4;
    // Dynamic LDS, kCoarseLds bytes: every wave owns two 8 KB landing buffers for ITS 16 centroids of a tile (below); the
    // queries' staging place in the prologue lies over them (the queries are in registers before the first tile lands).
    extern __shared__ __attribute__((aligned(16))) char co_smem[];
    float* const q_s = reinterpret_cast<float*>(co_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    {
        f32x4 vq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;  // float4 (row, column) of the 32 x 32 tile
            vq[i] = *reinterpret_cast<const f32x4*>(q + min(idx >> 5, B - 1) * kDim + 4 * (idx & 31));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<f32x4*>(q_s + (idx >> 5) * LD + 4 * (idx & 31)) = (idx >> 5) < B ? vq[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    f32x4 qf[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 8; ++c) qf[h][c] = *reinterpret_cast<const f32x4*>(q_s + (h * 16 + r) * LD + 16 * c + 4 * g);  // (rows >= B: zeros)
    {   // ||q||^2 in the reference's AVX2 order (cpu_baseline.cpp:95-114): 8 lanes per query
        const int row = tid >> 3, j = tid & 7;
        float acc = 0.f;
        if (row < B) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float x = q_s[row * LD + 8 * i + j];
                acc = fmaf(x, x, acc);
            }
        }
        const int b8 = lane & ~7;
        float sum = __shfl(acc, b8);
#pragma unroll
        for (int u = 1; u < 8; ++u) sum = sum + __shfl(acc, b8 + u);
        if (j == 0) qn_s[row] = row < B ? sum : 0.f;
        // wide pipeline: one of the batch's workgroups also writes the queries as bytes, their constant terms and the
        // batch's "byte valued" verdict (what seed_qnorm_kernel does for the brute-force scans); a thread converts 16
        // neighbouring components and stores them as one 16-byte word
        if (grp.w_q8 != nullptr && y % (int)gridDim.x == (int)blockIdx.x) {  // (workgroup-uniform)
            const int64_t qslot = (int64_t)y * kMaxBatch + row;
            int part = 0;
            bool q_ok = true;
            int w[4] = {0, 0, 0, 0};
            if (row < B) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(q_s + row * LD + 16 * j + 4 * v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int xi = (int)x[e];
                        q_ok = q_ok && ((float)xi == x[e]) && xi >= 0 && xi <= 255;
                        part += xi - 128;
                        w[v] |= ((xi - 128) & 0xff) << (8 * e);
                    }
                }
            }
            *reinterpret_cast<int4*>(grp.w_q8 + qslot * kDim + 16 * j) = make_int4(w[0], w[1], w[2], w[3]);  // (padding queries: 0)
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);
            part += __shfl_xor(part, 4);
            if (j == 0) {
                grp.w_qnorm[qslot] = row < B ? sum : 0.f;
                grp.w_qterm[qslot] = (int)sum - 256 * part - 4194304;
            }
            // (written as 0 or 1, never left over from the previous group: nobody has to clear it)
            const int bad = __syncthreads_or(q_ok ? 0 : 1);  // (workgroup-uniform branch: every thread is here)
            if (tid == 0) grp.w_invalid[y] = bad ? 1 : 0;
            if (tid == 0 && y == 0 && grp.w_overflow) grp.w_overflow[0] = 0;  // the previous group's verdict has been read
            if (tid == 0 && y == 0 && grp.w_glist) grp.w_glist[0] = 0;        // ... and so has its list of left-over queries
        }
    }
    __syncthreads();
    CO_STAMP(1);
    if (prep_only) return;  // (sharded front half: the queries of ALL slices are prepared on every rank, scored only on their own)
    const float qn[2] = {qn_s[r], qn_s[16 + r]};
    // A wave's A fragments of a tile -- its 16 centroids, chunk c8: the 16 bytes C[16 wave + r][16 c8 + 4 g ..] of every lane --
    // land in the wave's own LDS buffer by LDS-DMA, in fragment order (instruction c8 fills buffer + 1024 c8 + 16 lane), one
    // tile ahead: no staging registers, no workgroup barrier in the loop, every wave at its own pace.  (Through registers and
    // a shared 64-row tile the loop was stage 0.5 + fetch 0.9 + MFMA 1.5 + epilogue 0.8 us per tile and wave, with two
    // barriers.)  One wait per tile, at the top, for everything issued during the tile before: the next tile's DMA, its norms,
    // the previous tile's score stores -- all of them had a whole tile's MFMAs to complete.
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(co_smem + wave * 16384));
    const unsigned voff0 = (unsigned)((16 * wave + r) * (kDim * 4) + 16 * g);
    const char* cbytes = reinterpret_cast<const char*>(cents);
    auto dma_tile = [&](const int t, const int par) __attribute__((always_inline)) {
        const unsigned voff = voff0 + (unsigned)t * (64u * kDim * 4u);
        const unsigned dst = lds_w + (unsigned)par * 8192u;
        // (no instruction offset: it would move the LDS address as well as the global one)
#define VS_CDMA(c8) asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff + 64u * (c8)), "s"(cbytes), "n"((c8) * 1024) : "memory", "scc")
        VS_CDMA(0); VS_CDMA(1); VS_CDMA(2); VS_CDMA(3); VS_CDMA(4); VS_CDMA(5); VS_CDMA(6); VS_CDMA(7);
#undef VS_CDMA
    };
    f32x4 d_out[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // the wave's scores of a tile: 16 centroids x 2 x 16 queries
    auto store_scores = [&](int t) {
#ifdef VS_CO_NOSTORE  // (diagnostic variant: what the score stores cost)
        if (t >= 0) return;
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (h * 16 + r < B) *reinterpret_cast<f32x4*>(scores + (int64_t)(h * 16 + r) * ld + t * 64 + wave * 16 + 4 * g) = d_out[h];
    };
    dma_tile(t0, 0);
    f32x4 cn_nx = *reinterpret_cast<const f32x4*>(cnorm + t0 * 64 + wave * 16 + 4 * g);  // (padded by 64)
    int par = 0;
    for (int t = t0; t < t1; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4 cn = cn_nx;
        asm volatile("" ::"v"(cn));  // (the compiler's own wait for the norms belongs HERE, not in front of the epilogue behind the next DMA)
        if (t + 1 < t1) {  // (workgroup-uniform)
            dma_tile(t + 1, par ^ 1);
            cn_nx = *reinterpret_cast<const f32x4*>(cnorm + (t + 1) * 64 + wave * 16 + 4 * g);
        }
        if (t > t0) store_scores(t - 1);  // (a tile late: it has this tile's MFMAs to complete)
        if (t == t0 + 1) CO_STAMP(3);
        const f32x4* ab = reinterpret_cast<const f32x4*>(co_smem + wave * 16384 + par * 8192) + lane;
        const int row0 = t * 64 + wave * 16;  // this wave's 16 centroids
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (B > 16) {  // workgroup-uniform
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 a = ab[c * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], qf[0][c][i], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], qf[1][c][i], acc[1], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 a = ab[c * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], qf[0][c][i], acc[0], 0, 0, 0);
            }
        }
        if (t == t0 + 1) CO_STAMP(5);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = metric ? -acc[h][j] : fmaf(-2.0f, acc[h][j], qn[h] + cn[j]);
                if (!(v == v) || row0 + 4 * g + j >= nlist) v = VS_INF;  // NaN never wins; rows past nlist are padding
                d_out[h][j] = v;
            }
        }
        if (t == t0) CO_STAMP(2);
        if (t == t0 + 1) CO_STAMP(6);
        par ^= 1;
    }
    store_scores(t1 - 1);
    CO_STAMP(7);
}

constexpr int kCoarseLds = 4 * 16384;  // four waves x two landing buffers of 8 KB

// tiles of 64 centroids a workgroup of the coarse kernel takes in a row: as many as leave the launch 512 workgroups (two per
// CU), 8 at most
// (the coarse kernel's dynamic LDS: 64 KB + its static words pass the default limit)
static hipError_t coarse_lds_attr() {
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_coarse_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kCoarseLds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    return hipSuccess;
}

static int coarse_nct(int tiles, int n_batches) {
    static const int forced = getenv("VSEARCH_COARSE_NCT") ? atoi(getenv("VSEARCH_COARSE_NCT")) : 0;  // (tuning knob)
    return forced > 0 ? forced : std::max(1, std::min(8, tiles * n_batches / 512));
}

// Coarse stage, part 2 (std::nth_element at IVFIndex.cpp:711, made deterministic: ascending (dist, id)): one 256-thread
// workgroup per query reads its row of scores and selects without sorting rounds: the nprobe-th smallest of the 256
// per-thread minima bounds the answer, the few scores under that bound are compacted and ranked by counting (every
// candidate counts how many others precede it and writes itself to that slot).  Then the query's window offsets in the
// candidate array (grouping tables of the list-major scan).
template <int EPT>
__global__ __launch_bounds__(256) void ivf_pick_kernel(const float* __restrict__ scores, int ld, int nlist, int nprobe,
                                                       int32_t* __restrict__ probes, IvfGroup grp) {
    {   // multi-batch launch: this workgroup's batch
        const long long y = blockIdx.y;
        scores = mb_adv(scores, y * grp.mb.slab);
        probes = mb_adv(probes, y * grp.mb.probes);
    }
    // (score, list) pairs are compared as ONE 64-bit key (ordered float bits << 32 | list): the counting loops below
    // then read two keys per 16-byte LDS load and cost one compare each
    typedef unsigned long long u64;
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    __shared__ int s_probe[256];
    __shared__ __attribute__((aligned(16))) u64 mnk[256];
    __shared__ __attribute__((aligned(16))) u64 cdk[256 * EPT + 2];
    __shared__ u64 s_tk;
    __shared__ int s_cnt;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
#ifdef VS_STAMPS
#define PICK_STAMP(i) do { if (grp.dbg && tid == 0) grp.dbg[((int)blockIdx.y * 32 + b) * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff); } while (0)
#else
#define PICK_STAMP(i)
#endif
    PICK_STAMP(0);
    if (tid == 0) {
        s_cnt = 0;
        s_tk = ~0ull;
    }
    u64 mine[EPT];
    u64 mk = ~0ull;
    const float* sc = scores + (int64_t)b * ld;
    float v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int idx = tid + 256 * e;
        v[e] = idx < nlist ? sc[idx] : VS_INF;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int idx = tid + 256 * e;
        mine[e] = idx < nlist ? (((u64)f32_ordered(v[e]) << 32) | (unsigned)idx) : ~0ull;
        mk = mine[e] < mk ? mine[e] : mk;
    }
    mnk[tid] = mk;
    __syncthreads();
    PICK_STAMP(1);
    if (tid < 64) {
        // the nprobe-th smallest score among the 256 per-thread minima (they are 256 different lists, so at least nprobe
        // lists score that or less), bit by bit from the top: the largest x with fewer than nprobe minima below x.  One
        // wave, 32 rounds of 4 compares and 4 scalar popcounts (ranking every minimum against every other one cost 4 us).
        unsigned hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hi[i] = (unsigned)(mnk[tid + 64 * i] >> 32);
        // (the bound only has to be AT LEAST that score: the search stops after the top kPickBits bits and the rest is
        // filled with ones -- a bound 2^-9 of the score too high lets a list or two more into the ranking and saves
        // fourteen of the 32 rounds, which one wave runs while the other three wait)
        constexpr int kPickBits = 18;
        unsigned x = 0;
#pragma unroll 6
        for (int bit = 31; bit >= 32 - kPickBits; --bit) {
            const unsigned t = x | (1u << bit);
            int below = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) below += __popcll(__ballot(hi[i] < t));
            if (below < nprobe) x = t;  // wave-uniform
        }
        x |= (1u << (32 - kPickBits)) - 1;
        if (tid == 0) s_tk = ((u64)x << 32) | 0xffffffffull;  // every list scoring x or less is a candidate (ties included)
    }
    __syncthreads();
    PICK_STAMP(2);
    {
        const u64 tk = s_tk;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (mine[e] != ~0ull && mine[e] <= tk) {
                const int pos = atomicAdd(&s_cnt, 1);
                cdk[pos] = mine[e];
            }
    }
    __syncthreads();
    const int C = s_cnt;
    if (tid == 0) {  // pad to an even count for the paired reads
        cdk[C] = ~0ull;
        cdk[C + 1] = ~0ull;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const u64 key = cdk[c];
        int rank = 0;
        const u64x2* p2 = reinterpret_cast<const u64x2*>(cdk);
        for (int j = 0; j < (C + 1) / 2; ++j) {
            const u64x2 w = p2[j];
            rank += (w.x < key ? 1 : 0) + (w.y < key ? 1 : 0);
        }
        if (rank < nprobe) {
            const int id = (int)(unsigned)(key & 0xffffffffull);
            probes[(int64_t)b * nprobe + rank] = id;
            s_probe[rank] = id;
        }
    }
    for (int c = C + tid; c < nprobe; c += 256) {
        probes[(int64_t)b * nprobe + c] = -1;
        s_probe[c] = -1;
    }
    PICK_STAMP(3);
    __syncthreads();
#ifdef VS_PICK_NOTABLES  // (diagnostic variant: what the table entries -- 34 returning atomics per query -- cost)
    if (nlist > 0) return;
#endif
    if (grp.w_tq && tid < 64) {
        // the bound's tables (see ivf_bounds_list_body): the query enters itself with the first two of its probed lists
        // that hold rows, as segment 0 and 1 (looked for among the nearest 64)
        const int c = tid < nprobe ? s_probe[tid] : -1;
        const bool rows = c >= 0 && grp.t_offsets[c + 1] > grp.t_offsets[c];
        const unsigned long long mask = __ballot(rows);
        const int seg = __builtin_popcountll(mask & ((1ull << tid) - 1));  // lists with rows before this one
        const int qg = (int)blockIdx.y * kMaxBatch + b;
        if (rows && seg < kBoundSegs) {
            const int slot = atomicAdd(grp.w_tcnt + (int64_t)c * kIvfWideCntStride + 1, 1);  // < w_tq_cap: once per query
            grp.w_tq[(int64_t)c * grp.w_tq_cap + slot] = qg | seg << 16;
        }
        if (tid == 0) grp.w_nseg[qg] = min(__builtin_popcountll(mask), kBoundSegs);
    }
    if (!grp.w_cnt) return;  // probes only (sharded front half: the slot tables are filled after the exchange)
    {
        // every (query, probe) pair takes a slot in its list's table (one global atomic per pair; a list without rows
        // here has no records in the plan, its table is simply never read)
        const int c = tid < nprobe ? s_probe[tid] : -1;
        if (c >= 0) {
            const int sb = (int)blockIdx.y / grp.sb_batches;
            const int slot = atomicAdd(grp.w_cnt + sb * ivf_wide_plan_words(nlist) + (int64_t)c * kIvfWideCntStride, 1);  // < w_q: once per query
            // the table holds the byte offset of the query's 128 staged bytes in the scan's LDS (slot in the super-batch * 128)
            grp.w_lq[((int64_t)sb * nlist + c) * grp.w_q + slot] = (((int)blockIdx.y % grp.sb_batches) * kMaxBatch + b) * kDim;
        }
        PICK_STAMP(5);
    }
}

// (One WAVE per query instead -- 16 scores per lane, the same bound from the 64 lane minima, ballot compaction, counting
// rank -- was built and measured: 34.6 us per 8192 queries against this kernel's 29.3.  The selection is bound by the
// vector instructions it issues per query, about a thousand either way, not by the workgroups' turnaround.  Two waves per query
// (128 thread minima, 16 queries in flight per CU instead of 8): 27.2 us against 26.7 -- the same.)

// ------------------------------------------------------------------------------------------------
// IVF list scan.  One 256-thread workgroup per (query, probe) item; the four waves take
// alternating groups of 8 rows.  8 lanes share a row: every wave-instruction reads 8 rows x 128
// contiguous bytes (whole cache lines), four instructions cover the 512-byte rows, nothing is
// staged through LDS because no byte is used twice.  The 8 partial sums are folded with DPP
// (quad_perm xor 1, xor 2, row_half_mirror).  The running top-k of a wave is one sorted list with
// entry j living in lane j; inserting is a ballot + popcount + row_shr:1 shift.
// ------------------------------------------------------------------------------------------------
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
    __shared__ float s_qn;
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
    if (threadIdx.x < 8) {  // ||q||^2 in the reference's AVX2 order (cpu_baseline.cpp:95-114)
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float x = p.q[b * kDim + 8 * i + threadIdx.x];
            a = fmaf(x, x, a);
        }
        float sum = __shfl(a, 0);
#pragma unroll
        for (int u = 1; u < 8; ++u) sum = sum + __shfl(a, u);
        if (threadIdx.x == 0) s_qn = sum;
    }
    __syncthreads();
    const float qn = s_qn;

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


hipError_t launch_ivf_coarse_pick(const float* q, int B, const float* cents, const float* cnorm, int nlist, int nprobe,
                                  int metric, float* scores, int ld, int32_t* probes, const IvfGroup& grp, hipStream_t s, int n_batches) {
    if (nprobe > 256 || nlist > kIvfFastNlist || ld < ((nlist + 63) & ~63)) return hipErrorInvalidValue;
    const int tiles = (nlist + 63) / 64, nct = coarse_nct(tiles, n_batches);
    if (hipError_t e = coarse_lds_attr(); e != hipSuccess) return e;
    hipLaunchKernelGGL(ivf_coarse_mfma_kernel, dim3((tiles + nct - 1) / nct, n_batches), dim3(256), kCoarseLds, s, q, B, cents, cnorm, nlist, metric,
                       scores, ld, grp.mb, grp, 0, nct);
    if (nlist <= 1024) hipLaunchKernelGGL(ivf_pick_kernel<4>, dim3(B, n_batches), dim3(256), 0, s, scores, ld, nlist, nprobe, probes, grp);
    else if (nlist <= 2048) hipLaunchKernelGGL(ivf_pick_kernel<8>, dim3(B, n_batches), dim3(256), 0, s, scores, ld, nlist, nprobe, probes, grp);
    else hipLaunchKernelGGL(ivf_pick_kernel<16>, dim3(B, n_batches), dim3(256), 0, s, scores, ld, nlist, nprobe, probes, grp);
    return hipGetLastError();
}

// the queries of n_batches batches as bytes + terms + norms + the batches' "byte valued" verdicts (grp.w_*), nothing else
hipError_t launch_ivf_prep_queries(const float* q, int B, const float* cents, const float* cnorm, int nlist, const IvfGroup& grp,
                                   hipStream_t s, int n_batches) {
    if (!grp.w_q8 || !grp.w_qterm || !grp.w_qnorm || !grp.w_invalid) return hipErrorInvalidValue;
    if (hipError_t e = coarse_lds_attr(); e != hipSuccess) return e;
    hipLaunchKernelGGL(ivf_coarse_mfma_kernel, dim3(1, n_batches), dim3(256), kCoarseLds, s, q, B, cents, cnorm, nlist, 0, (float*)nullptr, 0, grp.mb, grp, 1, 1);
    return hipGetLastError();
}

// Sharded back half, first step: one wave per query of the launch group unpacks what the exchange delivered
// -- the query's probes (to the per-batch slab: the slow path reads them there), its bound and `slow` mark -- and takes a
// slot in the table of every probed list that is resident HERE (what ivf_pick_kernel does on an unsharded index).
// gathered: per slice s (= super-batch) a block of blk_words int32: probes [sb_q][nprobe] | tau [sb_q] | slow [sb_q],
// sb_q = sb_batches * 32 query slots (batch-padded).
__global__ __launch_bounds__(256) void ivf_fill_kernel(const int32_t* __restrict__ gathered, long long blk_words, int nprobe, int nlist,
                                                       const int32_t* __restrict__ offsets, int32_t* __restrict__ probes_out,
                                                       float* __restrict__ tau_out, int32_t* __restrict__ slow_out, IvfGroup grp, int B) {
    const int b = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), batch = blockIdx.y, lane = threadIdx.x & 63;  // a wave per query
    if (b >= B) return;
    const int sb = batch / grp.sb_batches, lb = batch % grp.sb_batches;
    const int sb_q = grp.sb_batches * kMaxBatch;
    const int lq = lb * kMaxBatch + b;  // the query's slot in its slice
    const int32_t* blk = gathered + sb * blk_words;
    int32_t* pout = mb_adv(probes_out, (long long)batch * grp.mb.probes) + b * nprobe;
    if (lane == 0) {
        tau_out[batch * kMaxBatch + b] = __builtin_bit_cast(float, blk[(long long)sb_q * nprobe + lq]);
        slow_out[batch * kMaxBatch + b] = blk[(long long)sb_q * nprobe + sb_q + lq];
    }
    for (int pp = lane; pp < nprobe; pp += 64) {
        const int c = blk[(long long)lq * nprobe + pp];
        pout[pp] = c;
        if (c < 0 || c >= nlist) continue;
        if (offsets[c + 1] <= offsets[c]) continue;  // not resident here: no records, no slot
        const int slot = atomicAdd(grp.w_cnt + sb * ivf_wide_plan_words(nlist) + (int64_t)c * kIvfWideCntStride, 1);
        grp.w_lq[((int64_t)sb * nlist + c) * grp.w_q + slot] = lq * kDim;
    }
}

hipError_t launch_ivf_fill(const int32_t* gathered, long long blk_words, int B, int nprobe, int nlist, const int32_t* offsets,
                           int32_t* probes_out, float* tau_out, int32_t* slow_out, const IvfGroup& grp, hipStream_t s, int n_batches) {
    if (!grp.w_cnt || !grp.w_lq || grp.sb_batches < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivf_fill_kernel, dim3((B + 3) / 4, n_batches), dim3(256), 0, s, gathered, blk_words, nprobe, nlist, offsets, probes_out,
                       tau_out, slow_out, grp, B);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Wide IVF pipeline (see IvfWideParams).
// ------------------------------------------------------------------------------------------------
// Bound of a query = k-th smallest distance among the first kIvfTauRows rows of each of its two nearest lists that hold rows
// (those rows are candidates, so k of them at most that far bound the k-th best of all candidates; two lists because the
// query's own neighbourhood is not always in the nearest one).  Fewer than k rows: tau = +inf and the query is marked for the
// exact slow path.  List-major: with one query per wave (rounds 2 and 3a) a query used one of the MFMA's 16 columns and read
// 64 KB of rows -- 537 MB out of the caches per 8192 queries, 69 us -- while a list's first rows are the same for every
// query that probes it.  So: the pick kernel enters each query in the tables of its two lists, and here a workgroup takes 16 ENTRIES
// of one list as the 16 columns ("unit").  The workgroup copies the list's first 256 rows (32 KB of the tiled byte copy)
// to LDS in ONE round trip, with the unit's entries and their query bytes requested beside it; wave w scores tiles w, w + 4,
// ... (two int8 MFMAs per tile score all 16 columns), keeps the k smallest of its 64 rows per column (k rounds of a
// minimum over the lane's 16 registers and the column's four lanes), and wave 0 merges the four waves' lists.  They go to
// tk[query][segment][0..k); ivf_tau_combine_kernel merges a query's two segments into its bound.
// A workgroup's life is a chain of three cache misses and about a microsecond per unit: all of a launch group's lists are
// meant to be in flight together (see ivf_bounds_plan_kernel).  (list, part): the units of a list are dealt to kBoundParts
// workgroups.
#ifndef VS_BOUND_PARTS
#define VS_BOUND_PARTS 2
#endif
constexpr int kBoundParts = VS_BOUND_PARTS;
constexpr int kBoundLds = kIvfTauRows * kDim + kIvfTauRows * 4 + 4 * 16 * 16 * 4;  // rows | row terms | the waves' lists
__device__ __forceinline__ void ivf_bounds_list_body(const IvfWideParams& p, const int c, const int part, char* const lds, const int pad_sb) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
#ifdef VS_STAMPS
#define BL_STAMP(i) do { if (p.dbg && threadIdx.x == 0 && part == 0 && c < 1024) p.dbg[(16384 + c) * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff); } while (0)
#else
#define BL_STAMP(i)
#endif
    BL_STAMP(0);
    // first round trip: the list's entry count and extent, the batches' flags
    const int n = min(p.zero[(int64_t)c * kIvfWideCntStride + 1], p.tq_cap);
    const int start = p.offsets[c];
    const int rows = min(p.offsets[c + 1] - start, kIvfTauRows);
    const int td = p.tdelta ? p.tdelta[c] : 0;
    int inv = 0;
    for (int b = lane; b < p.n_batches; b += 64) inv |= p.invalid[b];
    const int n_units = (n + 15) >> 4;
    if (part == 0 && tid < 16 * pad_sb) {
        // on the side (the plan's job otherwise, one store after the other): the scan takes a list's slot table 16 entries at
        // a time without looking at the count, the last block of every super-batch's table is filled up with the dummy slot
        const int sb = tid >> 4;
        const int nq = min(p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)c * kIvfWideCntStride], kIvfWideQ);
        const int sl = (nq & ~15) + (tid & 15);
        if (sl >= nq && (nq & 15)) p.lq[((int64_t)sb * p.nlist + c) * kIvfWideQ + sl] = kIvfWideQ * kDim;
    }
    BL_STAMP(1);
    if (part >= n_units) return;  // (most lists have a unit or two: the other parts' workgroups end here)
    const int tiles = (rows + 15) >> 4;
    // (a batch that is not byte valued sends the whole group to the fp32 rows here: the columns of a unit come from any batch)
    const bool i8 = p.vecs_t8 && p.metric == 0 && __ballot(inv != 0) == 0;
    const int32_t* tq = p.tq + (int64_t)c * p.tq_cap;
    i32x4* rows_s = reinterpret_cast<i32x4*>(lds);
    int* rt_s = reinterpret_cast<int*>(lds + kIvfTauRows * kDim);
    unsigned* cand_s = reinterpret_cast<unsigned*>(lds + kIvfTauRows * kDim + kIvfTauRows * 4);  // [wave][column][16]
    constexpr unsigned kNone = 0xffffffffu;
    constexpr int NT = kIvfTauRows / 16 / 4;  // tiles per wave at most
    constexpr int PIECES = kIvfTauRows * kDim / 16 / 256;  // 16-byte pieces of the rows per thread
    const int tstart = start + td;
    // second round trip: the first unit's entries and the rows (byte path); third: the entries' query bytes, requested
    // before the rows are waited for
    // (a workgroup's further units: their entries and query bytes are requested one unit ahead)
    int e = tq[min(part * 16 + r, n - 1)];
    int e_nx = tq[min((part + kBoundParts) * 16 + r, n - 1)];
    i32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    int qt = 0;
    if (i8) {
        i32x4 piece[PIECES];
        const i32x4* src = reinterpret_cast<const i32x4*>(p.vecs_t8 + (int64_t)tstart * kDim);
#pragma unroll
        for (int j = 0; j < PIECES; ++j) piece[j] = src[min(tid + 256 * j, tiles * 128 - 1)];
        const int rt_v = p.rterm_t[tstart + min(tid, tiles * 16 - 1)];
        const int qg = e & 0xffff;
        b0 = *reinterpret_cast<const i32x4*>(p.q8 + (int64_t)qg * kDim + 16 * g);
        b1 = *reinterpret_cast<const i32x4*>(p.q8 + (int64_t)qg * kDim + 64 + 16 * g);
        qt = p.qterm[qg];
#pragma unroll
        for (int j = 0; j < PIECES; ++j) rows_s[tid + 256 * j] = piece[j];
        if (tid < kIvfTauRows) rt_s[tid] = rt_v;
        __syncthreads();
    }
    for (int u = part; u < n_units; u += kBoundParts) {
        const int ei = u * 16 + r;
        const bool has = ei < n;
        i32x4 nb0 = {0, 0, 0, 0}, nb1 = {0, 0, 0, 0};
        int nqt = 0, e_nx2 = 0;
        if (u + kBoundParts < n_units) {  // (workgroup-uniform)
            if (i8) {
                const int qn = e_nx & 0xffff;
                nb0 = *reinterpret_cast<const i32x4*>(p.q8 + (int64_t)qn * kDim + 16 * g);
                nb1 = *reinterpret_cast<const i32x4*>(p.q8 + (int64_t)qn * kDim + 64 + 16 * g);
                nqt = p.qterm[qn];
            }
            e_nx2 = tq[min((u + 2 * kBoundParts) * 16 + r, n - 1)];
        }
        const int qg = e & 0xffff, seg = e >> 16;
        // a distance is kept as a 32-bit key that is unique in its column: value | row in the low byte.  Byte rows: the
        // distance is an integer below 2^23, the key holds it exactly.  fp32 rows: the ordered bits of the float without
        // their low byte (the bound is rounded up again when it is read back).
        unsigned v[4 * NT];
        if (i8) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = wave + 4 * i;
                if (t < tiles) {
                    const i32x4 a0 = rows_s[t * 128 + lane], a1 = rows_s[t * 128 + 64 + lane];
                    const i32x4 rt = *reinterpret_cast<const i32x4*>(rt_s + 16 * t + 4 * g);
                    i32x4 acc = {0, 0, 0, 0};
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[4 * i + j] = 16 * t + 4 * g + j < rows ? ((unsigned)(qt + rt[j] - 2 * acc[j]) << 8 | (unsigned)(16 * t + 4 * g + j)) : kNone;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * i + j] = kNone;
                }
            }
        } else {
            const float* qsrc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.q) + (long long)(qg >> 5) * p.q_batch_bytes) + (qg & 31) * kDim;
            f32x4 qf[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) qf[cc] = *reinterpret_cast<const f32x4*>(qsrc + 16 * cc + 4 * g);
            const float qn = p.qnorm[qg];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = wave + 4 * i;
                if (t < tiles) {
                    const int row = min(start + 16 * t + r, start + rows - 1);
                    f32x4 a[8];
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) a[cc] = *reinterpret_cast<const f32x4*>(p.vecs + (int64_t)row * kDim + 16 * cc + 4 * g);
                    const f32x4 bn = *reinterpret_cast<const f32x4_u*>(p.vnorm + start + 16 * t + 4 * g);  // padded by 64
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cc][j], qf[cc][j], acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = p.metric ? -acc[j] : fmaf(-2.0f, acc[j], qn + bn[j]);
                        v[4 * i + j] = 16 * t + 4 * g + j < rows ? ((f32_ordered(d) & ~0xffu) | (unsigned)(16 * t + 4 * g + j)) : kNone;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * i + j] = kNone;
                }
            }
        }
        BL_STAMP(2);
        // the wave's k smallest per column, ascending
        for (int round = 0; round < p.k; ++round) {
            unsigned m = v[0];
#pragma unroll
            for (int i = 1; i < 4 * NT; ++i) m = min(m, v[i]);
            const unsigned x = col4_min_u32(m);
#pragma unroll
            for (int i = 0; i < 4 * NT; ++i) v[i] = v[i] == x ? kNone : v[i];  // (it has one holder)
            if (g == 0) cand_s[(wave * 16 + r) * 16 + round] = x;
        }
        __syncthreads();
        if (wave == 0) {
            // four-way merge: lane (column, w) walks wave w's list
            const unsigned* mine = cand_s + (g * 16 + r) * 16;
            int idx = 0;
            unsigned head = mine[0];
            float* out = p.tk + ((int64_t)qg * kBoundSegs + seg) * 16;
            for (int round = 0; round < p.k; ++round) {
                const unsigned x = col4_min_u32(head);
                if (head == x && x != kNone) {
                    ++idx;
                    head = idx < p.k ? mine[idx] : kNone;
                }
                if (g == 0 && has) out[round] = x >= 0xff800000u ? VS_INF : i8 ? (float)(x >> 8) : f32_unordered(x | 0xffu);
            }
        }
        __syncthreads();
        e = e_nx;
        e_nx = e_nx2;
        b0 = nb0;
        b1 = nb1;
        qt = nqt;
        BL_STAMP(3);
#ifdef VS_STAMPS
        if (p.dbg && threadIdx.x == 0 && part == 0 && c < 1024) {
            p.dbg[(16384 + c) * 16 + 8] = n;
            p.dbg[(16384 + c) * 16 + 9] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);       // XCC_ID
            p.dbg[(16384 + c) * 16 + 10] = (int)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);  // HW_ID low 16 bits
        }
#endif
    }
}

// A query's bound from its segments' k smallest distances (ivf_bounds_list_body): the k-th smallest of their union, with
// a slack on fp32 rows (integer distances are exact: the bound sits right above the k-th value).  One thread per query; the launch also leaves the bound tables' counters zeroed.
__device__ __forceinline__ float ivf_tau_of(const IvfWideParams& p, const int qg) {
    const int ns = p.nseg[qg];
    const float* lists = p.tk + (int64_t)qg * kBoundSegs * 16;
    int pos[kBoundSegs];
    float head[kBoundSegs];
#pragma unroll
    for (int sgm = 0; sgm < kBoundSegs; ++sgm) {
        pos[sgm] = 0;
        head[sgm] = sgm < ns ? lists[sgm * 16] : VS_INF;
    }
    float kth = VS_INF;
    for (int t = 0; t < p.k; ++t) {  // k steps of a merge of the segments' ascending lists
        int best = 0;
#pragma unroll
        for (int sgm = 1; sgm < kBoundSegs; ++sgm)
            if (head[sgm] < head[best]) best = sgm;
        kth = head[0];
#pragma unroll
        for (int sgm = 1; sgm < kBoundSegs; ++sgm)
            if (sgm == best) kth = head[sgm];
#pragma unroll
        for (int sgm = 0; sgm < kBoundSegs; ++sgm)
            if (sgm == best) {
                ++pos[sgm];
                head[sgm] = (sgm < ns && pos[sgm] < p.k) ? lists[sgm * 16 + pos[sgm]] : VS_INF;
            }
    }
    const bool i8 = p.vecs_u8 && p.metric == 0 && p.invalid[qg >> 5] == 0;
    const float tb = i8 ? next_up(kth) : kth + 1e-4f * fabsf(kth) + 1e-30f;
    return kth < VS_INF ? tb : VS_INF;  // +inf: no bound (fewer than k rows in the query's segments)
}
// ... as a launch: the sharded front half, whose bounds travel to the other ranks.  (An unsharded group's scan works the bound
// out while it stages its queries, p.tau_inline: a launch of a thread per query was 5 us + a launch gap per group.)
__global__ __launch_bounds__(256) void ivf_tau_combine_kernel(const IvfWideParams p) {
    const int qg = blockIdx.x * 256 + threadIdx.x;
    for (int c = qg; c < p.nlist; c += (int)gridDim.x * 256) p.zero[(int64_t)c * kIvfWideCntStride + 1] = 0;
    const int batch = qg >> 5, qi = qg & 31;
    if (batch >= p.n_batches || qi >= p.B) return;
    const float t = ivf_tau_of(p, qg);
    p.tau[qg] = t;
    if (!(t < VS_INF)) p.slow[qg] = 1;
}

// Work plan of one super-batch (blockIdx.y), several workgroups each (see ivf_group_plan_kernel): records of bounded cost.
// A record is one kIvfWideUnit-row unit of a chunk whose list is probed, times one range of at most S of the slots of
// the list's query table: (first row, chunk end, list, first slot | end slot << 16).  S = 256 (a record costs between 1
// and 16 column blocks beside its rows; every further record of a unit reads the unit's rows again) unless the plan would
// not fit `units_cap`, then the next power of two that does (1024 = no split always fits).
constexpr int kIvfWideTiles = kIvfWideUnit / 16;  // 16-row MFMA tiles per unit
constexpr int kIvfWideSplits = 3;  // S = 256 << i
constexpr int kPlanThreads = 256, kPlanWaves = kPlanThreads / 64, kPlanClasses = 16;
constexpr int kPlanLds = (kIvfFastNlist + 2 + kPlanWaves * kIvfWideSplits + 3 * (kPlanClasses + 1)) * 4;
__device__ __forceinline__ void ivf_plan_body(const IvfWideParams& p, const int sb, const int slice, const int nsl, char* const lds,
                                              const bool pad_tables = true) {
    int* const cnt_s = reinterpret_cast<int*>(lds);  // [kIvfFastNlist]
    int& s_carry = cnt_s[kIvfFastNlist];
    int& s_shift = cnt_s[kIvfFastNlist + 1];
    int (*s_tot)[kIvfWideSplits] = reinterpret_cast<int (*)[kIvfWideSplits]>(cnt_s + kIvfFastNlist + 2);  // [kPlanWaves]
    int* const s_ctot = cnt_s + kIvfFastNlist + 2 + kPlanWaves * kIvfWideSplits;  // [kPlanClasses + 1] each
    int* const s_cpre = s_ctot + kPlanClasses + 1;
    int* const s_cpos = s_cpre + kPlanClasses + 1;
    const int tid = threadIdx.x;
#ifdef VS_STAMPS
#define PLAN_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[(20480 + sb * 16 + slice) * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff); } while (0)
#else
#define PLAN_STAMP(i)
#endif
    PLAN_STAMP(0);
    int32_t* units = p.units + (int64_t)sb * p.units_sb_stride;
    const int pl = tid & 63, wv = tid >> 6;
    const int c0 = (int)((long long)p.n_chunks * slice / nsl), c1 = (int)((long long)p.n_chunks * (slice + 1) / nsl);
    // Everything read from global memory is requested before the first barrier (one cache round trip, not one per phase):
    // the pair counters, the chunk table entries of the all-chunks pass (eight per thread in registers, more only for
    // very large indexes) and this thread's chunk of the workgroup's own slice.
    // (every load is unconditional, at a clamped index: a load under a per-lane condition is waited for on the spot, and the
    // dozen of them here were a dozen round trips, one after the other -- 9 of the body's 21 us)
    constexpr int EARLY = 8;
    int e_list[EARLY], e_rows[EARLY];
    const int last_chunk = max(p.n_chunks - 1, 0);  // (n_chunks > 0: the wide pipeline is not used on an index without rows)
#pragma unroll
    for (int i = 0; i < EARLY; ++i) {
        const int chunk = min(tid + kPlanThreads * i, last_chunk);
        e_list[i] = p.chunk_list[chunk];
        e_rows[i] = p.chunk_rows[chunk];
    }
    const int own = min(c0 + tid, last_chunk);
    const int o_list = p.chunk_list[own], o_rows = p.chunk_rows[own], o_row0 = p.chunk_trow0[own];
    long long cand = 0;
    {
        constexpr int CPT = kIvfFastNlist / kPlanThreads;  // counters per thread: loaded together, then stored
        int n[CPT], len[CPT];
        const bool want_len = p.cand_count && slice == 0;  // (workgroup-uniform)
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            if (kPlanThreads * i >= p.nlist) {  // (uniform)
                n[i] = len[i] = 0;
                continue;
            }
            const int c = min(tid + kPlanThreads * i, p.nlist - 1);
            n[i] = p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)c * kIvfWideCntStride];
            len[i] = want_len ? p.offsets[c + 1] - p.offsets[c] : 0;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + kPlanThreads * i;
            if (c < p.nlist) {
                const int nq = min(n[i], kIvfWideQ);
                cnt_s[c] = nq;
                // the scan takes a list's slot table 16 entries at a time without looking at the count: the last block
                // is filled up with the dummy slot (one workgroup does it; the entries are stale otherwise)
                // (the list-major bounds' workgroups do it instead when they run in the same launch)
                if (slice == 0 && pad_tables)
                    for (int sl = nq; sl < ((nq + 15) & ~15); ++sl) p.lq[((int64_t)sb * p.nlist + c) * kIvfWideQ + sl] = kIvfWideQ * kDim;
                cand += (long long)nq * len[i];
            }
        }
    }
    if (p.cand_count && slice == 0) {  // the candidate statistic (IVFIndex.cpp: total_candidates), one atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand += __shfl_xor(cand, o);
        if ((tid & 63) == 0 && cand) atomicAdd(p.cand_count, (unsigned long long)cand);
    }
    if (tid == 0) s_carry = 0;
    __syncthreads();
    PLAN_STAMP(1);
    auto units_of = [&](int rows) { return (rows + kIvfWideUnit - 1) / kIvfWideUnit; };
    {   // one pass over all chunks: for every split size the plan's record count -- does it fit? (every workgroup works
        // this out for itself: the same numbers, the same answer)
        int tot[kIvfWideSplits];
#pragma unroll
        for (int i = 0; i < kIvfWideSplits; ++i) tot[i] = 0;
        auto add = [&](int list, int rows) {
            const int nq = cnt_s[list];
            const int nu = units_of(rows);
#pragma unroll
            for (int i = 0; i < kIvfWideSplits; ++i) tot[i] += nu * ((nq + (256 << i) - 1) >> (8 + i));
        };
#pragma unroll
        for (int i = 0; i < EARLY; ++i)
            if (tid + kPlanThreads * i < p.n_chunks) add(e_list[i], e_rows[i]);
        for (int chunk = tid + kPlanThreads * EARLY; chunk < p.n_chunks; chunk += kPlanThreads) add(p.chunk_list[chunk], p.chunk_rows[chunk]);
#pragma unroll
        for (int i = 0; i < kIvfWideSplits; ++i) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tot[i] += __shfl_xor(tot[i], o);
            if (pl == 0) s_tot[wv][i] = tot[i];
        }
        if (tid <= kPlanClasses) s_ctot[tid] = s_cpre[tid] = 0;
        __syncthreads();
        if (tid == 0) {
            int sh = kIvfWideSplits - 1;
            for (int i = kIvfWideSplits - 1; i >= 0; --i) {
                long long t = 0;
                for (int w = 0; w < kPlanWaves; ++w) t += s_tot[w][i];
                if (t <= p.units_cap) sh = i;
            }
            s_shift = 8 + sh;
        }
        __syncthreads();
    }
    const int shift = s_shift;
    PLAN_STAMP(2);
    // The records are laid out by cost class, the most expensive class first (class = column blocks of a record = queries
    // of its slot range / 16, capped): the scan deals records round-robin, so every wave gets one record of every
    // stratum and the sums come out alike (dealt in list order the slowest of 4096 waves took 25 % longer than the
    // average).  Second pass over all chunks: records per class in all chunks and in the chunks before this slice.
    auto class_of = [&](int nq) { return min((min(nq, 1 << shift) + 15) >> 4, kPlanClasses); };
    {
        auto add = [&](int chunk, int list, int rows) {
            const int nq = cnt_s[list];
            if (nq == 0) return;
            const int n = units_of(rows) * ((nq + (1 << shift) - 1) >> shift);
            const int cl = class_of(nq);
            atomicAdd(&s_ctot[cl], n);
            if (chunk < c0) atomicAdd(&s_cpre[cl], n);
        };
#pragma unroll
        for (int i = 0; i < EARLY; ++i)
            if (tid + kPlanThreads * i < p.n_chunks) add(tid + kPlanThreads * i, e_list[i], e_rows[i]);
        for (int chunk = tid + kPlanThreads * EARLY; chunk < p.n_chunks; chunk += kPlanThreads) add(chunk, p.chunk_list[chunk], p.chunk_rows[chunk]);
        __syncthreads();
        if (tid == 0) {
            int base = 0;
            for (int cl = kPlanClasses; cl >= 1; --cl) {
                s_cpos[cl] = base + s_cpre[cl];  // where this slice's records of the class start
                base += s_ctot[cl];
            }
            s_carry = base;  // records in the plan
        }
        __syncthreads();
    }
    PLAN_STAMP(4);
    for (int base = c0; base < c1; base += kPlanThreads) {
        const int chunk = base + tid;
        if (chunk >= c1) break;
        const bool first = base == c0;
        const int c = first ? o_list : p.chunk_list[chunk];
        const int rows = first ? o_rows : p.chunk_rows[chunk];
        const int nq = cnt_s[c];
        if (nq == 0) continue;
        const int nu = units_of(rows);
        const int nr = nu * ((nq + (1 << shift) - 1) >> shift);
        int pos = atomicAdd(&s_cpos[class_of(nq)], nr);  // (the order inside a class is whatever the threads make it)
        const int r0 = first ? o_row0 : p.chunk_trow0[chunk];  // padded rows
        const int r_end = r0 + rows;
        // the records of one unit are neighbours: the waves that take them read the same rows at about the same time
        for (int i = 0; i < nu; ++i)
            for (int q0 = 0; q0 < nq; q0 += 1 << shift)
                reinterpret_cast<int4*>(units)[pos++] = make_int4(r0 + kIvfWideUnit * i, r_end, c, q0 | (min(nq, q0 + (1 << shift)) << 16));
    }
    PLAN_STAMP(3);
    if (tid == 0 && slice == nsl - 1) p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)p.nlist * kIvfWideCntStride] = s_carry;
}

// Bounds and plan in ONE launch (both need the pick kernel's output only: side by side instead of one after the other).  The
// first n_plan * n_sb workgroups plan, the rest take (list, part) pairs of the bounds.  Four workgroups per CU: the bounds'
// workgroups are one short chain of cache misses each, all of a launch group's lists should be in flight together.
__global__ __launch_bounds__(256, 4) void ivf_bounds_plan_kernel(const IvfWideParams p, const int n_plan, const int n_sb, const bool pad_here) {
    __shared__ __attribute__((aligned(16))) char lds[kBoundLds > kPlanLds ? kBoundLds : kPlanLds];
    const int wg = blockIdx.x;
    if (wg < n_plan * n_sb) ivf_plan_body(p, wg / n_plan, wg % n_plan, n_plan, lds, pad_here);
    else ivf_bounds_list_body(p, (wg - n_plan * n_sb) % p.nlist, (wg - n_plan * n_sb) / p.nlist, lds, pad_here ? 0 : n_sb);
}

// The list-major scan of one super-batch (blockIdx.y).  A workgroup stages the super-batch's queries once (as bytes: 128
// KB) with their constant terms and thresholds; after that every wave works alone on records of the plan: the unit's two
// 16-row tiles are the MFMA A operands, the queries of the record's slot range come 16 at a time as B operands (gathered
// from the staged bytes through the list's slot table), and a distance under its query's bound goes to the wave's
// candidate buffer (plain stores, positions from a ballot).  Rows that are not bytes, or a super-batch with a non-byte
// query: the same on the fp32 rows with queries gathered from global memory.
//
// Cost model of the int8 path (measured with -DVS_STAMPS: the loop took the same time on cache-hot rows): a wave64 VALU
// instruction occupies its SIMD for 4 cycles and four waves share the SIMD, a column block is 4 MFMAs (64 cycles), so the
// instructions around the MFMAs are what the kernel costs.  Hence:
//  - d < ti  <=>  2 dot - rt > th (th = qt - ti); with rt = 2 rh + ro (ro = 0 or 1) and acc = dot - rh that is
//    2 acc - ro > th, and for EVEN th simply acc > th / 2.  The bound is an upper bound of the k-th distance and only
//    filters, so ti is raised by one where th would be odd: the test per value is one comparison with th >> 1, -rh
//    enters as the MFMA's C operand straight from an array that holds it (`nrh`), and the hot path is the maximum of
//    the 8 results against th >> 1.
//  - the distance itself (qt + ro - 2 acc) is completed when the wave bins its candidates at the end;
//  - rows past the chunk end are poisoned in the C operand (only a chunk's last unit pays); a list's slot table is
//    padded to a multiple of 16 entries with a dummy query slot whose bound admits nothing (the plan does it), so a
//    column block never looks at the list's count;
//  - the slot table holds LDS byte offsets (slot * 128), a slot's bytes are kept as four 32-byte units [MFMA 1 | MFMA 2]
//    per lane group, swizzled by slot, and the per-query words sit in front of the query bytes: the B operands of a
//    column block cost 3 vector instructions of address arithmetic and their bound 2 (10 in all beside the 4 MFMAs);
//  - a record's fields are scalars, its rows are read unclamped (the arrays are padded) at scalar base + one lane offset,
//    the first four column blocks are straight-line code on slots fetched with the rows, and two register sets
//    alternate instead of being copied.
#ifndef VS_WIDE_WAVES
#define VS_WIDE_WAVES 16
#endif
#ifndef VS_WIDE_SETS
#define VS_WIDE_SETS 2
#endif
// One workgroup per CU (its LDS holds the group's queries): 16 waves, two register sets each (a record being scored,
// the next record's rows in flight).  Three sets (two records in flight) fit the 128 registers too and were measured:
// + 1.7 % on one stream, nothing on two.  The loop is bound by instruction issue, not by what a wave has in flight.
constexpr int kIvfWideThreads = 64 * VS_WIDE_WAVES;
constexpr int kIvfWideSets = VS_WIDE_SETS;
constexpr int kIvfWideWaves = kIvfWideThreads / 64;
constexpr int kIvfWideSlots = kIvfWideQ + 1;  // + the dummy slot
constexpr int kIvfWideLds = kIvfWideSlots * kDim + 4 * kIvfWideSlots * 4;  // query bytes + four per-query words
constexpr int kIvfWideDeadThr = 0x3fffffff;
__global__ __launch_bounds__(kIvfWideThreads) void ivf_scan_wide_kernel(const IvfWideParams p) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    extern __shared__ __attribute__((aligned(16))) char wide_smem[];
    // (the per-query words first: their LDS addresses then fit the 16-bit offset field of the read instructions)
    int* thh_s = reinterpret_cast<int*>(wide_smem);                         // [slot] int8 path: acc > thh  <=>  d < ti (see above)
    int* qt_s = thh_s + kIvfWideSlots;
    float* tau_s = reinterpret_cast<float*>(qt_s + kIvfWideSlots);
    float* qn_s = tau_s + kIvfWideSlots;
    constexpr int kQ8Off = 4 * kIvfWideSlots * 4;                           // 16400: 16-byte aligned
    int* q8_s = reinterpret_cast<int*>(wide_smem + kQ8Off);                 // [slot][128 bytes]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int sb = blockIdx.y;
    const int b0 = sb * p.sb_batches, b1 = min(p.n_batches, b0 + p.sb_batches);
    const int qbase = b0 * kMaxBatch;
    const int nslots = (b1 - b0) * kMaxBatch;
    const int wb = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * kIvfWideWaves + wave;  // this wave's candidate buffer
    const int nw = (int)gridDim.x * kIvfWideWaves;
    // records are dealt round-robin over the workgroups first: what a workgroup's 16 waves hold at any moment comes from
    // 16 places of the plan (a popular list's records are expensive and sit together)
    int u = wave * (int)gridDim.x + (int)blockIdx.x;
    const int4* recs = reinterpret_cast<const int4*>(p.units + (int64_t)sb * p.units_sb_stride);
    const int32_t* lq = p.lq + (int64_t)sb * p.nlist * kIvfWideQ;
    int4* wbuf = p.sink.wbuf + (int64_t)wb * p.sink.wcap;
    int wbase = 0;
    VS_STAMP(0);
    // everything the staging needs is requested in one go (a kernel start is a chain of cold round trips otherwise)
    const int n_units = p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)p.nlist * kIvfWideCntStride];
    constexpr int NS = kIvfWideSets, DEPTH = NS - 1;  // register sets; records whose rows are in flight beside the current one
    int4 rv[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) rv[i] = recs[min(u + i * nw, p.units_cap - 1)];
    int inv = 0;
    for (int b = b0; b < b1; ++b) inv |= p.invalid[b];
    constexpr int PER = (kIvfWideQ * 8 + kIvfWideThreads - 1) / kIvfWideThreads;
    int4 v[PER];
    {
        const int4* src = reinterpret_cast<const int4*>(p.q8 + (int64_t)qbase * kDim);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + j * kIvfWideThreads;
            v[j] = i < nslots * 8 ? src[i] : make_int4(0, 0, 0, 0);
        }
    }
    for (int s = tid; s < kIvfWideSlots; s += kIvfWideThreads) {
        const int qg = qbase + min(s, nslots - 1);
        bool live = s < nslots && (s & 31) < p.B;
        float t0 = -VS_INF;
        if (p.tau_inline) {  // (uniform) the bound from the bounds launch's segment lists, see ivf_tau_of
            if (live) {
                t0 = ivf_tau_of(p, qg);
                if (!(t0 < VS_INF)) {  // no bound: the slow path's (one workgroup of the super-batch says so to the ranking)
                    live = false;
                    t0 = -VS_INF;
                    if (blockIdx.x == 0) p.slow[qg] = 1;
                }
            }
        } else {
            live = live && p.slow[qg] == 0;  // a query without a bound goes through the slow path only
            if (live) t0 = p.tau[qg];
        }
        tau_s[s] = t0;
        qn_s[s] = p.qnorm[qg];
        const int qt = p.qterm[qg];
        qt_s[s] = qt;
        // d < tau for integer d  <=>  d < ceil(tau)  (distances are below 2^24: any bound from 2^26 on admits everything)
        const int ti = (int)ceilf(fminf(fmaxf(t0, -67108864.f), 67108864.f));
        thh_s[s] = live ? (qt - ti) >> 1 : kIvfWideDeadThr;
    }
    // A slot's 128 bytes are kept as four 32-byte units, unit g = [bytes 16g.. | bytes 64+16g..] = what lane group g feeds
    // the two MFMAs of a column block (one address, two reads), and unit g sits at position g ^ (slot & 3): the B-operand
    // gather reads the same unit of 16 arbitrary slots at once, which unswizzled is a 16-way bank conflict.
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + j * kIvfWideThreads;  // 16-byte segment i & 7 of slot i >> 3
        if (i < kIvfWideQ * 8) reinterpret_cast<int4*>(q8_s)[(i & ~7) + ((((i & 3) ^ ((i >> 3) & 3))) << 1) + ((i >> 2) & 1)] = v[j];
    }
    if (tid < 8) reinterpret_cast<int4*>(q8_s)[kIvfWideQ * 8 + tid] = make_int4(0, 0, 0, 0);
    const bool i8 = p.vecs_t8 && p.metric == 0 && inv == 0;
#pragma unroll
    for (int i = 0; i < NS; ++i)
        if (u + i * nw >= n_units) rv[i] = make_int4(0, 0, 0, 0);
    if ((int)blockIdx.x >= n_units) return;  // workgroup-uniform: not even wave 0 has a record (nothing to bin either)

    struct Rec {
        int r0, r_end, c, q0, nq;
    };
    auto unpack = [&](const int4& rv) __attribute__((always_inline)) {
        Rec rc;
        rc.r0 = __builtin_amdgcn_readfirstlane(rv.x);
        rc.r_end = __builtin_amdgcn_readfirstlane(rv.y);
        rc.c = __builtin_amdgcn_readfirstlane(rv.z);
        const int w = __builtin_amdgcn_readfirstlane(rv.w);
        rc.q0 = w & 0xffff;
        rc.nq = (w >> 16) - rc.q0;
#ifdef VS_STAMPS
        if (p.diag & 1) rc.nq = 0;
#endif
        return rc;
    };
    constexpr int PF = 4;  // column blocks whose query slots are fetched together with the unit's rows
    constexpr int NT = kIvfWideTiles;
    const unsigned loff = (unsigned)(16 * lane);  // the lane's bytes inside one half of a tile: a load is 1 KB in one piece
    const unsigned g32 = 32u * (unsigned)g;
    auto issue = [&](const Rec& rc, i32x4 (&a0)[NT], i32x4 (&a1)[NT], i32x4 (&nr)[NT], int (&qlp)[PF]) __attribute__((always_inline)) {
        const int8_t* rows = p.vecs_t8 + (int64_t)rc.r0 * kDim;  // r0 is a multiple of 32 (rows past the chunk end: poisoned below)
        const int32_t* nrp = p.nrh_t + rc.r0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            a0[t] = *reinterpret_cast<const i32x4*>(rows + loff + t * 16 * kDim);
            a1[t] = *reinterpret_cast<const i32x4*>(rows + loff + t * 16 * kDim + 1024);
            nr[t] = *reinterpret_cast<const i32x4*>(nrp + 4 * g + 16 * t);
        }
        const int32_t* lqn = lq + (int64_t)rc.c * kIvfWideQ + rc.q0;
        // (raw table entries, whatever the record's slot range: a row of the table has room for them, and selecting here
        // would make the compiler wait for the loads right away)
#pragma unroll
        for (int i = 0; i < PF; ++i) qlp[i] = lqn[16 * i + r];
    };
    if (i8) {
        i32x4 SA0[NS][NT], SA1[NS][NT], SN[NS][NT];
        int SQ[NS][PF];
        Rec SR[NS];
        // the first records' rows travel while the queries are stored
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            SR[i] = unpack(rv[i]);
            issue(SR[i], SA0[i], SA1[i], SN[i], SQ[i]);
        }
        __syncthreads();
        VS_STAMP(1);
        int4 rvn = rv[DEPTH];
        auto compute = [&](const Rec& rc, const i32x4 (&a0)[NT], const i32x4 (&a1)[NT], i32x4 (&nr)[NT], const int (&qlc)[PF]) __attribute__((always_inline)) {
            if (rc.r0 + kIvfWideUnit > rc.r_end) {  // wave-uniform, a chunk's last unit: rows past its end can never pass
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (rc.r0 + 16 * t + 4 * g + j >= rc.r_end) nr[t][j] = -(1 << 28);
            }
            // (e: byte offset of the slot's staged bytes, as the slot table holds it; 10 vector instructions beside the MFMAs)
            auto block = [&](const unsigned e) __attribute__((always_inline)) {
                const unsigned a = ((g32 ^ ((e >> 2) & 0x60u)) + e);  // unit g ^ (slot & 3) of the slot
                const i32x4 bq0 = *reinterpret_cast<const i32x4*>(wide_smem + kQ8Off + a);
                const i32x4 bq1 = *reinterpret_cast<const i32x4*>(wide_smem + kQ8Off + a + 16);
                const int thh = *reinterpret_cast<const int*>(wide_smem + (e >> 5));
                i32x4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[t], bq0, nr[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[t], bq1, acc[t], 0, 0, 0);
                }
                int emax = max(max(acc[0][0], acc[0][1]), max(acc[0][2], acc[0][3]));
#pragma unroll
                for (int t = 1; t < NT; ++t) emax = max(max(emax, acc[t][0]), max(max(acc[t][1], acc[t][2]), acc[t][3]));
#ifdef VS_STAMPS
                if ((p.diag & 8) && emax != 0x12345678) return;
#endif
                if (__ballot(emax > thh)) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool pass = acc[t][j] > thh;
                            const unsigned long long mask = __ballot(pass);
                            if (mask) {
                                const int pos = wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                                // (slot, acc, row of the lane group's first value, lane group): sink_bin_wave's hook below
                                // completes row and distance (no per-value lane constants here: they would be spilled)
                                if (pass && pos < p.sink.wcap) wbuf[pos] = make_int4((int)e, acc[t][j], rc.r0 + 16 * t + j, g);
                                wbase += __popcll(mask);
                            }
                        }
                }
            };
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (16 * i < rc.nq) block((unsigned)qlc[i]);  // wave-uniform (the table's last block is padded with the dummy slot)
            if (rc.nq > 16 * PF) {  // a list probed by more than 64 of the group's queries
                const int32_t* lqc = lq + (int64_t)rc.c * kIvfWideQ + rc.q0;
                for (int cb = 16 * PF; cb < rc.nq; cb += 16) block((unsigned)lqc[cb + r]);
            }
        };
        auto next_record = [&](int idx) __attribute__((always_inline)) {
#ifdef VS_STAMPS
            if (p.diag & 2) idx = wave * (int)gridDim.x + (int)blockIdx.x;
#endif
            return idx < n_units ? recs[idx] : make_int4(0, 0, 0, 0);  // (a zero record reads rows 0.. and no slots: harmless)
        };
        // software pipeline over the wave's records: set ph holds the current record, the other sets the next DEPTH ones
        // (rows requested DEPTH steps ahead); the set just scored is refilled
        for (bool more = true; more;) {
#pragma unroll
            for (int ph = 0; ph < NS; ++ph) {
                if (!more) break;
                constexpr int dummy = 0;
                (void)dummy;
                const int nx = (ph + DEPTH) % NS;
                SR[nx] = unpack(rvn);  // record u + DEPTH * nw
                rvn = next_record(u + (DEPTH + 1) * nw);
                issue(SR[nx], SA0[nx], SA1[nx], SN[nx], SQ[nx]);
                compute(SR[ph], SA0[ph], SA1[ph], SN[ph], SQ[ph]);
                u += nw;
                more = u < n_units;
            }
        }
        VS_STAMP(2);
#ifdef VS_STAMPS
        if (p.diag & 4) return;
#endif
        // the wave's candidates go to the per-query lists here (no binning launch); an entry's distance is qt + ro - 2 acc
        sink_bin_wave(p.sink, wb, wbase, lane, [&](const int4& c) {
            const int row = c.z + 4 * c.w, slot = c.x >> 7;
            const int d = qt_s[slot] + (p.rterm_t[row] & 1) - 2 * c.y;
            return make_int4(qbase + slot, __builtin_bit_cast(int, (float)d), row, 0);
        }
#ifdef VS_STAMPS
        , p.diag
#endif
        );
        VS_STAMP(3);
        return;
    }
    __syncthreads();
    int4 rec = rv[0];
    for (; u < n_units; u += nw) {
        const Rec rc = unpack(rec);
        const int td = p.tdelta ? p.tdelta[rc.c] : 0;  // the record is in padded rows, so are the candidates
        const int r0 = rc.r0 - td, r_end = rc.r_end - td, nq = rc.nq;
        if (u + nw < n_units) rec = recs[u + nw];  // the next record, in flight during this one
        const int32_t* lqc = lq + (int64_t)rc.c * kIvfWideQ + rc.q0;
        // the unit's rows stay in registers (both tiles) while its column blocks pass: a block's query fragments -- eight loads
        // of 64 separate 16-byte pieces each, the address unit's time -- are then fetched once per unit, not once per tile
        f32x4 a[kIvfWideTiles][8], bn[kIvfWideTiles];
#pragma unroll
        for (int t = 0; t < kIvfWideTiles; ++t) {
            const int row = min(r0 + 16 * t + r, r_end - 1);
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) a[t][c8] = *reinterpret_cast<const f32x4*>(p.vecs + (int64_t)row * kDim + 16 * c8 + 4 * g);
            bn[t] = *reinterpret_cast<const f32x4_u*>(p.vnorm + min(r0 + 16 * t, r_end - 1) + 4 * g);  // (padded by 64)
        }
        for (int cb = 0; cb < nq; cb += 16) {
            const int sq = cb + r;
            const bool live = sq < nq;
            const int ql = live ? lqc[sq] >> 7 : 0;  // (the table holds slot * 128)
            const int qg = qbase + ql;
            const float* qsrc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.q) + (long long)(qg >> 5) * p.q_batch_bytes) + (qg & 31) * kDim;
            f32x4 acc[kIvfWideTiles];
#pragma unroll
            for (int t = 0; t < kIvfWideTiles; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const f32x4 qf = *reinterpret_cast<const f32x4*>(qsrc + 16 * c8 + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < kIvfWideTiles; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][c8][i], qf[i], acc[t], 0, 0, 0);
            }
            const float qn = qn_s[ql];
            const float tq = live ? tau_s[ql] : -VS_INF;
#pragma unroll
            for (int t = 0; t < kIvfWideTiles; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = p.metric ? -acc[t][j] : fmaf(-2.0f, acc[t][j], qn + bn[t][j]);
                    const int rowj = r0 + 16 * t + 4 * g + j;
                    const bool pass = d < tq && rowj < r_end;
                    const unsigned long long mask = __ballot(pass);
                    if (mask) {
                        const int pos = wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                        if (pass && pos < p.sink.wcap) wbuf[pos] = make_int4(qg, __builtin_bit_cast(int, d), rowj + td, 0);
                        wbase += __popcll(mask);
                    }
                }
        }
    }
    sink_bin_wave(p.sink, wb, wbase, lane);
}

// The list-major scan on the fp32 rows as a kernel of its own -- what runs when the host knows that the group is scored in
// fp32 (vs_set_precision(h, 1), rows that are not bytes, the inner-product metric); ivf_scan_wide_kernel's fp32 branch stays
// for a group that turns out to hold a query that is not byte valued.  Same records, same candidates.  8 waves per workgroup
// instead of 16: nothing is staged but the bounds and norms, and 256 registers per wave hold what the loop needs to keep the
// MFMA pipe fed -- the unit's two 16-row tiles of the CURRENT and the NEXT record (A fragments) and a column block's query
// fragments (B), fetched once per unit.  In the shared kernel a column block waited for its slot-table entry, then
// for its eight fragment loads, then ran 32 MFMAs, once per tile: 2.9 ms per 8192 queries where the arithmetic takes 1.1.
constexpr int kIvfWideF32Threads = 512;
constexpr int kIvfWideF32Lds = (kIvfWideF32Threads / 64) * 2 * 8192;  // per wave: two landing buffers of a column block's fragments
__global__ __launch_bounds__(kIvfWideF32Threads) void ivf_scan_wide_f32_kernel(const IvfWideParams p) {
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    extern __shared__ __attribute__((aligned(16))) char f32_smem[];
    __shared__ float tau_s[kIvfWideSlots];
    __shared__ float qn_s[kIvfWideSlots];
    constexpr int WAVES = kIvfWideF32Threads / 64, NT = kIvfWideTiles, PF = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int sb = blockIdx.y;
    const int b0 = sb * p.sb_batches, b1 = min(p.n_batches, b0 + p.sb_batches);
    const int qbase = b0 * kMaxBatch;
    const int nslots = (b1 - b0) * kMaxBatch;
    const int wb = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * WAVES + wave;  // this wave's candidate buffer
    const int nw = (int)gridDim.x * WAVES;
    int u = wave * (int)gridDim.x + (int)blockIdx.x;
    const int4* recs = reinterpret_cast<const int4*>(p.units + (int64_t)sb * p.units_sb_stride);
    const int32_t* lq = p.lq + (int64_t)sb * p.nlist * kIvfWideQ;
    int4* wbuf = p.sink.wbuf + (int64_t)wb * p.sink.wcap;
    int wbase = 0;
    const int n_units = p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)p.nlist * kIvfWideCntStride];
    auto record = [&](const int idx) __attribute__((always_inline)) {  // (a zero record: rows 0.., no column block)
        return idx < n_units ? recs[min(idx, p.units_cap - 1)] : make_int4(0, 0, 0, 0);
    };
    int4 rv0 = recs[min(u, p.units_cap - 1)], rv1 = recs[min(u + nw, p.units_cap - 1)], rv2 = recs[min(u + 2 * nw, p.units_cap - 1)];
    for (int s = tid; s < kIvfWideSlots; s += kIvfWideF32Threads) {
        const int qg = qbase + min(s, nslots - 1);
        bool live = s < nslots && (s & 31) < p.B;
        float t0 = -VS_INF;
        if (p.tau_inline) {  // (uniform) see ivf_scan_wide_kernel
            if (live) {
                t0 = ivf_tau_of(p, qg);
                if (!(t0 < VS_INF)) {
                    t0 = -VS_INF;
                    if (blockIdx.x == 0) p.slow[qg] = 1;
                }
            }
        } else {
            live = live && p.slow[qg] == 0;  // a query without a bound goes through the slow path only
            if (live) t0 = p.tau[qg];
        }
        tau_s[s] = t0;
        qn_s[s] = p.qnorm[qg];
    }
    if (u >= n_units) rv0 = make_int4(0, 0, 0, 0);
    if (u + nw >= n_units) rv1 = make_int4(0, 0, 0, 0);
    if (u + 2 * nw >= n_units) rv2 = make_int4(0, 0, 0, 0);
    __syncthreads();
    if ((int)blockIdx.x >= n_units) return;  // workgroup-uniform: not even wave 0 has a record
    struct Rec {
        int r0, r_end, c, q0, nq, td;
    };
    auto unpack = [&](const int4& rv) __attribute__((always_inline)) {
        Rec rc;
        rc.c = __builtin_amdgcn_readfirstlane(rv.z);
        rc.td = p.tdelta ? p.tdelta[rc.c] : 0;  // (records are in padded rows when the index keeps the tiled byte copy)
        rc.r0 = __builtin_amdgcn_readfirstlane(rv.x) - rc.td;
        rc.r_end = __builtin_amdgcn_readfirstlane(rv.y) - rc.td;
        const int w = __builtin_amdgcn_readfirstlane(rv.w);
        rc.q0 = w & 0xffff;
        rc.nq = (w >> 16) - rc.q0;
        return rc;
    };
    auto issue_rows = [&](const Rec& rc, f32x4 (&a)[NT][8], f32x4 (&bn)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = max(min(rc.r0 + 16 * t + r, rc.r_end - 1), 0);
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) a[t][c8] = *reinterpret_cast<const f32x4*>(p.vecs + (int64_t)row * kDim + 16 * c8 + 4 * g);
            bn[t] = *reinterpret_cast<const f32x4_u*>(p.vnorm + max(min(rc.r0 + 16 * t, rc.r_end - 1), 0) + 4 * g);  // (padded by 64)
        }
    };
    auto issue_slots = [&](const Rec& rc, int (&sq)[PF]) __attribute__((always_inline)) {
        const int32_t* lqn = lq + (int64_t)rc.c * kIvfWideQ + rc.q0;
#pragma unroll
        for (int i = 0; i < PF; ++i) sq[i] = lqn[16 * i + r];  // (raw entries: a row of the table has room, selected at use)
    };
    auto slot_of = [&](const Rec& rc, const int (&sq)[PF], const int cb) __attribute__((always_inline)) {  // (wave-uniform cb)
        int e = 0;
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (cb == 16 * i) e = sq[i];
        if (cb >= 16 * PF) e = lq[(int64_t)rc.c * kIvfWideQ + rc.q0 + cb + r];
        return cb + r < rc.nq ? e >> 7 : 0;  // (the table holds slot * 128)
    };
    // A column block's query fragments land in the wave's LDS buffer `par` by LDS-DMA (no registers while they travel):
    // instruction c8 moves the 16 bytes Q[slot][16 c8 + 4 g ..] of every lane to buffer + 1024 c8 + 16 lane, which is where the
    // MFMA B operand of step c8 is read from.  The waits for these loads are written by hand (the compiler does not see them).
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(f32_smem + wave * 16384));
    const char* qbytes = reinterpret_cast<const char*>(p.q);
    auto dma = [&](const int ql, const int par) __attribute__((always_inline)) {
        const int qg = qbase + ql;
        const unsigned voff = (unsigned)((long long)(qg >> 5) * p.q_batch_bytes) + (unsigned)((qg & 31) * (kDim * 4) + 16 * g);
        const unsigned dst = lds_w + (unsigned)par * 8192u;
        // (no instruction offset: it would move the LDS address as well as the global one)
#define VS_QDMA(c8) asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff + 64u * (c8)), "s"(qbytes), "n"((c8) * 1024) : "memory", "scc")
        VS_QDMA(0); VS_QDMA(1); VS_QDMA(2); VS_QDMA(3); VS_QDMA(4); VS_QDMA(5); VS_QDMA(6); VS_QDMA(7);
#undef VS_QDMA
    };
    auto score = [&](const Rec& rc, const f32x4 (&a)[NT][8], const f32x4 (&bn)[NT], const int par, const int ql, const bool live) __attribute__((always_inline)) {
        const f32x4* qb = reinterpret_cast<const f32x4*>(f32_smem + wave * 16384 + par * 8192) + lane;
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const f32x4 qf = qb[c8 * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][c8][i], qf[i], acc[t], 0, 0, 0);
        }
        const float qn = qn_s[ql];
        const float tq = live ? tau_s[ql] : -VS_INF;
        const int qg = qbase + ql;
        // (the block's eight distances per lane, their minimum against the bound first: one ballot per block, not per value --
        // an ordinary vector instruction costs the MFMA pipe about eight cycles, and nearly every block has no hit)
        float d[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[t][j] = p.metric ? -acc[t][j] : fmaf(-2.0f, acc[t][j], qn + bn[t][j]);
        float dmin = d[0][0];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) dmin = fminf(dmin, d[t][j]);
        if (__ballot(dmin < tq) == 0) return;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rowj = rc.r0 + 16 * t + 4 * g + j;
                const bool pass = d[t][j] < tq && rowj < rc.r_end;  // (rows past the chunk's end are looked at here only)
                const unsigned long long mask = __ballot(pass);
                if (mask) {
                    const int pos = wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    if (pass && pos < p.sink.wcap) wbuf[pos] = make_int4(qg, __builtin_bit_cast(int, d[t][j]), rowj + rc.td, 0);
                    wbase += __popcll(mask);
                }
            }
    };
    // The wave's column blocks form one sequence across its records.  At block i: the DMA of block i + 1 goes out, then the
    // wait for block i's (everything but the eight loads just issued has landed), then -- at a record's first block -- the
    // rows of the NEXT record and the slot entries of the one after it are requested, then block i is scored.
    int par = 0;
    bool young = false;  // (wave-uniform) the previous block requested rows and slots after its wait
    auto compute = [&](const Rec& rc, const f32x4 (&a)[NT][8], const f32x4 (&bn)[NT], const int (&sq)[PF], const Rec& rn, f32x4 (&an)[NT][8],
                       f32x4 (&bnn)[NT], const int (&sqn)[PF], const Rec& rnn, int (&sqnn)[PF]) __attribute__((always_inline)) {
        int ql = slot_of(rc, sq, 0);  // (its DMA is in flight: issued by the previous record's last block, or by the prologue)
        // (the record's rows are waited for HERE, by the compiler: its wait in front of the first MFMA would also be a wait for
        // the DMA issued just before it.  The last row load issued stands for all of them.)
        if (rc.nq > 0) asm volatile("" ::"v"(bn[NT - 1]));
        for (int cb = 0; cb < rc.nq; cb += 16) {
            int ql_next = 0;
            const bool in_rec = cb + 16 < rc.nq, any = in_rec || rn.nq > 0;  // wave-uniform
            if (in_rec) ql_next = slot_of(rc, sq, cb + 16);
            else if (rn.nq > 0) ql_next = slot_of(rn, sqn, 0);
            // What is younger than this block's DMA: the eight loads of the next block's, and -- if the block before this one
            // was a record's first -- the 22 row and slot loads it requested after its wait (20 counted: a margin of two).
            if (any) {
                dma(ql_next, par ^ 1);
                if (young) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                if (young) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            young = false;
            if (cb == 0) {
                issue_rows(rn, an, bnn);
                issue_slots(rnn, sqnn);
                young = true;
            }
            score(rc, a, bn, par, ql, cb + r < rc.nq);
            par ^= 1;
            ql = ql_next;
        }
        if (rc.nq <= 0) {  // (a zero record, at the tail only: keep the row / slot pipeline moving)
            issue_rows(rn, an, bnn);
            issue_slots(rnn, sqnn);
            young = false;  // (no DMA is in flight across it: the record before had no next block to request)
        }
    };
    f32x4 A0[NT][8], A1[NT][8], BN0[NT], BN1[NT];
    int SQ0[PF], SQ1[PF], SQ2[PF];
    Rec R0 = unpack(rv0), R1 = unpack(rv1), R2 = unpack(rv2);
    issue_slots(R0, SQ0);
    issue_slots(R1, SQ1);
    issue_rows(R0, A0, BN0);
    if (R0.nq > 0) dma(slot_of(R0, SQ0, 0), 0);
    int4 rvn = record(u + 3 * nw);
    // three records in the pipeline: scored | rows travelling | slot entries travelling.  (Unrolled by six: the two row
    // sets and the three slot sets rotate through fixed names.)
    for (;;) {
#define VS_STEP(RC, AC, BC, SC, RN, AN, BNX, SN, RNN, SNN)                    \
        compute(RC, AC, BC, SC, RN, AN, BNX, SN, RNN, SNN);                   \
        u += nw;                                                              \
        if (u >= n_units) break;                                              \
        RC = unpack(rvn); /* the record three steps on takes the free name */ \
        rvn = record(u + 3 * nw);
        VS_STEP(R0, A0, BN0, SQ0, R1, A1, BN1, SQ1, R2, SQ2)
        VS_STEP(R1, A1, BN1, SQ1, R2, A0, BN0, SQ2, R0, SQ0)
        VS_STEP(R2, A0, BN0, SQ2, R0, A1, BN1, SQ0, R1, SQ1)
        VS_STEP(R0, A1, BN1, SQ0, R1, A0, BN0, SQ1, R2, SQ2)
        VS_STEP(R1, A0, BN0, SQ1, R2, A1, BN1, SQ2, R0, SQ0)
        VS_STEP(R2, A1, BN1, SQ2, R0, A0, BN0, SQ0, R1, SQ1)
#undef VS_STEP
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sink_bin_wave(p.sink, wb, wbase, lane);
}

// Exact slow path, one workgroup per query that has no usable bound (or every query when a candidate buffer overflowed:
// masses of duplicate rows): all rows of the query's probed lists, thread-private sorted lists, ranking through LDS.
// (sd, sp: 256 * 16 words of LDS each, from the kernel)
__device__ __forceinline__ void ivf_wide_slow_body(const IvfWideParams& p, const int qg, float* const sd, int* const sp) {
    const int batch = qg >> 5, qi = qg & 31;
    constexpr int KM = 16;
    static_assert(256 * KM <= kCompactCap, "the slow path shares the merge's LDS");
    __shared__ float r_d[4];
    __shared__ int r_p[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = min(p.k, KM);
    const int32_t* pr = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.probes) + (long long)batch * p.probes_batch_bytes) + qi * p.nprobe;
    const float* qv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.q) + (long long)batch * p.q_batch_bytes) + qi * kDim;
    const bool i8 = p.vecs_u8 && p.metric == 0 && p.invalid[batch] == 0;
    const float qn = p.qnorm[qg];
    const int qt = p.qterm[qg];
    float ld[KM];
    int lp[KM];
#pragma unroll
    for (int j = 0; j < KM; ++j) {
        ld[j] = VS_INF;
        lp[j] = 0x7fffffff;
    }
    for (int pp = 0; pp < p.nprobe; ++pp) {
        const int c = pr[pp];
        if (c < 0) continue;
        const int s0 = p.offsets[c], s1 = p.offsets[c + 1];
        for (int row = s0 + tid; row < s1; row += 256) {
            float d;
            if (i8) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const i32x4* b = reinterpret_cast<const i32x4*>(p.vecs_u8 + (int64_t)row * kDim);
                const i32x4* qq = reinterpret_cast<const i32x4*>(p.q8 + (int64_t)qg * kDim);
                int dot = 0;
#pragma unroll
                for (int t = 0; t < kDim / 16; ++t) {
                    const i32x4 bv = b[t], qv4 = qq[t];
#pragma unroll
                    for (int e = 0; e < 4; ++e) dot = __builtin_amdgcn_sdot4(bv[e], qv4[e], dot, false);  // four signed bytes at a time
                }
                d = (float)(qt + p.rterm[row] - 2 * dot);
            } else {
                const float* b = p.vecs + (int64_t)row * kDim;
                float dot = 0.f;
                for (int t = 0; t < kDim; ++t) dot = fmaf(b[t], qv[t], dot);
                d = p.metric ? -dot : fmaf(-2.0f, dot, qn + p.vnorm[row]);
            }
            if (lex_lt(d, row, ld[KM - 1], lp[KM - 1])) list_insert<KM>(ld, lp, d, row);
        }
    }
#pragma unroll
    for (int j = 0; j < KM; ++j) {
        sd[tid * KM + j] = ld[j];
        sp[tid * KM + j] = lp[j];
    }
    __syncthreads();
    float last_d = -VS_INF;
    int last_p = -1;
    for (int round = 0; round < k; ++round) {
        float bd = VS_INF;
        int bp = 0x7fffffff;
        for (int i = tid; i < 256 * KM; i += 256) {
            const float d = sd[i];
            const int ps = sp[i];
            if (ps == 0x7fffffff) continue;
            if (d < last_d || (d == last_d && ps <= last_p)) continue;  // already emitted
            if (lex_lt(d, ps, bd, bp)) {
                bd = d;
                bp = ps;
            }
        }
        float wd;
        int wp;
        wave_lexmin(bd, bp, wd, wp);
        if (lane == 0) {
            r_d[wave] = wd;
            r_p[wave] = wp;
        }
        __syncthreads();
        bd = r_d[0];
        bp = r_p[0];
        for (int w = 1; w < 4; ++w)
            if (lex_lt(r_d[w], r_p[w], bd, bp)) {
                bd = r_d[w];
                bp = r_p[w];
            }
        const bool none = bp == 0x7fffffff;
        if (tid == 0) {
            p.out_d[((int64_t)batch * p.B + qi) * p.k + round] = none ? VS_INF : bd;
            p.out_i[((int64_t)batch * p.B + qi) * p.k + round] = none ? -1 : (p.id_map ? p.id_map[bp] : bp);
        }
        last_d = none ? VS_INF : bd;
        last_p = none ? 0x7fffffff : bp;
        __syncthreads();
    }
    for (int round = k + tid; round < p.k; round += 256) {
        p.out_d[((int64_t)batch * p.B + qi) * p.k + round] = VS_INF;
        p.out_i[((int64_t)batch * p.B + qi) * p.k + round] = -1;
    }
}

int ivf_wide_grid_x(int num_cus, int n_sb) { return std::max(16, num_cus / n_sb); }
int ivf_wide_waves(int num_cus, int n_sb) { return ivf_wide_grid_x(num_cus, n_sb) * n_sb * kIvfWideWaves; }

hipError_t launch_ivf_wide_bounds_plan(const IvfWideParams& p, hipStream_t s, int what) {
    if (p.nlist > kIvfFastNlist || p.nprobe > kIvfMaxProbe || p.k > 16 || p.sb_batches < 1 || p.sb_batches > kIvfWideBatches) return hipErrorInvalidValue;
    const int n_sb = (p.n_batches + p.sb_batches - 1) / p.sb_batches;
    static const int plan_wgs = getenv("VSEARCH_PLAN_WGS") ? atoi(getenv("VSEARCH_PLAN_WGS")) : 16;  // (tuning knob)
    const int n_plan = (what & 2) ? std::max(plan_wgs, 16 / n_sb) : 0;  // (every planning workgroup reads all pair counters, a cache line each)
    const int n_tau = (what & 1) ? p.nlist * kBoundParts : 0;
    if (n_plan * n_sb + n_tau == 0) return hipSuccess;
    if ((what & 1) && (!p.tq || p.n_batches * kMaxBatch > 0x10000 || p.n_batches * kMaxBatch > p.tq_cap)) return hipErrorInvalidValue;
    // (who pads the slot tables: the bounds' workgroups if the tables are complete when they run, i.e. with the plan beside them)
    hipLaunchKernelGGL(ivf_bounds_plan_kernel, dim3(n_plan * n_sb + n_tau), dim3(256), 0, s, p, n_plan, n_sb, n_tau == 0 || n_plan == 0 || n_sb > 16);
    if ((what & 1) && !p.tau_inline) hipLaunchKernelGGL(ivf_tau_combine_kernel, dim3((p.n_batches * kMaxBatch + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_ivf_wide_scan(const IvfWideParams& p, int num_cus, hipStream_t s) {
    if (p.nlist > kIvfFastNlist || p.nprobe > kIvfMaxProbe || p.k > 16 || p.sb_batches < 1 || p.sb_batches > kIvfWideBatches) return hipErrorInvalidValue;
    const int n_sb = (p.n_batches + p.sb_batches - 1) / p.sb_batches;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_scan_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kIvfWideLds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_scan_wide_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kIvfWideF32Lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    static const bool f32_own = !(getenv("VSEARCH_IVF_F32_SHARED") && atoi(getenv("VSEARCH_IVF_F32_SHARED")));  // (A/B knob)
    if (f32_own && (!p.vecs_t8 || p.metric != 0))  // the group is scored in fp32, and the host knows it
        hipLaunchKernelGGL(ivf_scan_wide_f32_kernel, dim3(ivf_wide_grid_x(num_cus, n_sb), n_sb), dim3(kIvfWideF32Threads), kIvfWideF32Lds, s, p);
    else
        hipLaunchKernelGGL(ivf_scan_wide_kernel, dim3(ivf_wide_grid_x(num_cus, n_sb), n_sb), dim3(kIvfWideThreads), kIvfWideLds, s, p);
    return hipGetLastError();
}

// The ranking of the wide pipeline, one workgroup per query: the merge of the query's candidate lists, or -- for a query
// without a usable bound, or for every query when a candidate buffer overflowed -- the exact slow path (as one launch:
// a separate slow-path launch that finds nothing to do still costs its 4 us).
__global__ __launch_bounds__(256) void ivf_wide_rank_kernel(const MergeParams m, const MergeLayout L, const IvfWideParams p) {
    const int q = blockIdx.x;  // output query = batch * B + qi
    const int qg = (q / p.B) * kMaxBatch + q % p.B;
    __shared__ float cd[kCompactCap];
    __shared__ int ci[kCompactCap];
    if (p.sink.overflow[0] || p.slow[qg]) ivf_wide_slow_body(p, qg, cd, ci);  // workgroup-uniform
    else merge_compact_body(m, L, cd, ci, q);
    // Last kernel of the launch group: it leaves the group's counters zeroed for the next group (no memset launch per
    // group).  Every workgroup clears what belongs to its query and a share of the lists' pair counters.
    __syncthreads();
#ifdef VS_STAMPS
    if (p.diag & 128) return;  // diagnostics read the counters afterwards (the API then memsets before every group)
#endif
    const int tid = threadIdx.x;
    if (tid < p.sink.nsub) p.sink.cnt[(int64_t)tid * p.sink.cnt_sub_stride + qg] = 0;
    if (tid == 0) p.slow[qg] = 0;
    // (the workgroups of super-batch sb share out sb's pair counters)
    const int batch = q / p.B, sb = batch / p.sb_batches;
    const int nql = (min(p.n_batches, (sb + 1) * p.sb_batches) - sb * p.sb_batches) * p.B;  // workgroups of this super-batch
    const int ql = (batch - sb * p.sb_batches) * p.B + q % p.B;
    for (int c = ql + tid * nql; c < p.nlist; c += 256 * nql) {
        p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)c * kIvfWideCntStride] = 0;
        if (sb == 0) p.zero[(int64_t)c * kIvfWideCntStride + 1] = 0;  // (the list's bound-table counter lives in super-batch 0's line)
    }
    if (ql == 0 && tid == 0) p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)p.nlist * kIvfWideCntStride] = 0;  // the record count
    // (the words every workgroup reads: `overflow` is cleared by the next group's coarse kernel, the batches' "not byte
    // valued" flags are written as 0 or 1 there; a counter of finished workgroups here would be one contended atomic
    // per query and cost more than the memset it saves -- measured)
}

// The same ranking, four queries per workgroup: one WAVE per query.  A launch group of several super-batches ranks
// thousands of queries, most of them with a few dozen candidates (a rank of a sharded job holds an eighth of every query's
// candidates): a 256-thread workgroup per query then costs what its launch costs (42 us per 8192 queries).  A wave
// gathers its query's candidates (<= 256: four per lane), bounds the k-th best by the k-th smallest lane minimum, and ranks
// what is not above it by counting.  A query with more candidates, a `slow` mark or an overflowed launch is put on a
// list (glist: [0] = count, [1..] = output queries) and taken by ivf_wide_rank_list_kernel behind, with the workgroup-wide
// paths of ivf_wide_rank_kernel -- kept out of this kernel: their registers and LDS would halve its occupancy.
__global__ __launch_bounds__(256) void ivf_wide_rank4_kernel(const MergeParams m, const IvfWideParams p, int32_t* const glist) {
    __shared__ float wcd[4 * 256];
    __shared__ int wci[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = (int)blockIdx.x * 4 + wave;  // output query = batch * B + qi
    if (q >= m.nq) return;
    const int qg = (q / p.B) * kMaxBatch + q % p.B;
    const int nsub = p.sink.nsub, cap = p.sink.cap;
    bool generic = p.sink.overflow[0] != 0 || p.slow[qg] != 0;
    // The sub-lists' lengths (lanes 0 .. nsub - 1).  Four lanes walk a sub-list (lane = sub-list + 16 t takes its entries t,
    // t + 4, ...): no table of where entry e of the concatenation lives, every lane's addresses are its sub-list's own.
    int len = 0;
    if (lane < nsub) len = min(p.sink.cnt[(int64_t)lane * p.sink.cnt_sub_stride + qg], cap + 1);
    generic = generic || __any(len > cap);  // (a sub-list that overflowed marks the query `slow` as well)
    int T = len, longest = len;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        T += __shfl_xor(T, o);
        longest = max(longest, __shfl_xor(longest, o));
    }
    T = __builtin_amdgcn_readfirstlane(T);              // candidates in all
    longest = __builtin_amdgcn_readfirstlane(longest);  // the longest sub-list
    if (!generic) {
        constexpr int V = 8;  // candidates per lane and step
        float d[V];
        int id[V];
        const int my_len = __shfl(len, lane & 15), my_t = lane >> 4;
        const int64_t my_src = ((int64_t)qg * nsub + (lane & 15)) * cap;
        auto load_step = [&](const int first) {  // entries first + t + 4 u (u < V) of the lane's sub-list
#pragma unroll
            for (int u = 0; u < V; ++u) {
                const int e = first + my_t + 4 * u;
                d[u] = VS_INF;
                id[u] = 0x7fffffff;
                if (first + 4 * u < longest && e < my_len) {  // (the first condition is wave-uniform)
                    d[u] = m.part_d[my_src + e];
                    id[u] = m.part_i[my_src + e];
                }
            }
        };
        load_step(0);
        float sd = VS_INF;
        int si = 0x7fffffff;
        int M = 0;
        {
            // bound: the k-th smallest of the 64 lane minima of the first step (k distinct candidates are at least that
            // close); with 64 candidates or fewer there is nothing to filter
            float bound = VS_INF;
            if (T > 64) {
                float md = d[0];
                int mi = id[0];
#pragma unroll
                for (int u = 1; u < V; ++u)
                    if (lex_lt(d[u], id[u], md, mi)) {
                        md = d[u];
                        mi = id[u];
                    }
                const int mrank = wave_rank_count(md, mi, 64);
                const unsigned long long who = __ballot(mrank == min(m.kout, 64) - 1);
                bound = rdlane_f(md, (int)__builtin_ctzll(who | (1ull << 63)));
            }
            float* wd = wcd + wave * 256;
            int* wi = wci + wave * 256;
            // (a query with hundreds of candidates -- a loose bound: one in fifty -- takes its bound from the first step and
            // filters the rest step by step; it used to wait for the list kernel behind this one, 17 us for a handful)
            for (int first = 0; first < longest; first += 4 * V) {
                if (first) load_step(first);
#pragma unroll
                for (int u = 0; u < V; ++u) {
                    if (first + 4 * u >= longest) break;  // wave-uniform
                    const bool pass = id[u] != 0x7fffffff && d[u] <= bound;
                    const unsigned long long mask = __ballot(pass);
                    if (pass) {
                        const int pos = M + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                        if (pos < 256) {
                            wd[pos] = d[u];
                            wi[pos] = id[u];
                        }
                    }
                    M += __popcll(mask);
                }
                if (M > 64) break;  // (masses of equal distances at the bound: the list kernel's)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (M > 64) generic = true;  // masses of equal distances at the bound
            sd = lane < M ? wd[lane] : VS_INF;
            si = lane < M ? wi[lane] : 0x7fffffff;
        }
        if (!generic) {
            const int rank = wave_rank_count(sd, si, min(M, 64));
            if (lane < M && rank < m.kout) {
                m.out_d[(int64_t)q * m.kout + rank] = sd;
                m.out_i[(int64_t)q * m.kout + rank] = m.id_map ? m.id_map[si] : si;
            }
            if (lane >= M && lane < m.kout) {
                m.out_d[(int64_t)q * m.kout + lane] = VS_INF;
                m.out_i[(int64_t)q * m.kout + lane] = -1;
            }
        }
    }
    if (generic) {  // (wave-uniform) left to the list kernel, which also clears this query's counters
        if (lane == 0) glist[1 + atomicAdd(glist, 1)] = q;
        return;
    }
#ifdef VS_STAMPS
    if (p.diag & 128) return;
#endif
    // the group's counters are left zeroed for the next group (see ivf_wide_rank_kernel)
    if (lane < nsub) p.sink.cnt[(int64_t)lane * p.sink.cnt_sub_stride + qg] = 0;
    const int batch = q / p.B, sb = batch / p.sb_batches;
    const int nql = (min(p.n_batches, (sb + 1) * p.sb_batches) - sb * p.sb_batches) * p.B;
    const int ql = (batch - sb * p.sb_batches) * p.B + q % p.B;
    for (int c = ql + lane * nql; c < p.nlist; c += 64 * nql) {
        p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)c * kIvfWideCntStride] = 0;
        if (sb == 0) p.zero[(int64_t)c * kIvfWideCntStride + 1] = 0;  // (the list's bound-table counter lives in super-batch 0's line)
    }
}

// the queries ivf_wide_rank4_kernel left over: a fixed grid walks the list
__global__ __launch_bounds__(256) void ivf_wide_rank_list_kernel(const MergeParams m, const MergeLayout L, const IvfWideParams p, int32_t* const glist) {
    __shared__ float cd[kCompactCap];
    __shared__ int ci[kCompactCap];
    const int n = glist[0];
    const int tid = threadIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int q = glist[1 + i];
        const int qg = (q / p.B) * kMaxBatch + q % p.B;
        if (p.sink.overflow[0] || p.slow[qg]) ivf_wide_slow_body(p, qg, cd, ci);  // workgroup-uniform
        else merge_compact_body(m, L, cd, ci, q);
        __syncthreads();
#ifdef VS_STAMPS
        if (p.diag & 128) continue;
#endif
        if (tid < p.sink.nsub) p.sink.cnt[(int64_t)tid * p.sink.cnt_sub_stride + qg] = 0;
        if (tid == 0) p.slow[qg] = 0;
        const int batch = q / p.B, sb = batch / p.sb_batches;
        const int nql = (min(p.n_batches, (sb + 1) * p.sb_batches) - sb * p.sb_batches) * p.B;
        const int ql = (batch - sb * p.sb_batches) * p.B + q % p.B;
        for (int c = ql + (tid & 63) * nql; c < p.nlist; c += 64 * nql) {
            p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)c * kIvfWideCntStride] = 0;
            if (sb == 0) p.zero[(int64_t)c * kIvfWideCntStride + 1] = 0;  // (the list's bound-table counter lives in super-batch 0's line)
        }
        __syncthreads();
    }
}

hipError_t launch_ivf_wide_rank(const MergeParams& m, int64_t stride_g, int64_t stride_q, const IvfWideParams& p, hipStream_t s, int32_t* glist) {
    if (m.kout < 1 || m.G < 1 || m.nq != p.n_batches * p.B || (int64_t)m.G * m.kin > kCompactCap || m.q_group_out != p.B || m.q_group_in != kMaxBatch)
        return hipErrorInvalidValue;
    MergeLayout L{stride_g, stride_q};
    // a wave per query where a launch ranks thousands of queries (several super-batches); a workgroup per query otherwise
    if (glist && m.nq > 2048 && m.G <= 16 && m.flat_len && m.flat_len_sub_stride && !m.flags && !m.tau_out && !m.invalid && !m.run_if) {
        hipLaunchKernelGGL(ivf_wide_rank4_kernel, dim3((m.nq + 3) / 4), dim3(256), 0, s, m, p, glist);
        hipLaunchKernelGGL(ivf_wide_rank_list_kernel, dim3(256), dim3(256), 0, s, m, L, p, glist);
    }
    else hipLaunchKernelGGL(ivf_wide_rank_kernel, dim3(m.nq), dim3(256), 0, s, m, L, p);
    return hipGetLastError();
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
