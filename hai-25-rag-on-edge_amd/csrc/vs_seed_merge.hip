// vs_seed_merge.hip -- bounds for multi-batch scans from a sample of the rows (launch_seed) and the ranking / merge of
// candidate and partial lists (merge_compact_kernel, merge_kernel); see vs_kernels.h.
#include "vs_kernels.h"
#include "vs_dev.h"
#include "vs_merge.h"
#include <type_traits>
#include <algorithm>

namespace vs {

// ------------------------------------------------------------------------------------------------
// Seed bounds (see SeedParams).  seed_qnorm_kernel: ||q||^2 in the reference's order (+ the queries as bytes for the
// int8 paths); seed_kernel: minima of 64 groups of 32 sample tiles per (batch, query); seed_tau_kernel: k1-th smallest
// of a query's 64 group minima.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seed_qnorm_kernel(const SeedParams p) {
    // blockIdx.y = which of the batch's outputs: 0 norms (+ bytes, terms, verdict), 1 the fp32 fragments, 2 the byte fragments
    // (one workgroup doing all three was three memory round trips in a row on every call's critical path)
    const int batch = blockIdx.x, what = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int row = tid >> 3, j = tid & 7;
    const float* qb = p.q + (int64_t)batch * p.q_batch_stride;
    if (what == 3) {
        // the launch's zeroed block (overflow word, list counters: see bf_launch) -- a memset launch of its own otherwise.
        // The batches' "not byte valued" words [16, 48) belong to the workgroups of part 0, which write them as 0 or 1.
        const int per = (p.zero_words + (int)gridDim.x - 1) / (int)gridDim.x;
        for (int i = batch * per + tid; i < min(p.zero_words, (batch + 1) * per); i += 256)
            if (!p.invalid || i < 16 || i >= 48) p.zero[i] = 0;
        return;
    }
    if (what == 1) {
        if (!p.qfrag) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // B-fragment order for the fp32 streaming scan (see SeedParams)
            const int idx = tid + 256 * u;  // (h, c, lane)
            const int fl = idx & 63, c = (idx >> 6) & 7, hh = idx >> 9;
            const int qrow = 16 * hh + (fl & 15);
            f32x4 v = *reinterpret_cast<const f32x4*>(qb + min(qrow, p.nq_valid - 1) * kDim + 16 * c + 4 * (fl >> 4));
            if (qrow >= p.nq_valid) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(p.qfrag + ((int64_t)batch * 1024 + idx) * 4) = v;
        }
        return;
    }
    if (what == 2) {
        if (!p.q8frag) return;
        // the byte queries in B-fragment order (see SeedParams): thread = (h, half, lane), 16 bytes each
        const int fl = tid & 63, half = (tid >> 6) & 1, hh = tid >> 7;
        const int qrow = 16 * hh + (fl & 15);
        int w[4] = {0, 0, 0, 0};
        f32x4 x[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) x[v] = *reinterpret_cast<const f32x4*>(qb + min(qrow, p.nq_valid - 1) * kDim + 64 * half + 16 * (fl >> 4) + 4 * v);
        if (qrow < p.nq_valid) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) w[v] |= (((int)x[v][e] - 128) & 0xff) << (8 * e);
        }
        *reinterpret_cast<int4*>(p.q8frag + ((int64_t)batch * 256 + tid) * 16) = make_int4(w[0], w[1], w[2], w[3]);
        return;
    }
    float acc = 0.f;
    int part = 0;        // sum(q - 128) over this thread's 16 elements
    bool q_ok = true;    // ... all of them integers in [0, 255]
    if (row < p.nq_valid) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float x = qb[row * kDim + 8 * i + j];
            acc = fmaf(x, x, acc);
            if (p.q8) {
                const int xi = (int)x;
                q_ok = q_ok && ((float)xi == x) && xi >= 0 && xi <= 255;
                part += xi - 128;
                p.q8[((int64_t)batch * kMaxBatch + row) * kDim + 8 * i + j] = (int8_t)(xi - 128);
            }
        }
    } else if (p.q8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) p.q8[((int64_t)batch * kMaxBatch + row) * kDim + 8 * i + j] = 0;  // padding queries
    }
    const int b8 = lane & ~7;
    float sum = __shfl(acc, b8);
#pragma unroll
    for (int u = 1; u < 8; ++u) sum = sum + __shfl(acc, b8 + u);
    if (j == 0) p.qnorm[batch * kMaxBatch + row] = sum;
    if (p.q8) {
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        if (j == 0) p.qterm[batch * kMaxBatch + row] = (int)sum - 256 * part - 4194304;
    }
    if (p.invalid) {  // (workgroup-uniform) written as 0 or 1: nobody has to clear it before
        const int bad = __syncthreads_or(q_ok ? 0 : 1);
        if (tid == 0) p.invalid[batch] = bad ? 1 : 0;
    }
}

// One wave = (batch, chunk of kSeedTilesPerWave sample tiles): the batch's queries stay in registers as the B
// operand, the tiles stream through (they are shared by all batches: L2 / Infinity Cache hits).
constexpr int kSeedChunks = 64;                                // group minima per query
constexpr int kSeedTilesPerWave = kSeedWaves / kSeedChunks;    // 32 tiles = 512 rows per group
__device__ __forceinline__ int64_t seed_tile(int64_t tiles_total, int s) {  // sample tile s -> tile of the shard
    return tiles_total >= kSeedWaves ? (int64_t)s * (tiles_total / kSeedWaves) : s;
}

// this wave's share (8 of the group's 32 sample tiles) on the fp32 rows -> per-query minima m[h] of column 16 h + r
__device__ __forceinline__ void seed_body_f32(const SeedParams& p, int batch, int chunk, int wv, int r, int g, float (&m)[2]) {
    const int64_t tiles_total = (p.n_rows + kTileRows - 1) / kTileRows;
    const float* qb = p.q + (int64_t)batch * p.q_batch_stride;
    f32x4 qf[2][8];
    float qn[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int qrow = h * 16 + r;
        const bool qv = qrow < p.nq_valid;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            qf[h][c] = *reinterpret_cast<const f32x4*>(qb + (qv ? qrow : 0) * kDim + 16 * c + 4 * g);
            if (!qv) qf[h][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        qn[h] = p.qnorm[batch * kMaxBatch + qrow];
    }
    // two tiles per step: their loads go out together (one at a time the loop would pay the cache latency per tile)
    constexpr int U = 2;
    for (int t0 = wv * (kSeedTilesPerWave / 4); t0 < (wv + 1) * (kSeedTilesPerWave / 4); t0 += U) {
        f32x4 a[U][8], bn[U];
        int64_t row0[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int smp = chunk * kSeedTilesPerWave + t0 + u;
            const int64_t tile = seed_tile(tiles_total, smp);
            ok[u] = tile < tiles_total;  // wave-uniform
            row0[u] = (ok[u] ? tile : 0) * kTileRows;
            if (p.sample_f32) {  // compact copy in fragment order (1 KB per instruction)
                const float* sp = p.sample_f32 + ((int64_t)smp * 8 * 64 + (16 * g + r)) * 4;
#pragma unroll
                for (int c = 0; c < 8; ++c) a[u][c] = *reinterpret_cast<const f32x4*>(sp + c * 64 * 4);
                bn[u] = *reinterpret_cast<const f32x4*>(p.sample_bnorm + smp * 16 + 4 * g);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) a[u][c] = *reinterpret_cast<const f32x4*>(p.base + (row0[u] + r) * kDim + 16 * c + 4 * g);
                bn[u] = *reinterpret_cast<const f32x4*>(p.bnorm + row0[u] + 4 * g);  // padded by 64
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c][i], qf[h][c][i], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = p.metric ? -acc[j] : fmaf(-2.0f, acc[j], qn[h] + bn[u][j]);  // the scan's own expression
                    if (row0[u] + 4 * g + j < p.n_rows) m[h] = fminf(m[h], d);
                }
            }
        }
    }
}

// the same on the exact int8 copy (rows and queries integers in [0, 255]): the same distances as exact integers, see
// scan_kernel PREC = 1; the queries as bytes and their constant terms come from seed_qnorm_kernel
__device__ __forceinline__ void seed_body_i8(const SeedParams& p, int batch, int chunk, int wv, int r, int g, float (&m)[2]) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int64_t tiles_total = (p.n_rows + kTileRows - 1) / kTileRows;
    i32x4 qi[2][2];
    int qterm[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int qrow = h * 16 + r;
        const int8_t* src = p.q8 + ((int64_t)batch * kMaxBatch + qrow) * kDim;  // padding queries are all-zero rows
        qi[h][0] = *reinterpret_cast<const i32x4*>(src + 16 * g);
        qi[h][1] = *reinterpret_cast<const i32x4*>(src + 64 + 16 * g);
        qterm[h] = p.qterm[batch * kMaxBatch + qrow];
    }
    constexpr int U = 4;  // four tiles per step
    for (int t0 = wv * (kSeedTilesPerWave / 4); t0 < (wv + 1) * (kSeedTilesPerWave / 4); t0 += U) {
        i32x4 a0[U], a1[U], rt[U];
        int64_t row0[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int smp = chunk * kSeedTilesPerWave + t0 + u;
            const int64_t tile = seed_tile(tiles_total, smp);
            ok[u] = tile < tiles_total;  // wave-uniform
            row0[u] = (ok[u] ? tile : 0) * kTileRows;
            // A fragments: bytes k = 16 g .. 16 g + 15 and 64 + 16 g .. of row row0 + r
            if (p.sample_u8) {  // compact copy in fragment order (1 KB per instruction)
                const int8_t* sp = p.sample_u8 + ((int64_t)smp * 2 * 64 + (16 * g + r)) * 16;
                a0[u] = *reinterpret_cast<const i32x4*>(sp);
                a1[u] = *reinterpret_cast<const i32x4*>(sp + 64 * 16);
                rt[u] = *reinterpret_cast<const i32x4*>(p.sample_rterm + smp * 16 + 4 * g);
            } else {
                a0[u] = *reinterpret_cast<const i32x4*>(p.base_u8 + (row0[u] + r) * kDim + 16 * g);
                a1[u] = *reinterpret_cast<const i32x4*>(p.base_u8 + (row0[u] + r) * kDim + 64 + 16 * g);
                rt[u] = *reinterpret_cast<const i32x4*>(p.rterm + row0[u] + 4 * g);  // padded by 64
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                i32x4 acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[u], qi[h][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[u], qi[h][1], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (row0[u] + 4 * g + j < p.n_rows) m[h] = fminf(m[h], (float)(qterm[h] + rt[u][j] - 2 * acc[j]));
            }
        }
    }
}

// A batch's bounds from its 64 group minima per query: a wave per query (n_waves waves share the 32 queries), lane l holds
// group minimum l, k1 rounds of a wave minimum.
__device__ __forceinline__ void seed_tau_body(const SeedParams& p, const int batch, const int lane, const int wave, const int n_waves) {
    const float* src = p.wmin + (int64_t)batch * kSeedChunks * kMaxBatch;
    const int per = kMaxBatch / n_waves;
    float v[8];  // (per <= 8)
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) v[qq] = qq < per ? src[lane * kMaxBatch + per * wave + qq] : VS_INF;
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
        if (qq >= per) break;
        float kth = VS_INF;
        for (int round = 0; round < p.k1; ++round) {
            kth = wave_min_f32(v[qq]);
            const unsigned long long msk = __ballot(v[qq] == kth);
            if (msk != 0ull && lane == __builtin_ctzll(msk)) v[qq] = VS_INF;  // drop exactly one instance
        }
        if (lane == 0) p.tau0[batch * kMaxBatch + per * wave + qq] = kth < VS_INF ? next_up(kth) : VS_INF;
    }
}

// One workgroup per (batch, group of 32 sample tiles); its four waves take 8 tiles each (a quarter of the serial chain
// of one wave per group) and fold their minima through LDS.  A batch whose queries are byte valued uses the exact int8
// copy of the rows when there is one; any other batch (workgroup-uniform choice) the fp32 rows.
__global__ __launch_bounds__(256) void seed_kernel(const SeedParams p) {
    __shared__ float wm[4][kMaxBatch];
    const int lane = threadIdx.x & 63;
    const int wv = (int)(threadIdx.x >> 6);
    const int batch = (int)blockIdx.x / kSeedChunks, chunk = (int)blockIdx.x % kSeedChunks;
    const int r = lane & 15, g = lane >> 4;
    float m[2] = {VS_INF, VS_INF};
    if (p.base_u8 && p.invalid[batch] == 0) seed_body_i8(p, batch, chunk, wv, r, g, m);
    else seed_body_f32(p, batch, chunk, wv, r, g, m);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        m[h] = fminf(m[h], __shfl_xor(m[h], 16));
        m[h] = fminf(m[h], __shfl_xor(m[h], 32));
    }
    if (g == 0) {
        wm[wv][r] = m[0];
        wm[wv][16 + r] = m[1];
    }
    __syncthreads();
    if (threadIdx.x < kMaxBatch) {
        float* dst = p.wmin + ((int64_t)batch * kSeedChunks + chunk) * kMaxBatch;  // 128 contiguous bytes per group
        dst[threadIdx.x] = fminf(fminf(wm[0][threadIdx.x], wm[1][threadIdx.x]), fminf(wm[2][threadIdx.x], wm[3][threadIdx.x]));
    }
    // (The bounds by the batch's last-arriving workgroup instead of seed_tau_kernel's launch was measured: agent-scope stores,
    // the wait for them and 64 atomics per batch made this kernel take 28 instead of 14 us per 20 batches.)
}

__global__ __launch_bounds__(1024) void seed_tau_kernel(const SeedParams p) {
    seed_tau_body(p, blockIdx.x, threadIdx.x & 63, threadIdx.x >> 6, 16);
}

// Gathers sample tile blockIdx.x of the shard into the compact, fragment-ordered arrays of SeedParams (index creation).
__global__ __launch_bounds__(256) void seed_sample_kernel(const float* __restrict__ base, const float* __restrict__ bnorm,
                                                          const int8_t* __restrict__ base_u8, const int32_t* __restrict__ rterm,
                                                          int64_t n_rows, float* sample_f32, float* sample_bnorm, int8_t* sample_u8,
                                                          int32_t* sample_rterm) {
    const int smp = blockIdx.x, tid = threadIdx.x;
    const int64_t tiles_total = (n_rows + kTileRows - 1) / kTileRows;
    const int64_t tile = seed_tile(tiles_total, smp);
    const int64_t row0 = (tile < tiles_total ? tile : 0) * kTileRows;
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // fp32: (c, lane) -> row r = lane & 15, floats 16 c + 4 (lane >> 4) ..
        const int idx = tid + 256 * u, fl = idx & 63, c = idx >> 6;
        const int64_t row = min(row0 + (fl & 15), n_rows - 1);  // (rows past the end are masked by the seed itself)
        *reinterpret_cast<f32x4*>(sample_f32 + ((int64_t)smp * 512 + idx) * 4) =
            *reinterpret_cast<const f32x4*>(base + row * kDim + 16 * c + 4 * (fl >> 4));
    }
    if (tid < 16) sample_bnorm[smp * 16 + tid] = bnorm[min(row0 + tid, n_rows - 1)];
    if (sample_u8) {
        if (tid < 128) {  // bytes: (half, lane) -> row r, bytes 64 half + 16 (lane >> 4) ..
            const int fl = tid & 63, half = tid >> 6;
            const int64_t row = min(row0 + (fl & 15), n_rows - 1);
            *reinterpret_cast<int4*>(sample_u8 + ((int64_t)smp * 128 + tid) * 16) =
                *reinterpret_cast<const int4*>(base_u8 + row * kDim + 64 * half + 16 * (fl >> 4));
        }
        if (tid < 16) sample_rterm[smp * 16 + tid] = rterm[min(row0 + tid, n_rows - 1)];
    }
}

hipError_t launch_seed_sample(const float* base, const float* bnorm, const int8_t* base_u8, const int32_t* rterm, int64_t n_rows,
                              float* sample_f32, float* sample_bnorm, int8_t* sample_u8, int32_t* sample_rterm, hipStream_t s) {
    hipLaunchKernelGGL(seed_sample_kernel, dim3(kSeedWaves), dim3(256), 0, s, base, bnorm, base_u8, rterm, n_rows, sample_f32,
                       sample_bnorm, base_u8 ? sample_u8 : nullptr, sample_rterm);
    return hipGetLastError();
}

hipError_t launch_seed(const SeedParams& p, hipStream_t s) {
    hipLaunchKernelGGL(seed_qnorm_kernel, dim3(p.n_batches, p.zero ? 4 : 3), dim3(256), 0, s, p);
    const int wgs = p.n_batches * kSeedChunks;  // one workgroup per (batch, group of sample tiles)
    hipLaunchKernelGGL(seed_kernel, dim3(wgs), dim3(256), 0, s, p);
    hipLaunchKernelGGL(seed_tau_kernel, dim3(p.n_batches), dim3(1024), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Merge of sorted lists.  One 256-thread workgroup per query; thread t owns lists t, t+256, ...
// (LPT of them).  kout rounds of a workgroup-wide lexicographic argmin over the list heads.
// part_i == nullptr means "the id of entry (g, j) is g*kin + j" (used to pick probes out of a
// score matrix: G = nlist lists of length 1).
// ------------------------------------------------------------------------------------------------
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
            if (p.flag_empty && !(outd[0] < VS_INF)) f = 2;
            if (p.shard_flags)
                for (int g = 0; g < p.G; ++g)
                    if (p.shard_flags[(int64_t)g * p.shard_flags_stride + q] == 2) f = 2;
            p.flags[q] = f;
        }
        if (p.tau_out) {
            const float kth = outd[n - 1];
            p.tau_out[q] = kth < VS_INF ? next_up(kth) : VS_INF;
        }
    }
}

static hipError_t launch_merge_heads(const MergeParams& p, const MergeLayout& L, hipStream_t s) {
    const int lpt = (p.G + 255) / 256;
    if (lpt <= 1) hipLaunchKernelGGL(merge_kernel<1>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 2) hipLaunchKernelGGL(merge_kernel<2>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 4) hipLaunchKernelGGL(merge_kernel<4>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 8) hipLaunchKernelGGL(merge_kernel<8>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 16) hipLaunchKernelGGL(merge_kernel<16>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 32) hipLaunchKernelGGL(merge_kernel<32>, dim3(p.nq), dim3(256), 0, s, p, L);
    else if (lpt <= 64) hipLaunchKernelGGL(merge_kernel<64>, dim3(p.nq), dim3(256), 0, s, p, L);  // (G <= 16384: the probe pick of the nlist > 4096 fallback)
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Compacting merge (fast path, G*kin <= kCompactCap).  With the threshold exchange most partial
// lists are empty, so the finite entries are first compacted into LDS (one atomic append each);
// a handful of candidates is then ranked by a single wave with DPP/shuffle argmin rounds and no
// barriers.  Larger candidate sets use workgroup-wide rounds over the LDS array.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_compact_kernel(const MergeParams p, const MergeLayout L) {
    __shared__ float cd[kCompactCap];
    __shared__ int ci[kCompactCap];
    merge_compact_body(p, L, cd, ci, blockIdx.x);
}

hipError_t launch_merge_layout(const MergeParams& p, int64_t stride_g, int64_t stride_q, hipStream_t s) {
    if (p.kout < 1 || p.G < 1 || p.nq < 1) return hipErrorInvalidValue;
    MergeLayout L{stride_g, stride_q};
    if ((int64_t)p.G * p.kin <= kCompactCap) {
        hipLaunchKernelGGL(merge_compact_kernel, dim3(p.nq), dim3(256), 0, s, p, L);
        return hipGetLastError();
    }
    return launch_merge_heads(p, L, s);
}

hipError_t launch_merge(const MergeParams& p, hipStream_t s) {
    // scan partial layout: [G][nq_stride][kin]
    return launch_merge_layout(p, (int64_t)p.nq_stride * p.kin, p.kin, s);
}

}  // namespace vs
