// vs_scan_one.hip -- the single-call brute-force scan: ONE launch per query batch (cpu_baseline.cpp:222-254 runs one
// query per iteration; BASELINE configs[1] is batch = 1).
//
//   scan_one_kernel<NQH, KCAP>: Q[<= 16 NQH x 128] x base^T on v_mfma_f32_16x16x4_f32 with the L2 epilogue of
//   cpu_baseline.cpp:239-242 and select_topk's k smallest (:127-153) fused in, for calls too short to amortise the seed
//   launches of the streaming scans.  What it does without:
//     * no seed launches, no threshold exchange: a lane keeps its own sorted list of KCAP entries per query column and
//       looks at a distance only if it is below the list's last entry -- with 8192 lane streams per query the lists
//       settle after a few tiles (a lane inserts about k ln(n / k) of its n = N / 8192 rows);
//     * no second launch: every workgroup ranks its lanes' lists into one sorted partial list per query, and the
//       workgroup that arrives last (one atomic per workgroup) merges the partial lists into the result.
//   Data path = the streaming scans': per-wave ring of two 16-row tile slots in LDS filled by LDS-DMA (`nt`), XOR
//   swizzle on the source side, hand-counted vmcnt; tiles dealt statically (wave w of workgroup b takes tiles
//   b + (w + 8 n) G: at step n the 2048 waves of the grid read 2048 consecutive tiles) and no tile is fetched that is
//   not used -- with one pass over the rows per launch two discarded tail prefetches per wave would be 6 % of the bytes.
//   Bound: HBM (516 MB per call at SIFT-1M whatever the batch size).
#include "vs_kernels.h"
#include "vs_dev.h"

namespace vs {

namespace {

constexpr int kOneSlotBytes = kTileRows * kDim * 4 + 256;         // 16 rows + their norms
constexpr int kOneRing = kScanWaves * 2 * kOneSlotBytes;          // 135168
constexpr int kOneScratch = 8192;                                 // per-wave query norms [8][32] | counters [32] | bounds [32] | flag | lane minima [32][32]
constexpr int kOneLds = kOneRing + kOneScratch;

// k smallest of M (dist, id) pairs by ascending (dist, id), read through `get(e, d, id)`; ids are unique.  One wave,
// nothing is modified: round r takes the smallest pair above round r - 1's.  emit(round, d, id, none).
template <class Get, class Emit>
__device__ __forceinline__ void wave_rank_rounds(int M, int rounds, int lane, Get get, Emit emit) {
    float last_d = -VS_INF;
    int last_i = -1;
    bool first = true;
    for (int round = 0; round < rounds; ++round) {
        float md = VS_INF;
        int mi = 0x7fffffff;
        for (int e = lane; e < M; e += 64) {
            float d;
            int id;
            get(e, d, id);
            if (id < 0) continue;
            if (!first && (d < last_d || (d == last_d && id <= last_i))) continue;  // emitted already
            if (lex_lt(d, id, md, mi)) {
                md = d;
                mi = id;
            }
        }
        float bd;
        int bi;
        wave_lexmin(md, mi, bd, bi);
        const bool none = bi == 0x7fffffff;
        emit(round, bd, bi, none);
        if (none) {
            for (int r2 = round + 1; r2 < rounds; ++r2) emit(r2, VS_INF, 0x7fffffff, true);
            return;
        }
        last_d = bd;
        last_i = bi;
        first = false;
    }
}

}  // namespace

#ifdef VS_STAMPS
#define ONE_STAMP(i)                                                                                   \
    do {                                                                                               \
        if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff); \
    } while (0)
#else
#define ONE_STAMP(i)
#endif

template <int NQH, int KCAP>
__global__ __launch_bounds__(kScanThreads, 2) void scan_one_kernel(const OneParams p) {
    constexpr int TR = kTileRows;
    constexpr int NQ = NQH * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qn_w = reinterpret_cast<float*>(smem + kOneRing);       // [8 waves][32]
    int* lds_cnt = reinterpret_cast<int*>(qn_w + kScanWaves * 32);  // [32]
    unsigned* lds_bound = reinterpret_cast<unsigned*>(lds_cnt + 32);  // [32] ordered-float bits
    int* lds_flag = reinterpret_cast<int*>(lds_bound + 32);        // [1]
    int* lds_ticket = lds_flag + 1;                                 // [1] next tile ticket of the workgroup
    float* lds_lmin = reinterpret_cast<float*>(lds_flag + 32);      // [32 queries][32 lanes of the workgroup holding that column]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int G = (int)gridDim.x;
    const int tiles_total = (int)((p.n_rows + TR - 1) / TR);
    // Run n of the base = tiles [n G, n G + G): the grid reads one run at about the same time, every workgroup one tile of
    // it.  A workgroup's ticket n (its tile of run n) goes to whichever of its waves asks next (an LDS counter: a wave
    // that is served late by HBM takes fewer tiles); the first two tickets of every wave are fixed: wave and wave + 8.
    // (a workgroup stays on one residue b of the runs.  Moving through the residues -- tile n G + (b + 17 n) mod G -- was
    // measured: the slowest workgroup then finished 60 % later than the fastest instead of 25 %)
    // (p.reverse: the runs are walked from the last to the first.  Calls alternate, so a call starts on the rows the previous
    // one read last -- what of them the 256 MB Infinity Cache still holds is not fetched from HBM again: 85.4 against 87.2 us
    // per call at 1 M rows; with the rows loaded without `nt` the effect is larger, 88.6 against 93.9, but the level worse)
    // (only the FULL runs change places; an incomplete last run stays the last ticket: a workgroup's tickets are valid up
    // to some n and invalid from there on, which is what the loop below relies on)
    const int n_full = tiles_total / G;
    auto tile_of = [&](int n) { return ((p.reverse && n < n_full) ? n_full - 1 - n : n) * G + (int)blockIdx.x; };
    const int t_first = tile_of(wave), t_second = tile_of(wave + kScanWaves);

    // ---- data path (see scan_f32s_kernel): LDS-DMA pieces as instructions, scalar tile base + the lane's 32-bit offset
    char* ring = smem + wave * (2 * kOneSlotBytes);
    unsigned voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row_in = 2 * j + (lane >> 5);
        voff[j] = (unsigned)(row_in * 512 + 16 * ((lane & 31) ^ row_in));
    }
    const unsigned voff_n = (unsigned)lane * 4u;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ring);
    auto issue_tile = [&](int tile, int slot) __attribute__((always_inline)) {
        const int64_t row0 = (int64_t)tile * TR;
        const unsigned dst = ring_lds + (unsigned)(slot * kOneSlotBytes);
        const char* tb = reinterpret_cast<const char*>(p.base) + row0 * (kDim * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(dst), "v"(voff[j]), "s"(tb), "n"(j * 1024) : "memory", "scc");
        const char* nb = reinterpret_cast<const char*>(p.bnorm + row0);  // (the norm array is padded by 64)
        asm volatile("s_add_u32 m0, %0, 8192\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(dst), "v"(voff_n), "s"(nb) : "memory", "scc");
    };
    // ---- the queries as MFMA B operands: qf[h][c][i] = Q[16 h + r][16 c + 4 g + i]; padding queries (main.cpp:206-211)
    // are zero columns that nothing looks at.  Every wave for itself.  The loads go out FIRST and as instructions the
    // compiler does not see as memory operations (it would wait for them with vmcnt(0), i.e. for the two tiles requested
    // right behind them as well: the tile loop then started 3 us later).  ONE statement waits for them -- always exactly
    // 18 younger pieces: a wave without a first or second tile fetches tile 0 into the slot -- and has every loaded
    // register as an in/out operand: nothing that reads them can be placed above it (copies of a loaded register that
    // the compiler made ahead of a separate wait statement were seen to break a test at random).
    f32x4 qf[NQH][8];
#pragma unroll
    for (int h = 0; h < NQH; ++h) {
        const int qsafe = min(16 * h + r, p.nq_valid - 1);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float* pc = p.q + (int64_t)qsafe * kDim + 16 * c + 4 * g;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[h][c]) : "v"(pc) : "memory");
        }
    }
    ONE_STAMP(0);
    const bool have0 = t_first < tiles_total, have1 = t_second < tiles_total;  // (wave-uniform)
    issue_tile(have0 ? t_first : 0, 0);
    issue_tile(have1 ? t_second : 0, 1);
#define VS_TIE8(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
    if constexpr (NQH == 1) asm volatile("s_waitcnt vmcnt(18)" : VS_TIE8(qf[0]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(18)" : VS_TIE8(qf[0]), VS_TIE8(qf[NQH - 1]) : : "memory");
#undef VS_TIE8
#pragma unroll
    for (int h = 0; h < NQH; ++h)
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (16 * h + r >= p.nq_valid) qf[h][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ||q||^2 in the reference's order (cpu_baseline.cpp:95-114): FMA lane j = e mod 8 accumulates Q[e]^2 over e = j,
    // j + 8, ..., then r0 + r1 + ... + r7 left to right.  Element e = 16 c + 4 g + i sits in lane group g: lane j's chain
    // alternates between groups g = j / 4 (steps 2 c) and g + 2 (steps 2 c + 1), so group g fetches group g + 2's
    // fragments (the lane 32 further on), runs the four chains j = 4 g + i, and group 0 then adds its four sums and the
    // four of group 1 (the lane 16 further on) in order.  No second set of loads, no LDS.
    float qn[NQH], tau[NQH];
    bool live[NQH];
#pragma unroll
    for (int h = 0; h < NQH; ++h) {
        float chain[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 other;
#pragma unroll
            for (int i = 0; i < 4; ++i) other[i] = __shfl(qf[h][c][i], (lane + 32) & 63);
#pragma unroll
            for (int i = 0; i < 4; ++i) chain[i] = fmaf(qf[h][c][i], qf[h][c][i], chain[i]);  // step 2 c
#pragma unroll
            for (int i = 0; i < 4; ++i) chain[i] = fmaf(other[i], other[i], chain[i]);          // step 2 c + 1
        }
        float s = ((chain[0] + chain[1]) + chain[2]) + chain[3];  // r0 .. r3 in group 0 (r4 .. r7 in group 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) s = s + __shfl(chain[i], (lane + 16) & 63);  // group 0: + r4, + r5, + r6, + r7
        qn[h] = __shfl(s, r);  // group 0's lane of this column
    }
#pragma unroll
    for (int h = 0; h < NQH; ++h) {
        live[h] = 16 * h + r < p.nq_valid;
        tau[h] = live[h] ? VS_INF : -VS_INF;  // a padding column never takes a candidate
    }
    // the lanes' current minima, shared by the workgroup (see the tile loop)
    for (int i = tid; i < 32 * 33; i += kScanThreads) lds_lmin[i] = VS_INF;  // (rows 33 apart: bank spread)
    if (tid == 0) lds_ticket[0] = 2 * kScanWaves;
    __syncthreads();
    ONE_STAMP(1);
    float ld[NQH][KCAP];
    int li[NQH][KCAP];
#pragma unroll
    for (int h = 0; h < NQH; ++h)
#pragma unroll
        for (int j = 0; j < KCAP; ++j) {
            ld[h][j] = VS_INF;
            li[h][j] = -1;
        }

    unsigned fa[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) fa[c] = (unsigned)(wave * (2 * kOneSlotBytes) + r * 512 + (((4 * c + g) ^ r) << 4));
    const unsigned fa_n = (unsigned)(wave * (2 * kOneSlotBytes) + 8192 + 16 * g);
    const int last_row = (int)p.n_rows - 1;

    // t_cur sits in slot `sl` (landed or landing), t_nxt in the other slot: nine DMA pieces behind it if it is a tile at all
    int t_cur = t_first, t_nxt = t_second;
    auto step = [&](const int sl) __attribute__((always_inline)) {
        const int tile = t_cur;
        int tk = 0;
        if (lane == 0) tk = atomicAdd(lds_ticket, 1);
        const int t_new = tile_of(__builtin_amdgcn_readfirstlane(tk));
        if (t_nxt < tiles_total) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* src = smem + sl * kOneSlotBytes;
        f32x4 a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = *reinterpret_cast<const f32x4*>(src + fa[c]);
        const f32x4 bn = *reinterpret_cast<const f32x4*>(src + fa_n);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t_new < tiles_total) issue_tile(t_new, sl);  // the slot is refilled as soon as its fragments sit in registers
        f32x4 acc[NQH];
#pragma unroll
        for (int h = 0; h < NQH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < NQH; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[h][c][i], acc[h], 0, 0, 0);
        const int rbase = tile * TR + 4 * g;
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            float d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // cpu_baseline.cpp:241  dist = qn + bn - 2*dot  (gcc contracts to fnmadd(2, dot, qn+bn))
                const float l2 = fmaf(-2.0f, acc[h][j], qn[h] + bn[j]);
                d[j] = p.metric ? -acc[h][j] : l2;
            }
            if (tile * TR + TR - 1 > last_row) {  // wave-uniform: only the last tile holds rows past the end
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (rbase + j > last_row) d[j] = VS_INF;
            }
            const float dmin = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
            if (dmin < tau[h]) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (d[j] < tau[h]) {
                        list_insert<KCAP>(ld[h], li[h], d[j], rbase + j + p.id_offset);
                        tau[h] = fminf(tau[h], ld[h][KCAP - 1]);
                    }
                lds_lmin[(16 * h + r) * 33 + 4 * wave + g] = ld[h][0];  // this lane's best so far, for the shared bound
            }
        }
        t_cur = t_nxt;
        t_nxt = t_new;
    };
    // The bound that makes insertions rare: the k1-th smallest of the 32 lane minima of a query column in this workgroup
    // (k1 distinct rows are at least that close; a lane alone sees N / 8192 rows and would insert a fifth of them, and
    // with 64 lanes deciding independently some lane inserts at nearly every value).  Taken at a few checkpoints, from
    // whatever the other waves have published by then: a missing or stale minimum only loosens it.  Lane (r, g) reads
    // 8 of its column's 32 minima, the 4 lanes of the column take k1 rounds of "smallest not taken yet".
    auto shared_bound = [&](float (&out)[NQH]) __attribute__((always_inline)) {
        // (cheap form: each of the column's 4 lanes takes the t-th smallest of its 8 minima, t = ceil(k1 / 4); the
        // largest of the four has 4 t >= k1 minima at or below it.  The exact k1-th smallest of the 32 costs k1 rounds of
        // cross-lane traffic and was measured at 1.5 us per call and column block)
        const int t = (p.k1 + 3) >> 2;
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = lds_lmin[(16 * h + r) * 33 + 8 * g + i];
            float m = VS_INF;
            for (int round = 0; round < t; ++round) {
                m = fminf(fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), fminf(fminf(v[4], v[5]), fminf(v[6], v[7])));
                bool done = false;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (!done && v[i] == m) {
                        v[i] = VS_INF;
                        done = true;
                    }
            }
            float x = fmaxf(m, __shfl_xor(m, 16));
            x = fmaxf(x, __shfl_xor(x, 32));
            out[h] = x;  // (+inf while a lane holds fewer than t minima)
        }
    };
    int n = 0;
    while (t_cur < tiles_total) {  // (step() advances t_cur)
        step(0);
        if (t_cur < tiles_total) step(1);
        if (n == 0 || n == 2 || n == 6 || n == 14) {  // wave-uniform
            float sb[NQH];
            shared_bound(sb);
#pragma unroll
            for (int h = 0; h < NQH; ++h)
                if (live[h] && sb[h] < VS_INF) tau[h] = fminf(tau[h], next_up(sb[h]));
        }
        n += 2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ONE_STAMP(2);

    // ---- workgroup merge.  A query's 32 lane lists (4 lane groups x 8 waves): the k1-th smallest lane minimum bounds the
    // workgroup's k1-th best (+inf with fewer than k1 rows); entries not above it are compacted into LDS (the ring is
    // free now: room for every entry, so nothing can overflow) and ranked by one wave per query.
    __syncthreads();  // (every wave is done with the ring; the lane minima are final)
    if (tid < 32) lds_cnt[tid] = 0;
    constexpr int CAP = 32 * KCAP;
    float* cand_d = reinterpret_cast<float*>(smem);
    int* cand_i = reinterpret_cast<int*>(smem + (size_t)NQ * CAP * sizeof(float));
    static_assert((size_t)NQ * CAP * 8 <= kOneRing, "the merge buffers overlay the ring");
    float wg_bound[NQH];
    shared_bound(wg_bound);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NQH; ++h) {
        const int qidx = 16 * h + r;
#pragma unroll
        for (int j = 0; j < KCAP; ++j)
            if (li[h][j] >= 0 && ld[h][j] <= wg_bound[h]) {
                const int pos = atomicAdd(&lds_cnt[qidx], 1);
                cand_d[qidx * CAP + pos] = ld[h][j];
                cand_i[qidx * CAP + pos] = li[h][j];
            }
    }
    __syncthreads();
    ONE_STAMP(3);
    // partial lists: part[query][workgroup][k1], sorted by (dist, id), padded with (+inf, -1); written with agent-scope
    // (write-through) stores, one per lane
    for (int qq = wave; qq < p.nq_valid; qq += kScanWaves) {
        float* od = p.part_d + ((int64_t)qq * G + blockIdx.x) * p.k1;
        int32_t* oi = p.part_i + ((int64_t)qq * G + blockIdx.x) * p.k1;
        auto put = [&](int slot, float d, int id) {
            __hip_atomic_store(od + slot, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(oi + slot, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        const int M = lds_cnt[qq];
        if (M <= 64) {
            const float d = lane < M ? cand_d[qq * CAP + lane] : VS_INF;
            const int id = lane < M ? cand_i[qq * CAP + lane] : 0x7fffffff;
            const int rank = wave_rank_count(d, id, M);
            if (lane < M && rank < p.k1) put(rank, d, id);
            if (lane >= M && lane < p.k1) put(lane, VS_INF, -1);
        } else {
            wave_rank_rounds(
                M, p.k1, lane,
                [&](int e, float& d, int& id) {
                    d = cand_d[qq * CAP + e];
                    id = cand_i[qq * CAP + e];
                },
                [&](int round, float d, int id, bool none) {
                    if (lane == 0) put(round, none ? VS_INF : d, none ? -1 : id);
                });
        }
    }
    // ---- the workgroup that arrives last merges the partial lists.  No fences: the partial lists are written with
    // agent-scope stores that have completed (vmcnt(0)) before the workgroup's arrival is counted, and read with
    // agent-scope loads -- a release / acquire fence pair here writes back and invalidates the whole L2 of the XCD under
    // the workgroups that are still streaming (measured: 126 instead of 100 us per call at 1 M rows)
    ONE_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ONE_STAMP(5);
    if (tid == 0) lds_flag[0] = (__hip_atomic_fetch_add(p.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) ? 1 : 0;
    __syncthreads();
    ONE_STAMP(6);
    if (!lds_flag[0]) return;  // workgroup-uniform
    if (tid == 0) __hip_atomic_store(p.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch (stream ordered)
    // ONE acquire fence in ONE workgroup (every other workgroup has arrived, nobody is streaming any more): lines of the
    // partial lists this XCD's L2 may still hold from an earlier launch are dropped, plain cached loads below.  (Agent-scope
    // loads instead bypass the caches one dword at a time: 16 K of them took 35 us.)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ONE_STAMP(8);
    auto ld_d = [](const float* a) { return *a; };
    auto ld_i = [](const int32_t* a) { return *a; };
    // Every memory round trip here is a microsecond or two of the call's latency, so a query costs two of them: (1) the
    // FIRST entry of each of its G partial lists (a wave reads 256 at once); their k1-th smallest bounds the k1-th best --
    // k1 distinct rows are at least that close (the lists' k1-th entries would bound it too, but a hundred times looser:
    // the smallest "sixth best of a 256th of the rows" is about the 400th best of all).  (2) the lists whose first entry
    // is not above the bound -- about k1 of them -- are read whole, all of them at once, one (list, entry) pair per lane.
    // What is not above the bound (a handful) is ranked by counting.  One wave per query; the wave's queries' first
    // entries are requested together up front.
    constexpr int QPW = kMaxBatch / kScanWaves;       // queries per wave at most: 4
    constexpr int SPL = kSlotStride / 64;             // lists per lane and query at most: 4
    constexpr int FCAP = 256;                         // candidates per query kept in LDS
    float* fin_d = reinterpret_cast<float*>(smem) + wave * FCAP;
    int* fin_i = reinterpret_cast<int*>(smem + kScanWaves * FCAP * sizeof(float)) + wave * FCAP;
    int* pl = reinterpret_cast<int*>(smem + 2 * kScanWaves * FCAP * sizeof(float)) + wave * kSlotStride;  // passing lists of the current query
    float fd[QPW][SPL];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int qq = wave + kScanWaves * qi;
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const int w = lane + 64 * u;
            fd[qi][u] = VS_INF;
            if (qq < p.nq_valid && 64 * u < G) fd[qi][u] = ld_d(p.part_d + ((int64_t)qq * G + min(w, G - 1)) * p.k1);
            if (w >= G) fd[qi][u] = VS_INF;
        }
    }
    ONE_STAMP(9);
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int qq = wave + kScanWaves * qi;
        if (qq >= p.nq_valid) break;  // wave-uniform
        const float* pd = p.part_d + (int64_t)qq * G * p.k1;
        const int32_t* pi = p.part_i + (int64_t)qq * G * p.k1;
        float* od = p.out_d + (int64_t)qq * p.k1;
        int32_t* oi = p.out_i + (int64_t)qq * p.k1;
        const float b = wave_kth_smallest(fd[qi][0], fd[qi][1], fd[qi][2], fd[qi][3], p.k1, lane);  // (+inf with fewer than k1 rows in all)
        // the passing lists, compacted (ballot prefix), then one (list, entry) pair per lane
        int P = 0;
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const bool pass = fd[qi][u] < VS_INF && fd[qi][u] <= b;
            const unsigned long long mask = __ballot(pass);
            if (pass) pl[P + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0))] = lane + 64 * u;
            P += __popcll(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int pairs = P * p.k1;
        int M = 0;
        for (int e0 = 0; e0 < pairs; e0 += 64) {
            const int e = e0 + lane;
            float d = VS_INF;
            int id = -1;
            if (e < pairs) {
                const int off = pl[e / p.k1] * p.k1 + e % p.k1;
                d = ld_d(pd + off);
                id = ld_i(pi + off);
            }
            const bool pass = id >= 0 && d <= b;
            const unsigned long long mask = __ballot(pass);
            const int pos = M + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (pass && pos < FCAP) {
                fin_d[pos] = d;
                fin_i[pos] = id;
            }
            M += __popcll(mask);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef VS_STAMPS
        if (p.dbg && lane == 0 && qq == 0) p.dbg[blockIdx.x * 16 + 11] = M;
#endif
        if (M <= 64) {
            const float d = lane < M ? fin_d[lane] : VS_INF;
            const int id = lane < M ? fin_i[lane] : 0x7fffffff;
            const int rank = wave_rank_count(d, id, M);
            if (lane < M && rank < p.k1) {
                od[rank] = d;
                oi[rank] = id;
            }
            if (lane >= M && lane < p.k1) {
                od[lane] = VS_INF;
                oi[lane] = -1;
            }
            // two of the k1 outputs at one distance (select_topk's order among them is history dependent: the caller's flag)
            bool tie = false;
            for (int j = 0; j < M; ++j) {
                const float dj = rdlane_f(d, j);
                const int rj = __builtin_amdgcn_readlane(rank, j);
                tie = tie || (j != lane && dj == d && rj < p.k1);
            }
            const bool any_tie = __any(lane < M && rank < p.k1 && tie);
            if (lane == 0 && p.flags) p.flags[qq] = any_tie ? 1 : 0;
        } else {
            float prev = VS_INF;
            int tie = 0;
            auto emit = [&](int round, float d, int id, bool none) {
                if (!none && round > 0 && d == prev) tie = 1;
                prev = none ? VS_INF : d;
                if (lane == 0) {
                    od[round] = none ? VS_INF : d;
                    oi[round] = none ? -1 : id;
                }
            };
            if (M <= FCAP) {
                wave_rank_rounds(M, p.k1, lane, [&](int e, float& d, int& id) { d = fin_d[e]; id = fin_i[e]; }, emit);
            } else {  // masses of equal distances at the bound: rank straight from the partial lists
                wave_rank_rounds(G * p.k1, p.k1, lane, [&](int e, float& d, int& id) { d = ld_d(pd + e); id = ld_i(pi + e); }, emit);
            }
            if (lane == 0 && p.flags) p.flags[qq] = tie;
        }
        __builtin_amdgcn_wave_barrier();  // (fin / pl are reused by the wave's next query)
    }
#ifdef VS_STAMPS
    __syncthreads();
    ONE_STAMP(7);
#endif
}

int scan_one_grid(int64_t n_rows, int num_cus) {
    const int64_t tiles = (n_rows + kTileRows - 1) / kTileRows;
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(num_cus, kSlotStride), (tiles + kScanWaves - 1) / kScanWaves));
}

template <int NQH, int KCAP>
static hipError_t launch_one_t(const OneParams& p, int grid, hipStream_t s) {
    auto kfn = scan_one_kernel<NQH, KCAP>;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kOneLds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(kScanThreads), kOneLds, s, p);
    return hipGetLastError();
}

hipError_t launch_scan_one(const OneParams& p, int grid, hipStream_t s) {
    if (p.nq_valid < 1 || p.nq_valid > kMaxBatch || p.k1 < 1 || p.k1 > 16 || grid < 1 || grid > kSlotStride || p.n_rows < 1) return hipErrorInvalidValue;
    const int kcap = p.k1 <= 8 ? 8 : 16;
    if (p.nq_valid <= 16) return kcap == 8 ? launch_one_t<1, 8>(p, grid, s) : launch_one_t<1, 16>(p, grid, s);
    return kcap == 8 ? launch_one_t<2, 8>(p, grid, s) : launch_one_t<2, 16>(p, grid, s);
}

}  // namespace vs
