// vs_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the distance + top-k hot path.
//
//   scan_f32s_kernel<NB> : the graded brute-force path: streaming fp32 scan of Q[32 x 128] x base^T on
//                          v_mfma_f32_16x16x4_f32 with the L2 epilogue (cpu_baseline.cpp:229-242), NB batches per pass
//                          over the rows, candidates under seeded bounds to per-wave buffers; scan_i8w_kernel: the same
//                          on exact u8 rows (v_mfma_i32_16x16x64_i8), four batches per pass.
//   scan_kernel          : per-batch scan with the top-k (cpu_baseline.cpp:127-153) fused in (short calls, fallback);
//                          kModeStore = the B x N score matrix of QnnRunner::executeBatchRaw, kModeAssign = k-means
//                          assignment for the index builder, kModeFilter = tie-resolver candidates.
//   seed_*_kernel        : bounds for a multi-batch scan from 2048 sample tiles, queries in MFMA fragment order (launch_seed).
//   merge_compact_kernel : ranking of candidate lists, cross-workgroup / cross-GPU merge (merge_kernel: general fallback).
//   row_sqnorm_kernel    : compute_norms (cpu_baseline.cpp:95-125) in the reference's summation order.
//   ivf_coarse_mfma_kernel, ivf_pick_kernel, ivf_tau_plan_kernel, ivf_scan_wide_kernel, ivf_wide_rank_kernel :
//                          IVFIndex::searchBatch (IVFIndex.cpp:640-859) as a list-major pipeline over launch groups of up
//                          to 32 batches (one pass over the probed lists per group);
//                          ivf_group_plan_kernel / ivf_unit_scan_kernel / ivf_select_kernel (one pass per batch),
//                          ivf_list_scan_kernel / pick_probes_kernel / ivf_scan_kernel: earlier and fallback paths.
//   kpp_*_kernel, kmeans_*_kernel : index builder (create_ivf_model_reordered.py:88-118).
//
// (The UFIXED_POINT_8 score path of the reference's device runner -- quantiser, uint8 score matrix, top-k over it --
//  lives in vs_q8.hip with its own C ABI.)
//
// Wavefront = 64 lanes everywhere; nothing here is written for 32-wide warps.  In the MFMA-bound scans every ordinary
// vector instruction costs the SIMD about 8 cycles of MFMA pipe and an LDS-DMA instruction about 50
// (scripts/microbench/mfma_f32_ceiling.hip): their tile loops keep vector work to one fma + one compare per value and
// issue the LDS-DMA as instructions (scalar base + lane offset, M0 by scalar add).
#include "vs_kernels.h"
#include "vs_dev.h"
#include <type_traits>
#include <algorithm>

namespace vs {


// ------------------------------------------------------------------------------------------------
// Brute-force scan: Q[<=32 x 128] x base^T on v_mfma_f32_16x16x4_f32, L2 epilogue and top-k fused.
//
// Workgroup = 8 waves (2 per SIMD).  A wave owns whole 16-row base tiles.  In the MFMA the base
// tile is the A operand (lane (r = l & 15, g = l >> 4) supplies base[row0 + r][16 t + 4 g + i] to
// step (t, i)) and the queries are the B operand with the same k permutation, so in the 16x16
// result a lane holds ONE query column (l & 15) and four base rows (4 g + reg): the top-k state of
// a query is lane-private and needs no cross-lane traffic until the workgroup is done.
//
// Data path.  Every wave owns a private ring of DEPTH tile slots in LDS and fills them with
// LDS-DMA (global_load_lds_dwordx4: one wave-instruction moves two whole 512-byte rows, fully
// coalesced, no VGPR destination).  The only VMEM operations in the loop are those DMA pieces and
// they are counted by hand (s_waitcnt vmcnt(9*(DEPTH-1))): nothing drains the queue, and no
// barrier is needed because a wave reads only what it loaded itself.  LDS image of a slot:
// 16 rows x 512 B with 16-byte chunk c of row r stored at chunk c ^ r (XOR applied to the DMA
// *source* address, the LDS side stays lane-linear), which makes the ds_read_b128 of the A
// fragments bank-conflict free; followed by the tile's squared norms.
//
// Everything a batch needs is inside this one launch:
//   * query zero-padding (main.cpp:206-211) and squared norms in the reference's summation order
//     (cpu_baseline.cpp:95-114, :211) -- queries are staged once through LDS;
//   * the threshold exchange: after its first tile round every workgroup publishes, per query,
//     the smallest distance it has seen (write-through stores into slots[query][workgroup]); soon
//     after it reads the published minima (one coalesced 1 KB row per query) and takes the k1-th
//     smallest as an upper bound tau0 of the final k1-th best distance: k1 distinct rows are known
//     to be at least that close.  From then on a distance is looked at only if it is below
//     min(tau0, own lane's KCAP-th best), so the insertion path goes cold.  Nothing waits for
//     anybody: an unpublished slot reads +inf and merely loosens the bound, so the result never
//     depends on timing, placement or residency;
//   * the workgroup merge: surviving candidates (d < tau0) are compacted into LDS and ranked with
//     DPP reductions; the per-workgroup sorted lists go to the cross-workgroup merge kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kSlotBytes = kTileRows * kDim * 4 + 256;  // 16 rows + 64 norms
constexpr int kDepth = 2;
constexpr int kRingBytes = kScanWaves * kDepth * kSlotBytes;  // 135168
constexpr int kQStageBytes = 32 * kDim * 4;                  // 16384
constexpr int kScratchBytes = 2048;
constexpr int kMergeSmall = 32;     // entries per query of the small workgroup-merge buffer
constexpr int kMergeSmallBytes = kMaxBatch * kMergeSmall * 8;  // 8192
constexpr int kScanLds = kRingBytes + kQStageBytes + kScratchBytes + kMergeSmallBytes;  // 161792 <= 160 KiB (163840)

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

// PREC = 0: fp32 rows, v_mfma_f32_16x16x4_f32, 16-row tiles.
// PREC = 1: u8 rows stored as (x - 128) int8, v_mfma_i32_16x16x64_i8, 64-row tiles; exact for integer-valued data in
//           [0, 255] (SIFT): dist = qterm + rterm - 2 * sum((q-128)(b-128)) in int32, then converted (< 2^24).
#ifndef VS_ROW_CPOL
#define VS_ROW_CPOL 2  // nt: the rows are streamed once per batch and 512 MB never fits a cache
#endif
template <int NQH, int KCAP, int MODE, int PREC>
__global__ __launch_bounds__(kScanThreads, 2) void scan_kernel(const ScanParams p) {
    static_assert(PREC == 0 || MODE == kModeTopK, "the int8 data path only serves the top-k scan");
    constexpr int TR = PREC ? 64 : kTileRows;  // rows per tile (a slot is 8 KB of rows + 256 B of row terms either way)
    constexpr int NRG = TR / 16;               // 16-row MFMA blocks per tile
    constexpr int NKEEP = PREC ? 2 : 3;        // warm-up tiles whose distances are only kept
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q_s = reinterpret_cast<float*>(smem + kRingBytes);           // [32][128], chunk-swizzled
    float* lds_qn = reinterpret_cast<float*>(smem + kRingBytes + kQStageBytes);  // [32]
    float* lds_tau = lds_qn + 32;                                       // [32]
    float* lds_wmin = lds_tau + 32;                                     // [8][32]
    int* lds_flag = reinterpret_cast<int*>(lds_wmin + kScanWaves * 32); // [1]
    int* lds_cnt = lds_flag + 4;                                        // [32]
    int* lds_ticket = lds_cnt + 32;                                     // [1]
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform
    VS_STAMP(0);
    if (p.run_if && !p.run_if[0]) return;  // fallback launch behind a streaming scan that did not overflow

    const int64_t n_rows = p.row_end - p.row_begin;
    const int tiles_total = (int)((n_rows + TR - 1) / TR);
    // Tiles are dealt round-robin over the workgroups (ticket n of workgroup b is tile b + n*G): while
    // the workgroups run in lock-step (start of every batch) they then read CONSECUTIVE tiles, which
    // spread over all HBM channels.  Contiguous per-workgroup chunks put every workgroup on the same
    // few channels at those moments (chunk stride = 245 tiles aliases 4-way at SIFT-1M).
    const int tile0 = blockIdx.x;
    const int tile_step = gridDim.x;
    const int tile1 = tiles_total;
    const int64_t last_row = p.row_end - 1;
    const int tlast = max(tiles_total - 1, 0);

    // One persistent launch serves n_batches query batches back to back (no launch gaps, no grid fill/drain per
    // batch).  The workgroups only meet in the threshold exchange, which waits for half of them.
    bool q_staged = false;  // the next batch's queries are already in (or on their way to) the LDS stage
    bool tiles_staged = false;  // ... and so are this wave's first two tiles (slot 0 and slot 1)
#pragma clang loop unroll(disable)
    for (int batch = 0; batch < p.n_batches; ++batch) {
    // Lane-derived values are re-derived per batch from an opaque copy: otherwise hipcc hoists dozens
    // of address registers out of the batch loop, they stay live across everything and the kernel
    // falls off its 256-VGPR budget into scratch.
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int r = lane & 15;
    const int g = lane >> 4;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    char* ring = smem + wave * (kDepth * kSlotBytes);
    // DMA source mapping: piece j (0..7) writes LDS chunks [64 j, 64 j + 64) of the slot.
    //   fp32: lane l lands at row 2 j + (l >> 5), stored chunk (l & 31)  <-  source chunk (l & 31) ^ row
    //   int8: 128-byte rows, piece j moves rows 8j..8j+7; lane l lands at row 8j + (l >> 3), stored chunk l & 7
    //         <-  source chunk (l & 7) ^ ((row >> 1) & 7)
    // (the XOR makes the ds_read_b128 of the A fragments conflict free).  The per-lane byte offsets inside a
    // tile are fixed, so a DMA is "scalar tile base + 32-bit lane offset" with no address arithmetic in the
    // loop: while the other wave of the SIMD streams MFMAs, every extra VALU instruction here costs about one
    // MFMA slot.  Rows past row_end are fetched unclamped (every row array has kScanPadRows spare rows) and
    // masked in the epilogue of the last tile.
    unsigned voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (PREC == 0) {
            const int row_in = 2 * j + (lane >> 5);
            voff[j] = (unsigned)(row_in * 512 + 16 * ((lane & 31) ^ row_in));
        } else {
            const int row_in = 8 * j + (lane >> 3);
            voff[j] = (unsigned)(row_in * 128 + 16 * ((lane & 7) ^ ((row_in >> 1) & 7)));
        }
    }
    const unsigned voff_n = (unsigned)lane * 4u;
    auto issue_tile = [&](int tile, int slot) __attribute__((always_inline)) {
        const int64_t row0 = p.row_begin + (int64_t)tile * TR;
        char* dst = ring + slot * kSlotBytes;
        const char* tb = PREC ? reinterpret_cast<const char*>(p.base_u8) + row0 * kDim
                              : reinterpret_cast<const char*>(p.base) + row0 * (kDim * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned vo = voff[j];
            asm volatile("" : "+v"(vo));  // keep the zero-extension here: "scalar base + 32-bit lane offset" addressing
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + vo),
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0,
                                             PREC ? 0 : VS_ROW_CPOL);
        }
        // norms (fp32) / row terms (int8) of rows row0 .. row0+63; both arrays are padded by 64
        const char* nb = PREC ? reinterpret_cast<const char*>(p.rterm + row0) : reinterpret_cast<const char*>(p.bnorm + row0);
        unsigned vn = voff_n;
        asm volatile("" : "+v"(vn));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nb + vn),
                                         (__attribute__((address_space(3))) void*)(dst + 8192), 4, 0, 0);
    };

    // Tiles of the workgroup's chunk are handed out through an LDS ticket counter, so a wave that
    // is served late by HBM simply takes fewer tiles: the eight private streams stay balanced.
    // Tickets 0..15 are pre-assigned (wave, wave + 8); the counter starts at 16.
    //
    // DMA queue bookkeeping (vmcnt counts LDS-DMA, loads and stores together, in issue order):
    // "s_waitcnt vmcnt(N)" is placed where exactly N younger operations follow the data needed.
    auto lds_barrier = [&]() {  // workgroup barrier that leaves the DMA queue alone
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const float* qb = p.q + (int64_t)batch * p.q_batch_stride;
    float* slots = p.slots_cur ? p.slots_cur + (int64_t)batch * 32 * kSlotStride : nullptr;
    // queries -> LDS by DMA as well (2 pieces per wave): chunk c of row q lands at chunk c ^ (q & 15);
    // rows >= nq_valid read row 0 and are zeroed when used (main.cpp:206-211 zero padding)
    auto issue_queries = [&](const float* qsrc) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * wave + u;
            const int row = 2 * j + (lane >> 5);
            const int c4 = lane & 31;
            const float* src = qsrc + (row < p.nq_valid ? row : 0) * kDim + 4 * (c4 ^ (row & 15));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(q_s) + j * 1024),
                                             16, 0, 0);
        }
    };
    // (from the second batch on the queries were staged during the previous batch, see below)
    if (!q_staged) issue_queries(qb);
    q_staged = false;
    // A wave's first two tiles are the same in every batch; from the second batch on they were fetched by the tail of
    // the previous batch's loop (instead of two prefetches that would be thrown away) and are already in the ring.
    int t_a = tile0 + wave * tile_step, t_b = tile0 + (wave + kScanWaves) * tile_step;
    if (!tiles_staged) {
        issue_tile(min(t_a, tlast), 0);
        issue_tile(min(t_b, tlast), 1);
    }
    tiles_staged = MODE != kModeStore;
    // a ticket past the end of the batch: what goes into the slot is the tile the NEXT batch starts with in it
    auto issue_or_stage = [&](int t, int sl) __attribute__((always_inline)) {
        issue_tile(t < tile1 ? t : min(sl == 0 ? t_a : t_b, tlast), sl);
    };
    if (tid == 0) lds_ticket[0] = 2 * kScanWaves;
    if (tid < 32) lds_cnt[tid] = 0;
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  // the two query pieces have landed (two tiles follow)
    lds_barrier();
    // squared norms in the reference's AVX2 order (8 FMA lanes, then r0+...+r7): threads 0..255
    if (tid < 256) {
        const int row = tid >> 3, j = tid & 7;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float x = q_s[row * kDim + 4 * ((2 * i + (j >> 2)) ^ (row & 15)) + (j & 3)];
            x = row < p.nq_valid ? x : 0.f;
            acc = fmaf(x, x, acc);
        }
        const int b8 = lane & ~7;
        float sum = __shfl(acc, b8);
#pragma unroll
        for (int u = 1; u < 8; ++u) sum = sum + __shfl(acc, b8 + u);
        if (j == 0) lds_qn[row] = sum;
    }
    // query fragments (B operand): qf[h][c][i] = Q[16 h + r][16 c + 4 g + i]
    f32x4 qf[PREC ? 1 : NQH][PREC ? 1 : 8];
    i32x4 qi8[NQH][2];   // int8 path: bytes (q - 128) of k = 16 g + j and 64 + 16 g + j
    int qpart[NQH];      // int8 path: this lane's share of sum(q - 128)
    bool q_ok = true;    // int8 path: every query element is an integer in [0, 255]
    if (PREC == 0) {
#pragma unroll
        for (int h = 0; h < NQH; ++h)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(q_s + (h * 16 + r) * kDim + 4 * ((4 * c + g) ^ r));
                qf[h][c] = (h * 16 + r) < p.nq_valid ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    } else {
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            const bool live = (h * 16 + r) < p.nq_valid;
            int part = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                i32x4 packed;
#pragma unroll
                for (int w = 0; w < 4; ++w) {  // 4 floats -> one dword of 4 signed bytes
                    const int c16 = (half * 16 + 4 * g + w);  // 16-byte float chunk index: k = 4 * c16 .. + 3
                    const f32x4 v = *reinterpret_cast<const f32x4*>(q_s + (h * 16 + r) * kDim + 4 * (c16 ^ r));
                    unsigned word = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = live ? v[e] : 128.f;  // padding queries become all-zero int8 rows
                        const int xi = (int)x;
                        q_ok = q_ok && ((float)xi == x) && xi >= 0 && xi <= 255;
                        const int sb = xi - 128;
                        part += sb;
                        word |= ((unsigned)(sb & 0xff)) << (8 * e);
                    }
                    packed[w] = (int)word;
                }
                qi8[h][half] = packed;
            }
            qpart[h] = part;
        }
    }
    lds_barrier();
    float qn[NQH], tau[NQH], tq[NQH];
    int qterm[NQH];
#pragma unroll
    for (int h = 0; h < NQH; ++h) {
        qn[h] = lds_qn[h * 16 + r];
        tau[h] = VS_INF;
        tq[h] = VS_INF;
        qterm[h] = 0;
    }
    if (PREC == 1) {
        // sum(q - 128) over the 4 lanes (g = 0..3) that share a query column; qterm = ||q||^2 - 256 sum - 2 * 128 * 128^2
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            int sq = qpart[h];
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            qterm[h] = (int)qn[h] - 256 * sq - 4194304;
        }
        // a batch with a non-integer query cannot use this path: flag it, skip it (the caller reruns it in fp32)
        const bool all_ok = __all(q_ok);
        if (wave == 0 && lane == 0) lds_flag[0] = 1;
        lds_barrier();
        if (!all_ok && lane == 0) lds_flag[0] = 0;
        lds_barrier();
        if (!lds_flag[0]) {  // workgroup-uniform
            if (blockIdx.x == 0 && tid == 0 && p.invalid) p.invalid[batch] = 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            continue;
        }
    }
    VS_STAMP(1);
    // The query stage is free from here on: stage the NEXT batch's queries now (two more DMA pieces per wave, older
    // than every tile that will be waited for, so the counted waits below only ever wait a little longer).
    if (MODE != kModeStore && batch + 1 < p.n_batches) {
        issue_queries(qb + p.q_batch_stride);
        q_staged = true;
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

    // distances of one tile: d[rg][h][j] for query column 16 h + r, base rows 16 rg + 4 g + j
    // LDS byte offsets of this lane's A fragments inside slot 0 of its wave's ring (slot 1: + kSlotBytes, an
    // immediate once the slot is a compile-time constant): no address arithmetic in the loop
    unsigned fa[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (PREC == 0) {
            fa[c] = (unsigned)(wave * (kDepth * kSlotBytes) + r * 512 + (((4 * c + g) ^ r) << 4));
        } else {
            // c = 2 rg + half: row 16 rg + r, chunk (4 half + g) ^ ((row >> 1) & 7)
            const int row_in = 16 * (c >> 1) + r;
            fa[c] = (unsigned)(wave * (kDepth * kSlotBytes) + row_in * 128 + ((((c & 1) * 4 + g) ^ ((row_in >> 1) & 7)) << 4));
        }
    }
    const unsigned fa_n = (unsigned)(wave * (kDepth * kSlotBytes) + 8192 + 16 * g);  // norms / row terms of rows 4g..4g+3 (+16 rg)
    // distances of one tile: d[rg][h][j] for query column 16 h + r, base rows 16 rg + 4 g + j
    // after_frags(): called once the tile's fragments have been requested from LDS (the caller waits for them and may
    // then refill the slot while the MFMAs run)
    auto tile_distances = [&](int tt, int slot, float (&d)[NRG][NQH][4], auto after_frags) __attribute__((always_inline)) {
        const char* src = smem + slot * kSlotBytes;
        if (PREC == 0) {
            f32x4 a[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] = *reinterpret_cast<const f32x4*>(src + fa[c]);
            const f32x4 bn = *reinterpret_cast<const f32x4*>(src + fa_n);
            after_frags();
            f32x4 acc[NQH];
#pragma unroll
            for (int h = 0; h < NQH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < NQH; ++h)
                        acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[h][c][i], acc[h], 0, 0, 0);
            // all nine LDS reads first, then the MFMA stream (the waits become counted lgkmcnt(N))
            __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 32 * NQH, 0);
#pragma unroll
            for (int h = 0; h < NQH; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // cpu_baseline.cpp:241  dist = qn + bn - 2*dot  (gcc contracts to fnmadd(2, dot, qn+bn))
                    const float l2 = fmaf(-2.0f, acc[h][j], qn[h] + bn[j]);
                    d[0][h][j] = p.metric ? -acc[h][j] : l2;
                }
        } else {
            i32x4 a0[NRG], a1[NRG], rtv[NRG];
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
                a0[rg] = *reinterpret_cast<const i32x4*>(src + fa[2 * rg]);
                a1[rg] = *reinterpret_cast<const i32x4*>(src + fa[2 * rg + 1]);
                rtv[rg] = *reinterpret_cast<const i32x4*>(src + fa_n + 64 * rg);
            }
            after_frags();
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
#pragma unroll
                for (int h = 0; h < NQH; ++h) {
                    i32x4 acc = (i32x4){0, 0, 0, 0};
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[rg], qi8[h][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[rg], qi8[h][1], acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // the same integer the fp32 path computes exactly: ||q||^2 + ||b||^2 - 2 q.b
                        const int di = qterm[h] + rtv[rg][j] - 2 * acc[j];
                        d[rg][h][j] = (float)di;
                    }
                }
            }
        }
        if (tt >= tlast) {  // only the last tile can hold rows past row_end (wave-uniform branch)
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
                const int64_t rbase = p.row_begin + (int64_t)tt * TR + 16 * rg + 4 * g;
#pragma unroll
                for (int h = 0; h < NQH; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (rbase + j > last_row) d[rg][h][j] = VS_INF;
            }
        }
    };
    auto consume = [&](int tt, const float (&dd)[NRG][NQH][4]) {
      if (MODE == kModeTopK) {
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) {
            const int64_t rbase = p.row_begin + (int64_t)tt * TR + 16 * rg + 4 * g;
#pragma unroll
            for (int h = 0; h < NQH; ++h) {
                const float dmin = fminf(fminf(dd[rg][h][0], dd[rg][h][1]), fminf(dd[rg][h][2], dd[rg][h][3]));
                if (dmin < tau[h]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (dd[rg][h][j] < tau[h]) {
                            list_insert<KCAP>(ld[h], li[h], dd[rg][h][j], (int)(rbase + j) + p.id_offset);
                            tau[h] = fminf(tau[h], ld[h][KCAP - 1]);
                        }
                }
            }
        }
        return;
      }
      if (MODE == kModeFilter) {
        // candidate rows for the exact replay of select_topk: everything under the query's bound (a few per million)
        const int rloc = tt * TR + 4 * g;
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            const float dmin = fminf(fminf(dd[0][h][0], dd[0][h][1]), fminf(dd[0][h][2], dd[0][h][3]));
            if (dmin < tau[h]) {
                const int qidx = h * 16 + r;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (dd[0][h][j] < tau[h]) {
                        const int pos = atomicAdd(p.f_cnt + qidx, 1);
                        if (pos < p.f_cap) {
                            p.f_row[(int64_t)qidx * p.f_cap + pos] = (int)p.row_begin + rloc + j;
                            p.f_d[(int64_t)qidx * p.f_cap + pos] = dd[0][h][j];
                        }
                    }
            }
        }
        return;
      }
        const float (&d)[NQH][4] = dd[0];
        const int64_t rbase = p.row_begin + (int64_t)tt * TR + 4 * g;
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            if (MODE == kModeStore) {
                const int qidx = h * 16 + r;
                if (qidx < p.nq_valid) {
                    float* dst = p.store + (int64_t)qidx * p.store_ld + (rbase - p.row_begin);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (rbase + j <= last_row) dst[j] = d[h][j];
                }
            }
        }
        if (MODE == kModeAssign) {
            // k-means assignment: "queries" are a block of 32 centroids; every base row keeps its nearest
            // centroid so far in best_d/best_i.  A lane holds rows 4g..4g+3 for columns r and 16+r: fold
            // its columns, then the 16 lanes of the DPP row (same rows, different columns).
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float bd = VS_INF;
                int bi = 0x7fffffff;
#pragma unroll
                for (int h = 0; h < NQH; ++h) {
                    const int cid = p.assign_base + batch * kMaxBatch + h * 16 + r;
                    if (h * 16 + r < p.nq_valid && lex_lt(d[h][j], cid, bd, bi)) {
                        bd = d[h][j];
                        bi = cid;
                    }
                }
                float od;
                int oi;
                od = dpp_mov_f<0xB1>(bd); oi = dpp_mov_i<0xB1>(bi);
                if (lex_lt(od, oi, bd, bi)) { bd = od; bi = oi; }
                od = dpp_mov_f<0x4E>(bd); oi = dpp_mov_i<0x4E>(bi);
                if (lex_lt(od, oi, bd, bi)) { bd = od; bi = oi; }
                od = dpp_mov_f<0x141>(bd); oi = dpp_mov_i<0x141>(bi);
                if (lex_lt(od, oi, bd, bi)) { bd = od; bi = oi; }
                od = dpp_mov_f<0x140>(bd); oi = dpp_mov_i<0x140>(bi);
                if (lex_lt(od, oi, bd, bi)) { bd = od; bi = oi; }
                const int64_t row = rbase + j;
                if (r == 0 && row <= last_row && bi != 0x7fffffff) {
                    const float cur_d = p.best_d[row];
                    const int cur_i = p.best_i[row];
                    if (lex_lt(bd, bi, cur_d, cur_i < 0 ? 0x7fffffff : cur_i)) {
                        p.best_d[row] = bd;
                        p.best_i[row] = bi;
                    }
                }
            }
        }
    };
    // The ticket is taken with an opaque ds_add_rtn: hipcc orders a visible LDS atomic behind EVERY pending LDS-DMA
    // (s_waitcnt vmcnt(0): it cannot tell that the counter and the ring do not overlap), which would drain the
    // tile queue once per tile and leave a wave with one tile in flight instead of two.
    const unsigned ticket_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)lds_ticket;
    auto next_ticket = [&]() -> int {
        int tk = 0;
        if (lane == 0)
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(tk) : "v"(ticket_addr), "v"(1) : "memory");
        return tile0 + __builtin_amdgcn_readfirstlane(tk) * tile_step;
    };

    const bool exchange = MODE == kModeTopK && slots != nullptr;
    int t_cur = t_a, t_nxt = t_b, slot = 0;
    if ((MODE == kModeTopK || MODE == kModeFilter) && p.tau0) {  // bounds computed up front (launch_seed): stream from the first tile on
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            tq[h] = p.tau0[batch * kMaxBatch + h * 16 + r];
            tau[h] = tq[h];
        }
    }
    if (exchange) {
        // ---- warm-up: NKEEP tiles per wave whose distances are only kept (no top-k work yet) ----
        float wk[NKEEP][NRG][NQH][4];
        auto kill = [&](bool dead, float (&w)[NRG][NQH][4]) {
            if (dead) {
#pragma unroll
                for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                    for (int h = 0; h < NQH; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[rg][h][j] = VS_INF;
            }
        };
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // A landed (B follows)
        VS_STAMP(11);
        tile_distances(min(t_a, tlast), 0, wk[0], [] {});
        VS_STAMP(12);
        kill(t_a >= tile1, wk[0]);
        const int t_c = next_ticket();
        issue_or_stage(t_c, 0);  // queue: B C
        // publish this workgroup's per-query minimum (distinct workgroups hold distinct rows)
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            float m = VS_INF;
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg)
                m = fminf(m, fminf(fminf(wk[0][rg][h][0], wk[0][rg][h][1]), fminf(wk[0][rg][h][2], wk[0][rg][h][3])));
            m = fminf(m, __shfl_xor(m, 16));
            m = fminf(m, __shfl_xor(m, 32));
            if (g == 0) lds_wmin[wave * 32 + h * 16 + r] = m;
        }
        lds_barrier();
        if (tid >= 64 && tid < 64 + NQH * 16) {  // wave 1 publishes (one extra op in its queue)
            const int qx = tid - 64;
            float m = lds_wmin[qx];
#pragma unroll
            for (int w = 1; w < kScanWaves; ++w) m = fminf(m, lds_wmin[w * 32 + qx]);
            __hip_atomic_store(slots + qx * kSlotStride + blockIdx.x, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        VS_STAMP(2);
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // B landed (C, and on wave 1 the store, follow)
        tile_distances(min(t_b, tlast), 1, wk[1], [] {});
        kill(t_b >= tile1, wk[1]);
        VS_STAMP(7);
        // DPP row `g` of wave w will reduce query 4w+g: its 16 lanes read that query's 1 KB row of
        // minima (write-through-coherent sc1 loads, 64 contiguous bytes per lane = workgroups
        // 16 l .. 16 l + 15).  Issued now, consumed after the next tile.
        f32x4 v0, v1, v2, v3;
        {
            const float* s0 = slots + (4 * wave + g) * kSlotStride + 16 * r;
            asm volatile(
                "global_load_dwordx4 %0, %4, off sc1\n\t"
                "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                "global_load_dwordx4 %3, %4, off offset:48 sc1"
                : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3)
                : "v"(s0)
                : "memory");
        }
        const int t_d = next_ticket();
        issue_or_stage(t_d, 1);  // queue: C loads(4) D
        int t_e = t_d;
        if (NKEEP == 3) {
            asm volatile("s_waitcnt vmcnt(13)" ::: "memory");  // C landed
            VS_STAMP(9);
            tile_distances(min(t_c, tlast), 0, wk[NKEEP - 1], [] {});
            kill(t_c >= tile1, wk[NKEEP - 1]);
            VS_STAMP(8);
            t_e = next_ticket();
            issue_or_stage(t_e, 0);  // queue: loads(4) D E
            asm volatile("s_waitcnt vmcnt(18)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"memory");  // the minima are here
        } else {
            asm volatile("s_waitcnt vmcnt(9)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"memory");  // minima (and C) are here
        }
        VS_STAMP(10);
        // A workgroup that runs ahead of the others would find most slots still unpublished (+inf), i.e. a loose or
        // infinite bound, and then insert a large part of what it scans.  It re-reads its rows of minima until at
        // least half of the workgroups have published; every workgroup of the (resident, persistent) grid
        // publishes without waiting for anybody, so this cannot deadlock, and the spin is bounded anyway.
        {
            const int need = (int)gridDim.x / 2;
            for (int spin = 0;; ++spin) {
                int cf = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) cf += (v0[i] < VS_INF) + (v1[i] < VS_INF) + (v2[i] < VS_INF) + (v3[i] < VS_INF);
                cf += dpp_mov_i<0xB1>(cf);
                cf += dpp_mov_i<0x4E>(cf);
                cf += dpp_mov_i<0x141>(cf);
                cf += dpp_mov_i<0x140>(cf);  // row sum: workgroups that have published this row's query
                // (DPP rows of waves that hold no query -- 16-query launches use half of them -- have nothing to wait for)
                if (__all(4 * wave + g >= NQH * 16 || cf >= need) || spin >= 2048) break;
                __builtin_amdgcn_s_sleep(24);
                const float* s0 = slots + (4 * wave + g) * kSlotStride + 16 * r;
                asm volatile(
                    "global_load_dwordx4 %0, %4, off sc1\n\t"
                    "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                    "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                    "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                    : "v"(s0)
                    : "memory");
            }
        }
        {
            // Each lane folds its 16 workgroups into one minimum; the k1-th smallest of the row's
            // 16 lane minima is still backed by k1 distinct rows (one per lane group), and with
            // groups this large it is within a few per cent of the k1-th smallest of all 256.
            float m = fminf(fminf(fminf(v0[0], v0[1]), fminf(v0[2], v0[3])), fminf(fminf(v1[0], v1[1]), fminf(v1[2], v1[3])));
            m = fminf(m, fminf(fminf(fminf(v2[0], v2[1]), fminf(v2[2], v2[3])), fminf(fminf(v3[0], v3[1]), fminf(v3[2], v3[3]))));
            float kth = VS_INF;
            for (int round = 0; round < p.k1; ++round) {
                float x = m;
                x = fminf(x, dpp_mov_f<0xB1>(x));
                x = fminf(x, dpp_mov_f<0x4E>(x));
                x = fminf(x, dpp_mov_f<0x141>(x));
                x = fminf(x, dpp_mov_f<0x140>(x));  // row minimum in every lane of the row
                kth = x;
                const unsigned rowmask = (unsigned)((__ballot(m == x) >> (16 * g)) & 0xFFFFull);
                if (rowmask != 0u && r == __builtin_ctz(rowmask)) m = VS_INF;  // drop exactly one instance
            }
            if (r == 0) lds_tau[4 * wave + g] = kth < VS_INF ? next_up(kth) : VS_INF;
        }
        lds_barrier();
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            tq[h] = lds_tau[h * 16 + r];
            tau[h] = tq[h];
        }
        VS_STAMP(3);
        VS_STAMPC(13);
        // replay the kept tiles against the bound (almost nothing passes)
        consume(min(t_a, tlast), wk[0]);
        consume(min(t_b, tlast), wk[1]);
        if (NKEEP == 3) {
            consume(min(t_c, tlast), wk[NKEEP - 1]);
            t_cur = t_d;  // slot 1
            t_nxt = t_e;  // slot 0
            slot = 1;
        } else {
            t_cur = t_c;  // slot 0
            t_nxt = t_d;  // slot 1
            slot = 0;
        }
    }
    // ---- steady state: t_cur sits in `slot` (landed or landing), t_nxt in the other slot ----
    // (written per slot so that the slot is a compile-time constant: LDS offsets become immediates)
    auto step = [&](const int sl) __attribute__((always_inline)) {
        const int t_new = next_ticket();
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        float d[NRG][NQH][4];
        // the slot is refilled as soon as its fragments sit in registers, before the MFMAs: two tiles in flight
        tile_distances(t_cur, sl, d, [&] {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_or_stage(t_new, sl);
        });
        consume(t_cur, d);
        t_cur = t_nxt;
        t_nxt = t_new;
    };
    if (slot == 1 && t_cur < tile1) step(1);
    while (t_cur < tile1) {
        step(0);
        if (t_cur >= tile1) break;
        step(1);
    }
    VS_STAMP(5);
    VS_STAMPC(14);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the discarded tail prefetches: LDS is reused below
    if (MODE == kModeStore || MODE == kModeFilter) return;
    if (MODE == kModeAssign) {
        __syncthreads();  // LDS is reused by the next centroid block
        continue;
    }
    __syncthreads();
    VS_STAMP(4);

    // ---- workgroup merge: compact the entries that can still matter (d < tau0), rank them ----
    // With a bound in force a query keeps a handful of entries per workgroup: they fit a small buffer behind the
    // ring (kMergeSmall per query), which leaves the ring -- and the next batch's two staged tiles per wave --
    // alone.  Only an unbounded scan (small shards: lists full of unfiltered entries) needs the big buffers; they
    // overlay the ring, so the staged tiles are then fetched again.
    constexpr int NQ = NQH * 16;
    constexpr int CAP = 32 * KCAP;  // 32 lane lists per query: cannot overflow
    float* small_d = reinterpret_cast<float*>(smem + kRingBytes + kQStageBytes + kScratchBytes);
    int* small_i = reinterpret_cast<int*>(small_d + kMaxBatch * kMergeSmall);
    auto compact = [&](float* cand_d, int* cand_i, const int cap) {
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            const int qidx = h * 16 + r;
#pragma unroll
            for (int j = 0; j < KCAP; ++j)
                if (li[h][j] >= 0 && ld[h][j] < tq[h]) {
                    const int pos = atomicAdd(&lds_cnt[qidx], 1);
                    if (pos < cap) {
                        cand_d[qidx * cap + pos] = ld[h][j];
                        cand_i[qidx * cap + pos] = li[h][j];
                    }
                }
        }
    };
    auto rank = [&](const float* cand_d, const int* cand_i, const int cap, auto epl_tag) {
        constexpr int EPL = decltype(epl_tag)::value;
        for (int qq = wave; qq < NQ; qq += kScanWaves) {
            const int M = lds_cnt[qq];
            float cd[EPL];
            int ci[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int idx = e * 64 + lane;
                cd[e] = idx < M ? cand_d[qq * cap + idx] : VS_INF;
                ci[e] = idx < M ? cand_i[qq * cap + idx] : 0x7fffffff;
            }
            // partial lists are query-major: [batch][query][workgroup][KCAP] (one merge launch ranks all batches)
            float* od = p.part_d + (((int64_t)batch * kMaxBatch + qq) * kSlotStride + blockIdx.x) * KCAP;
            int32_t* oi = p.part_i + (((int64_t)batch * kMaxBatch + qq) * kSlotStride + blockIdx.x) * KCAP;
            const int rounds = min(min(p.k1, KCAP), M);
            for (int round = 0; round < rounds; ++round) {
                float md = cd[0];
                int mi = ci[0];
#pragma unroll
                for (int e = 1; e < EPL; ++e)
                    if (lex_lt(cd[e], ci[e], md, mi)) {
                        md = cd[e];
                        mi = ci[e];
                    }
                float bd;
                int bi;
                wave_lexmin(md, mi, bd, bi);
                if (lane == 0) {
                    od[round] = bd;
                    oi[round] = bi;
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e)
                    if (ci[e] == bi && cd[e] == bd) {
                        cd[e] = VS_INF;
                        ci[e] = 0x7fffffff;
                    }
            }
            if (lane < KCAP && lane >= rounds) {
                od[lane] = VS_INF;
                oi[lane] = -1;
            }
        }
    };
    compact(small_d, small_i, kMergeSmall);
    const bool too_many = __syncthreads_or(lds_cnt[tid & 31] > kMergeSmall);
    if (!too_many) {
        rank(small_d, small_i, kMergeSmall, std::integral_constant<int, 1>{});
    } else {
        if (tid < 32) lds_cnt[tid] = 0;
        __syncthreads();
        float* big_d = reinterpret_cast<float*>(smem);
        int* big_i = reinterpret_cast<int*>(smem + (size_t)NQ * CAP * sizeof(float));
        compact(big_d, big_i, CAP);
        __syncthreads();
        rank(big_d, big_i, CAP, std::integral_constant<int, CAP / 64>{});
        tiles_staged = false;  // the ring was overwritten
    }
    __syncthreads();  // LDS (ring, counters, ticket, query stage) is reused by the next batch
    }  // batch loop
    VS_STAMP(6);
}


// ------------------------------------------------------------------------------------------------
// Wide exact-int8 scan (see WideParams).  The data path is the PREC = 1 path of scan_kernel: per-wave ring of two
// 64-row tile slots filled by LDS-DMA, XOR-swizzled so that the A fragments read conflict free, hand-counted vmcnt.
// What differs: NQH query column blocks per pass (the B operands come straight from global memory, prepared by
// seed_qnorm_kernel), the bound is fixed (launch_seed), survivors go to global candidate lists, and the tile ticket
// runs over all passes of the launch (ticket = pass * T + n), so a wave slides from one pass into the next without
// meeting anybody: the kernel has one barrier (ticket initialisation).
// ------------------------------------------------------------------------------------------------
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

constexpr int kWideLds = kRingBytes + 64;

template <int NQH>
__global__ __launch_bounds__(kScanThreads, NQH <= 8 ? 2 : 1) void scan_i8w_kernel(const WideParams p) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr int TR = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lds_ticket = reinterpret_cast<int*>(smem + kRingBytes);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int tiles_total = (int)((p.n_rows + TR - 1) / TR);
    const int G = (int)gridDim.x;
    const int T = (tiles_total - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup per pass (grid <= tiles_total)
    const int NB = NQH / p.bpb;                                  // batches per pass
    const int n_pass = (p.n_batches + NB - 1) / NB;
    (void)lds_ticket;
    // a wave takes whole passes, and a contiguous range of the tiles of the passes that do not deal evenly to the 8 waves:
    // see scan_f32s_kernel.  No shared ticket, no barrier.
    const int n_whole = n_pass & ~(kScanWaves - 1), n_rest = n_pass - n_whole;
    char* ring = smem + wave * (kDepth * kSlotBytes);
    unsigned voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row_in = 8 * j + (lane >> 3);
        voff[j] = (unsigned)(row_in * 128 + 16 * ((lane & 7) ^ ((row_in >> 1) & 7)));
    }
    const unsigned voff_n = (unsigned)lane * 4u;
    // LDS-DMA written as instructions (see scan_f32s_kernel): no vector instruction per piece
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ring);
    auto issue_tile = [&](int tile, int slot) __attribute__((always_inline)) {
        const int64_t row0 = (int64_t)tile * TR;
        const unsigned dst = ring_lds + (unsigned)(slot * kSlotBytes);
        const char* tb = reinterpret_cast<const char*>(p.base_u8) + row0 * kDim;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff[j]), "s"(tb), "n"(j * 1024) : "memory", "scc");
        const char* nb = reinterpret_cast<const char*>(p.rterm + row0);
        asm volatile("s_add_u32 m0, %0, 8192\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(dst), "v"(voff_n), "s"(nb) : "memory", "scc");
    };
    unsigned fa[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int row_in = 16 * (c >> 1) + r;
        fa[c] = (unsigned)(wave * (kDepth * kSlotBytes) + row_in * 128 + ((((c & 1) * 4 + g) ^ ((row_in >> 1) & 7)) << 4));
    }
    const unsigned fa_n = (unsigned)(wave * (kDepth * kSlotBytes) + 8192 + 16 * g);
    int it_u = wave - kScanWaves, it_n = 0, it_end = 0, it_pass = 0;
    int rem_pos = (int)((long long)wave * n_rest * T / kScanWaves), rem_end = (int)((long long)(wave + 1) * n_rest * T / kScanWaves);
    auto next_tile = [&](int& pass_out) __attribute__((always_inline)) -> int {
        while (it_n >= it_end) {
            if (it_u + kScanWaves < n_whole) {          // the next whole pass of this wave
                it_u += kScanWaves;
                it_pass = it_u, it_n = 0, it_end = T;
            } else if (rem_pos < rem_end) {             // its range of the remaining passes' tiles: at most two passes
                const int pr = rem_pos / T;
                it_pass = n_whole + pr;
                it_n = rem_pos - pr * T;
                it_end = min(T, it_n + (rem_end - rem_pos));
                rem_pos += it_end - it_n;
            } else {
                pass_out = n_pass;
                return (int)blockIdx.x;  // past the end: the DMA still goes out (queue accounting), to a tile nobody uses
            }
        }
        pass_out = it_pass;
        return (int)blockIdx.x + (it_n++) * G;
    };

    // per-pass state: B operands of the NQH column blocks, the queries' constant terms and integer bounds.
    // d = qt + rt - 2 acc < tau  <=>  2 acc - rt > qt - tau =: thr  (the hot loop never forms d)
    i32x4 qb[NQH][2];
    int qt[NQH], thr[NQH];  // (thr >> 1 and the query's global index are formed where they are used: registers)
    auto load_pass = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            const int batch = pass * NB + h / p.bpb;
            const int qrow = 16 * (h % p.bpb) + r;
            const bool live = batch < p.n_batches && qrow < p.nq_valid;
            const int bq = live ? batch * kMaxBatch + qrow : 0;
            // fragment order (launch_seed): 1 KB per instruction in one piece (a dead block reads batch 0's: masked by thr)
            const int8_t* src = p.q8frag + (((int64_t)(live ? batch : 0) * 2 + (h % p.bpb)) * 2 * 64 + lane) * 16;
            qb[h][0] = *reinterpret_cast<const i32x4*>(src);
            qb[h][1] = *reinterpret_cast<const i32x4*>(src + 64 * 16);
            qt[h] = p.qterm[bq];
            const float t0 = p.tau0[bq];
            const bool dead = !live || p.invalid[live ? batch : 0] != 0;
            // d < tau0 for integer d  <=>  d < ceil(tau0)  (tau0 is next_up of an integer-valued float, or +inf; distances
            // are below 2^24, so any bound from 2^26 on admits everything)
            const int ti = (int)ceilf(fminf(fmaxf(t0, -67108864.f), 67108864.f));
            thr[h] = dead ? 0x7fffffff : qt[h] - ti;  // (the hot loop compares with thr >> 1: floor, an odd thr is lowered by one)
        }
    };

    int4* wbuf = p.sink.wbuf + ((int64_t)blockIdx.x * kScanWaves + wave) * p.sink.wcap;
    int wbase = 0;  // wave-uniform fill of the private candidate buffer
    int pass_cur, pass_nxt;
    int tile_cur = next_tile(pass_cur);
    int tile_nxt = next_tile(pass_nxt);
    issue_tile(tile_cur, 0);
    issue_tile(tile_nxt, 1);
    int have_pass = -1;

    auto step = [&](const int sl) __attribute__((always_inline)) {
        if (pass_cur != have_pass) {  // wave-uniform: this wave enters a new pass (four batches: the drain is amortised)
            load_pass(pass_cur);
            have_pass = pass_cur;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // operands are here (and so are both staged tiles)
        }
        int pass_new;
        const int tile_new = next_tile(pass_new);
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        const char* src = smem + sl * kSlotBytes;
        i32x4 a0[4], a1[4], rtv[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            a0[rg] = *reinterpret_cast<const i32x4*>(src + fa[2 * rg]);
            a1[rg] = *reinterpret_cast<const i32x4*>(src + fa[2 * rg + 1]);
            rtv[rg] = *reinterpret_cast<const i32x4*>(src + fa_n + 64 * rg);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_tile(tile_new, sl);  // the slot is refilled as soon as its fragments sit in registers
        const int row_t = tile_cur * TR + 4 * g;
        // hot loop, branch free: d < tau  <=>  2 dot - rt > thr.  With rt = 2 rh + ro (ro = 0 / 1) that is
        // 2 (dot - rh) - ro > thr, which for an EVEN thr means dot - rh > thr / 2 whatever ro is -- and an odd thr may be
        // lowered by one here, because a block that passes is recomputed and tested exactly below.  -rh goes in as the C
        // operand of the first MFMA, so the accumulators come out as dot - rh and the only vector work per column block
        // is the maximum of its 16 values.
        i32x4 nrh[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) nrh[rg] = -(rtv[rg] >> 1);
        unsigned hit = 0;
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            i32x4 acc[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                acc[rg] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[rg], qb[h][0], nrh[rg], 0, 0, 0);
                acc[rg] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[rg], qb[h][1], acc[rg], 0, 0, 0);
            }
            int emax = max(max(acc[0][0], acc[0][1]), max(acc[0][2], acc[0][3]));
#pragma unroll
            for (int rg = 1; rg < 4; ++rg) emax = max(max(emax, acc[rg][0]), max(max(acc[rg][1], acc[rg][2]), acc[rg][3]));
            hit |= emax > (thr[h] >> 1) ? (1u << h) : 0u;
        }
        // wave-uniform union of the hit bits (DPP or-reduction)
        unsigned um = hit;
        um |= (unsigned)dpp_mov_i<0xB1>((int)um);
        um |= (unsigned)dpp_mov_i<0x4E>((int)um);
        um |= (unsigned)dpp_mov_i<0x141>((int)um);
        um |= (unsigned)dpp_mov_i<0x140>((int)um);
        um = (unsigned)(__builtin_amdgcn_readlane((int)um, 0) | __builtin_amdgcn_readlane((int)um, 16) |
                        __builtin_amdgcn_readlane((int)um, 32) | __builtin_amdgcn_readlane((int)um, 48));
        if (um) {
#pragma unroll
            for (int h = 0; h < NQH; ++h) {
                if (!(um & (1u << h))) continue;  // scalar branch
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    i32x4 acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[rg], qb[h][0], (i32x4){0, 0, 0, 0}, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[rg], qb[h][1], acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = row_t + 16 * rg + j;
                        const bool pass = 2 * acc[j] - rtv[rg][j] > thr[h] && row < (int)p.n_rows;
                        const unsigned long long mask = __ballot(pass);
                        if (mask) {  // wave-uniform
                            // This wave's private candidate buffer, positions from the ballot: plain stores, no atomics -- a
                            // returning atomic would have to be waited for with vmcnt(0), i.e. drain the tile queue per hit.
                            const int pos = wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                            if (pass && pos < p.sink.wcap) {
                                // the integer the fp32 path computes exactly: ||q||^2 + ||b||^2 - 2 q.b
                                const int d = qt[h] + rtv[rg][j] - 2 * acc[j];
                                // (a block that hits is live: its query is batch pass * NB + h / bpb, row 16 (h % bpb) + r)
                                wbuf[pos] = make_int4((pass_cur * NB + h / p.bpb) * kMaxBatch + 16 * (h % p.bpb) + r, __builtin_bit_cast(int, (float)d), row + p.id_offset, 0);
                            }
                            wbase += __popcll(mask);
                        }
                    }
                }
            }
        }
        tile_cur = tile_nxt;
        pass_cur = pass_nxt;
        tile_nxt = tile_new;
        pass_nxt = pass_new;
    };
    while (pass_cur < n_pass) {
        step(0);
        if (pass_cur >= n_pass) break;
        step(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the tail prefetches before the wave ends
    sink_bin_wave(p.sink, (int)blockIdx.x * kScanWaves + wave, wbase, lane);  // no separate binning launch
}

// ------------------------------------------------------------------------------------------------
// Streaming fp32 scan (see StreamParams): scan_kernel's fp32 data path and arithmetic, the wide int8 scan's organisation.
// ------------------------------------------------------------------------------------------------
// NB = batches per pass over the rows.  NB = 1: one batch per pass, HBM bound (516 MB per batch).  NB = 2: two batches
// (four 16-query column blocks) share a pass; a tile is then 128 MFMAs for its 8 KB and the kernel is MFMA bound --
// the same FMA chain per (row, query), so the same bits.  The B operands of four column blocks are 128 registers:
// NB = 2 runs at two waves per SIMD with up to 256 registers each.
template <int NB>
__global__ __launch_bounds__(kScanThreads, NB == 1 ? 2 : 1) void scan_f32s_kernel(const StreamParams p) {
    constexpr int NH = 2 * NB;  // 16-query column blocks per pass
    constexpr int TR = kTileRows;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lds_ticket = reinterpret_cast<int*>(smem + kRingBytes);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int tiles_total = (int)((p.n_rows + TR - 1) / TR);
    const int G = (int)gridDim.x;
    const int T = (tiles_total - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup per batch (grid <= tiles_total)
    const int n_pass = (p.n_batches + NB - 1) / NB;
    (void)lds_ticket;
    // Work of a wave = whole passes, not tiles dealt one by one: a pass costs its operand fetch and a drain on entry (1.3 us
    // with one batch per pass, 6.6 us with two), so a wave should enter as few passes as possible.  The workgroup's passes
    // are dealt whole as far as they deal evenly to the 8 waves (the first n_pass & ~7); the tiles of the n_rest others,
    // pass after pass, are shared out as 8 equal contiguous ranges (a range is at most one pass long, so it touches at most
    // two passes): 16 passes -> a wave takes 2 whole passes instead of entering all 16; 10 passes -> one whole pass and a
    // quarter of another; 5 passes -> 5/8 of a pass in at most two entries instead of an eighth of each of the five.
    // No shared ticket, no barrier: waves never meet.
    const int n_whole = n_pass & ~(kScanWaves - 1), n_rest = n_pass - n_whole;  // passes taken whole / shared out by range
    char* ring = smem + wave * (kDepth * kSlotBytes);
    unsigned voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row_in = 2 * j + (lane >> 5);
        voff[j] = (unsigned)(row_in * 512 + 16 * ((lane & 31) ^ row_in));
    }
    const unsigned voff_n = (unsigned)lane * 4u;
    // LDS-DMA written as instructions: "scalar tile base + this lane's 32-bit offset", M0 = the slot's LDS address + the
    // piece (one wait state between the scalar write of M0 and the instruction that reads it).  (Through the builtin every piece cost two vector instructions -- a register copy and the M0 value read
    // back from a spilled scalar -- and a vector instruction costs this kernel about 8 cycles of MFMA pipe.)
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ring);
    auto issue_tile = [&](int tile, int slot) __attribute__((always_inline)) {
        const int64_t row0 = (int64_t)tile * TR;
        const unsigned dst = ring_lds + (unsigned)(slot * kSlotBytes);
        const char* tb = reinterpret_cast<const char*>(p.base) + row0 * (kDim * 4);
        static_assert(VS_ROW_CPOL == 2, "the row pieces are issued with the nt policy");
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(dst), "v"(voff[j]), "s"(tb), "n"(j * 1024) : "memory", "scc");
        const char* nb = reinterpret_cast<const char*>(p.bnorm + row0);
        asm volatile("s_add_u32 m0, %0, 8192\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(dst), "v"(voff_n), "s"(nb) : "memory", "scc");
    };
    unsigned fa[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) fa[c] = (unsigned)(wave * (kDepth * kSlotBytes) + r * 512 + (((4 * c + g) ^ r) << 4));
    const unsigned fa_n = (unsigned)(wave * (kDepth * kSlotBytes) + 8192 + 16 * g);
    // this wave's position: whole pass it_u, then [rem_pos, rem_end) of the remaining passes' tiles; inside a pass tile
    // index it_n of the workgroup's T, up to it_end
    int it_u = wave - kScanWaves, it_n = 0, it_end = 0, it_pass = 0;
    int rem_pos = (int)((long long)wave * n_rest * T / kScanWaves), rem_end = (int)((long long)(wave + 1) * n_rest * T / kScanWaves);
    auto next_tile = [&](int& pass_out) __attribute__((always_inline)) -> int {
        while (it_n >= it_end) {
            if (it_u + kScanWaves < n_whole) {          // the next whole pass of this wave
                it_u += kScanWaves;
                it_pass = it_u, it_n = 0, it_end = T;
            } else if (rem_pos < rem_end) {             // its range of the remaining passes' tiles: at most two passes
                const int pr = rem_pos / T;
                it_pass = n_whole + pr;
                it_n = rem_pos - pr * T;
                it_end = min(T, it_n + (rem_end - rem_pos));
                rem_pos += it_end - it_n;
            } else {
                pass_out = n_pass;
                return (int)blockIdx.x;  // past the end: the DMA still goes out (queue accounting), to a tile nobody uses
            }
        }
        pass_out = it_pass;
        return (int)blockIdx.x + (it_n++) * G;
    };

    // per-batch state: the 32 queries as B operands (qf[h][c][i] = Q[16 h + r][16 c + 4 g + i]), their norms and bounds.
    // The loads are inline asm on purpose: the compiler does not know them as memory operations, so it puts no
    // s_waitcnt of its own in front of their first use (it would be vmcnt(0): a drain of the tile queue in every
    // step); the hand-counted waits of the tile loop cover them.  A padding query (main.cpp:206-211) reads row 0 and
    // is masked where candidates are taken: its MFMA column influences nothing else.
    f32x4 qf[NH][8];
    float qn[NH], tau[NH], thr[NH];
    int qglob[NH];
    bool live[NH];
    auto load_pass = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int batch = min(pass * NB + h / 2, p.n_batches - 1);  // (a pass of the last, odd batch: its second half is dead)
            const int qrow = 16 * (h & 1) + r;
            live[h] = qrow < p.nq_valid && pass * NB + h / 2 < p.n_batches;
            qglob[h] = batch * kMaxBatch + (live[h] ? qrow : 0);
            // fragment order (launch_seed): 1 KB per instruction in one piece
            const float* src = p.qfrag + (((int64_t)batch * 2 + (h & 1)) * 8 * 64 + lane) * 4;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* pc = src + c * 64 * 4;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[h][c]) : "v"(pc) : "memory");
            }
            const float* pn = p.qnorm + qglob[h];
            const float* pt = p.tau0 + qglob[h];
            asm volatile("global_load_dword %0, %1, off" : "=v"(qn[h]) : "v"(pn) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(tau[h]) : "v"(pt) : "memory");
        }
    };

    int4* wbuf = p.sink.wbuf + ((int64_t)blockIdx.x * kScanWaves + wave) * p.sink.wcap;
    int wbase = 0;  // wave-uniform fill of the private candidate buffer
    int pass_cur, pass_nxt;
    int tile_cur = next_tile(pass_cur);
    int tile_nxt = next_tile(pass_nxt);
    issue_tile(tile_cur, 0);
    issue_tile(tile_nxt, 1);
    int have_pass = -1;

    auto step = [&](const int sl) __attribute__((always_inline)) {
        if (pass_cur != have_pass) {  // wave-uniform: this wave enters the next batch
            // (fetching the operands behind the previous batch's last tile instead, ahead of the refill, saves the drain
            //  but delays that refill by the tile's arithmetic: measured slower, 12.3 vs 11.6 us per batch on a 125 K-row shard)
            load_pass(pass_cur);
            have_pass = pass_cur;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // operands are here (and so are both staged tiles)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                // bound of the hot test.  L2: d = RN(RN(qn + bn) - 2 dot) < tau implies RN(bn - 2 dot) < tau - qn + slack:
                // the three roundings together move the comparison by less than 2^-24 * 8 (qn + |tau|) (a row under the
                // bound has bn < 2 (qn + tau)); the slack is 16 times that.  IP: -dot < tau <=> dot > -tau, exactly.
                const float l2thr = (tau[h] - qn[h]) + 9.5367431640625e-7f * (qn[h] + fabsf(tau[h]));
                thr[h] = p.metric ? (live[h] ? -tau[h] : __builtin_inff()) : (live[h] ? l2thr : -__builtin_inff());
            }
        }
        int pass_new;
        const int tile_new = next_tile(pass_new);
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        const char* src = smem + sl * kSlotBytes;
        f32x4 a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = *reinterpret_cast<const f32x4*>(src + fa[c]);
        const f32x4 bn = *reinterpret_cast<const f32x4*>(src + fa_n);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_tile(tile_new, sl);  // the slot is refilled as soon as its fragments sit in registers
        f32x4 acc[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (NB == 1 || (pass_cur + 1) * NB <= p.n_batches) {  // wave-uniform
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < NH; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[h][c][i], acc[h], 0, 0, 0);
        } else {  // the pass of a last, odd batch: its second half is dead, so are its MFMAs
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[h][c][i], acc[h], 0, 0, 0);
        }
        // Hot path: ONE fma and ONE compare per value, the verdicts collected as wave masks in scalar registers (every
        // vector instruction here costs the SIMD about 8 cycles of its MFMA pipe).  The test is a superset of d < tau
        // (thr carries the rounding slack, load_pass); whatever passes it is judged again below with the exact expression.
        unsigned long long hit[NH][4], hits = 0;
        if (!p.metric) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hit[h][j] = __ballot(fmaf(-2.0f, acc[h][j], bn[j]) < thr[h]);
                    hits |= hit[h][j];
                }
        } else {
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hit[h][j] = __ballot(acc[h][j] > thr[h]);  // -acc < tau, exactly
                    hits |= hit[h][j];
                }
        }
        if (hits) {  // rare: a few hundred rows per query per million
            const int row_t = tile_cur * TR + 4 * g;
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!hit[h][j]) continue;  // wave-uniform
                    // cpu_baseline.cpp:241  dist = qn + bn - 2*dot  (gcc contracts to fnmadd(2, dot, qn+bn))
                    const float l2 = fmaf(-2.0f, acc[h][j], qn[h] + bn[j]);
                    const float d = p.metric ? -acc[h][j] : l2;
                    const int row = row_t + j;
                    const bool pass = live[h] && d < tau[h] && row < (int)p.n_rows;
                    const unsigned long long mask = __ballot(pass);
                    if (mask) {
                        const int pos = wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                        if (pass && pos < p.sink.wcap)
                            wbuf[pos] = make_int4(qglob[h], __builtin_bit_cast(int, d), row + p.id_offset, 0);
                        wbase += __popcll(mask);
                    }
                }
        }
        tile_cur = tile_nxt;
        pass_cur = pass_nxt;
        tile_nxt = tile_new;
        pass_nxt = pass_new;
    };
    while (pass_cur < n_pass) {
        step(0);
        if (pass_cur >= n_pass) break;
        step(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the tail prefetches before the wave ends
    sink_bin_wave(p.sink, (int)blockIdx.x * kScanWaves + wave, wbase, lane);  // no separate binning launch
}

hipError_t launch_scan_f32_stream(const StreamParams& p, int grid, hipStream_t s) {
    static bool attr_set[64][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    const int v = p.batches_per_pass == 2 ? 1 : 0;
    const void* fn = v ? reinterpret_cast<const void*>(scan_f32s_kernel<2>) : reinterpret_cast<const void*>(scan_f32s_kernel<1>);
    if (!attr_set[dev][v]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds);
        if (e != hipSuccess) return e;
        attr_set[dev][v] = true;
    }
    if (v) hipLaunchKernelGGL(scan_f32s_kernel<2>, dim3(grid), dim3(kScanThreads), kWideLds, s, p);
    else hipLaunchKernelGGL(scan_f32s_kernel<1>, dim3(grid), dim3(kScanThreads), kWideLds, s, p);
    return hipGetLastError();
}

hipError_t launch_scan_i8_wide(const WideParams& p, int grid, int nqh, hipStream_t s) {
    static bool attr_set[64][3] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    auto k4 = scan_i8w_kernel<4>;
    auto k8 = scan_i8w_kernel<8>;
    auto k12 = scan_i8w_kernel<12>;
    const int which = nqh == 12 ? 2 : nqh == 8 ? 1 : 0;
    if (nqh != 4 && nqh != 8 && nqh != 12) return hipErrorInvalidValue;
    const void* fn = which == 2 ? reinterpret_cast<const void*>(k12) : which ? reinterpret_cast<const void*>(k8) : reinterpret_cast<const void*>(k4);
    if (!attr_set[dev][which]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds);
        if (e != hipSuccess) return e;
        attr_set[dev][which] = true;
    }
    if (which == 2) hipLaunchKernelGGL(k12, dim3(grid), dim3(kScanThreads), kWideLds, s, p);
    else if (which) hipLaunchKernelGGL(k8, dim3(grid), dim3(kScanThreads), kWideLds, s, p);
    else hipLaunchKernelGGL(k4, dim3(grid), dim3(kScanThreads), kWideLds, s, p);
    return hipGetLastError();
}

template <int NQH, int KCAP, int MODE, int PREC = 0>
static hipError_t launch_scan_t(const ScanParams& p, int grid, hipStream_t s) {
    auto kfn = scan_kernel<NQH, KCAP, MODE, PREC>;
    static bool attr_set[64] = {};  // per device: the attribute belongs to the device's copy of the code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kScanLds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(kScanThreads), kScanLds, s, p);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Seed bounds (see SeedParams).  seed_qnorm_kernel: ||q||^2 in the reference's order (+ the queries as bytes for the
// int8 paths); seed_kernel: minima of 64 groups of 32 sample tiles per (batch, query); seed_tau_kernel: k1-th smallest
// of a query's 64 group minima.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seed_qnorm_kernel(const SeedParams p) {
    const int batch = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int row = tid >> 3, j = tid & 7;
    const float* qb = p.q + (int64_t)batch * p.q_batch_stride;
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
    if (p.qfrag) {  // B-fragment order for the fp32 streaming scan (see SeedParams)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + 256 * u;  // (h, c, lane)
            const int fl = idx & 63, c = (idx >> 6) & 7, hh = idx >> 9;
            const int qrow = 16 * hh + (fl & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (qrow < p.nq_valid) v = *reinterpret_cast<const f32x4*>(qb + qrow * kDim + 16 * c + 4 * (fl >> 4));
            *reinterpret_cast<f32x4*>(p.qfrag + ((int64_t)batch * 1024 + idx) * 4) = v;
        }
    }
    if (p.q8) {
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        if (j == 0) p.qterm[batch * kMaxBatch + row] = (int)sum - 256 * part - 4194304;
        if (!q_ok) p.invalid[batch] = 1;  // same value from every thread that sees a bad element
    }
    if (p.q8frag) {  // the byte queries in B-fragment order (see SeedParams): thread = (h, half, lane), 16 bytes each
        const int fl = tid & 63, half = (tid >> 6) & 1, hh = tid >> 7;
        const int qrow = 16 * hh + (fl & 15);
        int w[4] = {0, 0, 0, 0};
        if (qrow < p.nq_valid) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(qb + qrow * kDim + 64 * half + 16 * (fl >> 4) + 4 * v);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[v] |= (((int)x[e] - 128) & 0xff) << (8 * e);
            }
        }
        *reinterpret_cast<int4*>(p.q8frag + ((int64_t)batch * 256 + tid) * 16) = make_int4(w[0], w[1], w[2], w[3]);
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
}

__global__ __launch_bounds__(1024) void seed_tau_kernel(const SeedParams p) {
    // one wave per query (two queries per wave): lane l holds group minimum l; k1 rounds of wave minimum
    const int batch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* src = p.wmin + (int64_t)batch * kSeedChunks * kMaxBatch;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int q = 2 * wave + qq;
        float v = src[lane * kMaxBatch + q];
        float kth = VS_INF;
        for (int round = 0; round < p.k1; ++round) {
            kth = wave_min_f32(v);
            const unsigned long long msk = __ballot(v == kth);
            if (msk != 0ull && lane == __builtin_ctzll(msk)) v = VS_INF;  // drop exactly one instance
        }
        if (lane == 0) p.tau0[batch * kMaxBatch + q] = kth < VS_INF ? next_up(kth) : VS_INF;
    }
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
    hipLaunchKernelGGL(seed_qnorm_kernel, dim3(p.n_batches), dim3(256), 0, s, p);
    const int wgs = p.n_batches * kSeedChunks;  // one workgroup per (batch, group of sample tiles)
    hipLaunchKernelGGL(seed_kernel, dim3(wgs), dim3(256), 0, s, p);
    hipLaunchKernelGGL(seed_tau_kernel, dim3(p.n_batches), dim3(1024), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanParams& p, int grid, int kcap, int nqh, int mode, hipStream_t s) {
    if (mode == kModeTopK && p.base_u8) {  // int8 data path
        if (kcap == 8) return nqh == 1 ? launch_scan_t<1, 8, kModeTopK, 1>(p, grid, s) : launch_scan_t<2, 8, kModeTopK, 1>(p, grid, s);
        if (kcap == 16) return nqh == 1 ? launch_scan_t<1, 16, kModeTopK, 1>(p, grid, s) : launch_scan_t<2, 16, kModeTopK, 1>(p, grid, s);
        return hipErrorInvalidValue;
    }
    if (mode == kModeAssign) return launch_scan_t<2, 8, kModeAssign>(p, grid, s);
    if (mode == kModeFilter) return launch_scan_t<2, 8, kModeFilter>(p, grid, s);
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
            if (p.flag_empty && !(outd[0] < VS_INF)) f = 2;
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
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Compacting merge (fast path, G*kin <= kCompactCap).  With the threshold exchange most partial
// lists are empty, so the finite entries are first compacted into LDS (one atomic append each);
// a handful of candidates is then ranked by a single wave with DPP/shuffle argmin rounds and no
// barriers.  Larger candidate sets use workgroup-wide rounds over the LDS array.
// ------------------------------------------------------------------------------------------------
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
__device__ __forceinline__ void merge_compact_body(const MergeParams& p, const MergeLayout& L, float* const cd, int* const ci) {
    __shared__ int cnt;
    __shared__ float wbd[4];
    __shared__ int wbi[4];
    __shared__ int wbp[4];
    __shared__ float outd[kMergeTrack];
    const int q = blockIdx.x;  // output query; input lists may be grouped in padded batches
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
            p.flags[q] = f;
        }
        if (p.tau_out) {
            const float kth = outd[n_track - 1];
            p.tau_out[q] = kth < VS_INF ? next_up(kth) : VS_INF;
        }
    }
}

__global__ __launch_bounds__(256) void merge_compact_kernel(const MergeParams p, const MergeLayout L) {
    __shared__ float cd[kCompactCap];
    __shared__ int ci[kCompactCap];
    merge_compact_body(p, L, cd, ci);
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

// ------------------------------------------------------------------------------------------------
// k-means update (index builder, create_ivf_model_reordered.py:96-105): cluster sums are accumulated
// in 44.20 fixed point with 64-bit integer atomics, so the result does not depend on the order in
// which rows arrive (float atomics would make the index differ from run to run).
// ------------------------------------------------------------------------------------------------
constexpr double kFix = 1048576.0;  // 2^20

__global__ __launch_bounds__(256) void kmeans_accum_kernel(const float* __restrict__ x, const int32_t* __restrict__ assign,
                                                           int64_t rows, unsigned long long* __restrict__ acc,
                                                           int32_t* __restrict__ counts) {
    // 128 threads per row, 2 rows per workgroup pass
    const int t = threadIdx.x & 127;
    for (int64_t row = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 7); row < rows; row += (int64_t)gridDim.x * 2) {
        const int c = assign[row];
        if (c < 0) continue;
        const long long v = __double2ll_rn((double)x[row * kDim + t] * kFix);
        atomicAdd(acc + (int64_t)c * kDim + t, (unsigned long long)v);
        if (t == 0) atomicAdd(counts + c, 1);
    }
}

__global__ __launch_bounds__(128) void kmeans_finalize_kernel(float* __restrict__ cents, const unsigned long long* __restrict__ acc,
                                                              const int32_t* __restrict__ counts, double* __restrict__ shift) {
    const int c = blockIdx.x, t = threadIdx.x;
    const int n = counts[c];
    float delta2 = 0.f;
    if (n > 0) {
        const double mean = (double)(long long)acc[(int64_t)c * kDim + t] / kFix / (double)n;
        const float nv = (float)mean;
        const float ov = cents[c * kDim + t];
        cents[c * kDim + t] = nv;
        delta2 = (nv - ov) * (nv - ov);
    }  // an empty cluster keeps its centroid
    __shared__ float red[128];
    red[t] = delta2;
    __syncthreads();
    for (int sft = 64; sft > 0; sft >>= 1) {
        if (t < sft) red[t] += red[t + sft];
        __syncthreads();
    }
    if (t == 0) shift[c] = (double)red[0];
}

hipError_t launch_kmeans_update(const float* x, const int32_t* assign, int64_t rows, int nlist, float* cents,
                                unsigned long long* acc, int32_t* counts, double* shift, hipStream_t s) {
    hipError_t e = hipMemsetAsync(acc, 0, (size_t)nlist * kDim * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(counts, 0, (size_t)nlist * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kmeans_accum_kernel, dim3(4096), dim3(256), 0, s, x, assign, rows, acc, counts);
    hipLaunchKernelGGL(kmeans_finalize_kernel, dim3(nlist), dim3(128), 0, s, cents, acc, counts, shift);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// IVF coarse stage + probe selection in one launch (IVFIndex.cpp:654-666 centroid scores, :697-723
// top-nprobe): one 256-thread workgroup per query.  Distances to all centroids with the same
// 8-lanes-per-row dot product as the list scan (centroids are L2 resident), then selection without
// sorting rounds: the nprobe-th smallest of the 256 per-thread minima bounds the answer, the few
// scores under that bound are compacted and ranked by counting (every candidate counts how many
// others precede it in (dist, id) order and writes itself to that slot).  Deterministic.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_add_xor1(float x);
__device__ __forceinline__ float dpp_add_xor2(float x);
__device__ __forceinline__ float dpp_add_half_mirror(float x);

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
__global__ __launch_bounds__(256) void ivf_coarse_mfma_kernel(const float* __restrict__ q, int B, const float* __restrict__ cents,
                                                             const float* __restrict__ cnorm, int nlist, int metric,
                                                             float* __restrict__ scores, int ld, IvfMulti mb, IvfGroup grp) {
    {
        const long long y = blockIdx.y;
        q = mb_adv(q, y * mb.q);
        scores = mb_adv(scores, y * mb.slab);
    }
    __shared__ float qn_s[kMaxBatch];
    // Both operands go through LDS: the workgroup's 64 centroids and the batch's queries are read from global memory as
    // whole rows (a wave instruction = 1 KB in one piece) and the MFMA fragments are cut out of LDS.  Read as fragments
    // straight from memory, every load instruction touched 64 separate 16-byte pieces 512 bytes apart, and the address
    // unit, not the arithmetic, set the kernel's time (13 us).  Rows are 132 floats apart in LDS: fragment reads of
    // 8 neighbouring rows then fall into different banks.
    constexpr int LD = kDim + 4;
    __shared__ __attribute__((aligned(16))) float q_s[kMaxBatch * LD];
    __shared__ __attribute__((aligned(16))) float c_s[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = ((int)blockIdx.x * 4 + wave) * 16;  // this wave's centroid tile (the centroid array has kScanPadRows spare rows)
    {
        f32x4 vc[8], vq[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i;  // float4 (row, column) of the 64 x 32 tile
            vc[i] = *reinterpret_cast<const f32x4*>(cents + ((int64_t)blockIdx.x * 64 + (idx >> 5)) * kDim + 4 * (idx & 31));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            vq[i] = (idx >> 5) < B ? *reinterpret_cast<const f32x4*>(q + (idx >> 5) * kDim + 4 * (idx & 31)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<f32x4*>(c_s + (idx >> 5) * LD + 4 * (idx & 31)) = vc[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<f32x4*>(q_s + (idx >> 5) * LD + 4 * (idx & 31)) = vq[i];
        }
    }
    const f32x4 cn = *reinterpret_cast<const f32x4*>(cnorm + row0 + 4 * g);  // padded by 64
    __syncthreads();
    f32x4 a[8], qf[2][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = *reinterpret_cast<const f32x4*>(c_s + (wave * 16 + r) * LD + 16 * c + 4 * g);
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
        // wide pipeline: the first block of every batch also writes the queries as bytes, their constant terms and the
        // batch's "byte valued" verdict (what seed_qnorm_kernel does for the brute-force scans); a thread converts 16
        // neighbouring components and stores them as one 16-byte word
        if (grp.w_q8 != nullptr && blockIdx.x == 0) {
            const int64_t qslot = (int64_t)blockIdx.y * kMaxBatch + row;
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
            if (tid == 0) grp.w_invalid[blockIdx.y] = bad ? 1 : 0;
            if (tid == 0 && blockIdx.y == 0 && grp.w_overflow) grp.w_overflow[0] = 0;  // the previous group's verdict has been read
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int qrow = h * 16 + r;
        if (h * 16 >= B) break;  // workgroup-uniform
        const bool qv = qrow < B;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[h][c][i], acc, 0, 0, 0);
        if (qv) {
            const float qn = qn_s[qrow];
            f32x4 d;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = metric ? -acc[j] : fmaf(-2.0f, acc[j], qn + cn[j]);
                if (!(v == v) || row0 + 4 * g + j >= nlist) v = VS_INF;  // NaN never wins; rows past nlist are padding
                d[j] = v;
            }
            *reinterpret_cast<f32x4*>(scores + (int64_t)qrow * ld + row0 + 4 * g) = d;
        }
    }
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
        probes = mb_adv(probes, y * grp.mb.slab);
        grp.qoff = mb_adv(grp.qoff, y * grp.mb.slab);
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
        unsigned x = 0;
#pragma unroll 4
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned t = x | (1u << bit);
            int below = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) below += __popcll(__ballot(hi[i] < t));
            if (below < nprobe) x = t;  // wave-uniform
        }
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
    if (!grp.lcnt) return;
    if (grp.w_cnt) {
        // wide pipeline: every (query, probe) pair takes a slot in its list's table (one global atomic per pair; a list
        // without rows here has no records in the plan, its table is simply never read)
        __syncthreads();
        const int c = tid < nprobe ? s_probe[tid] : -1;
        if (c >= 0) {
            const int sb = (int)blockIdx.y / kIvfWideBatches;
            const int slot = atomicAdd(grp.w_cnt + sb * ivf_wide_plan_words(nlist) + (int64_t)c * kIvfWideCntStride, 1);  // < w_q: once per query
            // the table holds the byte offset of the query's 128 staged bytes in the scan's LDS (slot in the launch group * 128)
            grp.w_lq[((int64_t)sb * nlist + c) * grp.w_q + slot] = (((int)blockIdx.y % kIvfWideBatches) * kMaxBatch + b) * kDim;
        }
        PICK_STAMP(5);
        return;
    }
    // ---- every (query, probe) gets a window [qoff[p], qoff[p+1]) in the query's candidate-score array (probe order) ----
    __syncthreads();
    int sz = 0;
    if (tid < nprobe) {
        const int c = s_probe[tid];
        sz = c >= 0 ? grp.offsets[c + 1] - grp.offsets[c] : 0;
    }
    // exclusive scan of the <= 256 window sizes: wave scan + four wave totals
    __shared__ int s_wt[4];
    const int lane = tid & 63, wave = tid >> 6;
    int incl = sz;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wt[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) woff += s_wt[w];
        tot += s_wt[w];
    }
    if (tid < nprobe) grp.qoff[(int64_t)b * (kIvfMaxProbe + 1) + tid] = woff + incl - sz;
    if (tid == 0) {
        grp.qoff[(int64_t)b * (kIvfMaxProbe + 1) + nprobe] = tot;
        if (grp.cand_count) atomicAdd(grp.cand_count, (unsigned long long)tot);
    }
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


// ------------------------------------------------------------------------------------------------
// Grouping + work plan, one workgroup per batch (after ivf_coarse_pick_kernel): every (query, probe) takes a slot
// in its list's query set (LDS counters: no global atomics, no fences -- with one workgroup per query doing this
// through global atomics and a last-arriver, the grouping cost grew with the number of batches in flight), then
// the plan of the list scan is written: the 32-row units of every chunk whose list is probed by some query.
// ------------------------------------------------------------------------------------------------
constexpr int kPlanSplit = 8;  // workgroups per batch: each writes the unit records of a slice of the chunk table
__global__ __launch_bounds__(1024) void ivf_group_plan_kernel(const int32_t* __restrict__ probes, int B, int nlist, int nprobe,
                                                              IvfGroup grp) {
    {   // multi-batch launch: this workgroup's batch
        const long long y = blockIdx.y;
        probes = mb_adv(probes, y * grp.mb.slab);
        grp.lcnt = mb_adv(grp.lcnt, y * grp.mb.zslab);
        grp.n_units = mb_adv(grp.n_units, y * grp.mb.zslab);
        grp.lq = mb_adv(grp.lq, y * grp.mb.slab);
        grp.lbase = mb_adv(grp.lbase, y * grp.mb.slab);
        grp.qoff = mb_adv(grp.qoff, y * grp.mb.slab);
        grp.units = mb_adv(grp.units, y * grp.mb.slab);
    }
    __shared__ int cnt_s[kIvfFastNlist];
    __shared__ int s_carry;
    __shared__ int s_wtot[16];
    const int tid = threadIdx.x;
    const bool first = blockIdx.x == 0;  // the slice that also writes the lists' query sets
    for (int c = tid; c < nlist; c += 1024) cnt_s[c] = 0;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    // every slice counts the queries per list (it needs them for its unit records); the slot a query takes in its list's
    // set is whatever slice 0 hands out -- only that slice writes lq / lbase
    for (int e = tid; e < B * nprobe; e += 1024) {
        const int b = e / nprobe, pp = e - b * nprobe;
        const int c = probes[(int64_t)b * nprobe + pp];
        if (c < 0) continue;
        const int o0 = grp.qoff[(int64_t)b * (kIvfMaxProbe + 1) + pp], o1 = grp.qoff[(int64_t)b * (kIvfMaxProbe + 1) + pp + 1];
        if (o1 == o0) continue;  // empty (or not resident) list
        const int slot = atomicAdd(&cnt_s[c], 1);  // < 32: a list is probed at most once per query, B <= 32
        if (first) {
            grp.lq[c * kMaxBatch + slot] = b;
            grp.lbase[c * kMaxBatch + slot] = (int64_t)b * grp.cand_stride + o0;
        }
    }
    __syncthreads();
    if (first)
        for (int c = tid; c < nlist; c += 1024) grp.lcnt[c] = cnt_s[c];
    if (!grp.units) return;
    const int pl = tid & 63, wv = tid >> 6;
    const int nsl = (int)gridDim.x;
    const int c0 = (int)((long long)grp.n_chunks * blockIdx.x / nsl), c1 = (int)((long long)grp.n_chunks * (blockIdx.x + 1) / nsl);
    auto units_of = [&](int chunk) { return cnt_s[grp.chunk_list[chunk]] > 0 ? (grp.chunk_rows[chunk] + 31) >> 5 : 0; };
    {   // units of the chunks in front of this slice
        int pre = 0;
        for (int chunk = tid; chunk < c0; chunk += 1024) pre += units_of(chunk);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o);
        if (pl == 0) s_wtot[wv] = pre;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += s_wtot[w];
            s_carry = t;
        }
        __syncthreads();
    }
    for (int base = c0; base < c1; base += 1024) {
        const int chunk = base + tid;
        const int nu = chunk < c1 ? units_of(chunk) : 0;
        int incl = nu;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (pl >= o) incl += t;
        }
        if (pl == 63) s_wtot[wv] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const int t = s_wtot[w];
            if (w < wv) woff += t;
            tot += t;
        }
        const int pos = s_carry + woff + incl - nu;
        if (nu > 0) {
            // unit record: first row, end row of the chunk, list | queries << 16, first row of the list -- everything
            // the scan needs, so that a wave's next unit costs ONE (prefetched) load instead of five dependent ones
            const int c = grp.chunk_list[chunk];
            const int r0 = grp.chunk_row0[chunk];
            const int r_end = r0 + grp.chunk_rows[chunk];
            const int ls = grp.offsets[c];
            const int nq = min(cnt_s[c], kMaxBatch);
            for (int i = 0; i < nu; ++i) reinterpret_cast<int4*>(grp.units)[pos + i] = make_int4(r0 + 32 * i, r_end, c | (nq << 16), ls);
        }
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0 && (int)blockIdx.x == nsl - 1) *grp.n_units = s_carry;  // the last slice ends at the total
}

hipError_t launch_ivf_coarse_pick(const float* q, int B, const float* cents, const float* cnorm, int nlist, int nprobe,
                                  int metric, float* scores, int ld, int32_t* probes, const IvfGroup& grp, hipStream_t s, int n_batches) {
    if (nprobe > 256 || nlist > kIvfFastNlist || ld < ((nlist + 63) & ~63)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivf_coarse_mfma_kernel, dim3((nlist + 63) / 64, n_batches), dim3(256), 0, s, q, B, cents, cnorm, nlist, metric,
                       scores, ld, grp.mb, grp);
    if (nlist <= 1024) hipLaunchKernelGGL(ivf_pick_kernel<4>, dim3(B, n_batches), dim3(256), 0, s, scores, ld, nlist, nprobe, probes, grp);
    else if (nlist <= 2048) hipLaunchKernelGGL(ivf_pick_kernel<8>, dim3(B, n_batches), dim3(256), 0, s, scores, ld, nlist, nprobe, probes, grp);
    else hipLaunchKernelGGL(ivf_pick_kernel<16>, dim3(B, n_batches), dim3(256), 0, s, scores, ld, nlist, nprobe, probes, grp);
    return hipGetLastError();
}

// grouping + work plan: one workgroup per batch
hipError_t launch_ivf_group_plan(const int32_t* probes, int B, int nlist, int nprobe, const IvfGroup& grp, hipStream_t s, int n_batches) {
    if (!grp.lcnt) return hipSuccess;
    hipLaunchKernelGGL(ivf_group_plan_kernel, dim3(n_batches > 1 ? kPlanSplit : 2 * kPlanSplit, n_batches), dim3(1024), 0, s, probes, B, nlist, nprobe, grp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// List-major IVF scan.  One workgroup per (list, 1024-row chunk); it runs only if some query of
// the batch probes the list and then serves ALL of them: the rows are read from HBM once per batch
// instead of once per (query, probe).  8 lanes share a row (whole 128-byte lines per
// wave-instruction, as in ivf_scan_kernel); the rows of a step stay in registers while the
// workgroup's queries (staged in LDS) are applied one after the other.  Distances are not ranked
// here: they go to the query's candidate-score array in probe order (computeDotProductsContiguous
// writes `scores[i]` the same way, IVFIndex.cpp:313-320) and ivf_select_kernel picks the top-k.
// ------------------------------------------------------------------------------------------------
constexpr int kIvfScanThreads = 1024;  // 16 waves: a popular list's rows are spread thin

__global__ __launch_bounds__(kIvfScanThreads) void ivf_list_scan_kernel(const IvfListScanParams p, int n_chunks) {
    __shared__ __attribute__((aligned(16))) float q_s[kMaxBatch * kDim];
    __shared__ float qn_s[kMaxBatch];
    __shared__ long long base_s[kMaxBatch];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int rr = lane >> 3, s8 = lane & 7;
    // A few hundred resident workgroups walk the (list, chunk) table; items nobody probes cost one
    // cached load (dispatching a workgroup per item would cost more than the skipped items' work).
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int c = p.chunk_list[chunk];
        const int nq = min(p.lcnt[c], kMaxBatch);
        if (nq <= 0) continue;  // workgroup-uniform
        const int list_start = p.offsets[c];
        const int r_begin = p.chunk_row0[chunk];
        const int r_end = r_begin + p.chunk_rows[chunk];
        __syncthreads();  // previous item's LDS readers are done
        // stage the list's queries (gathered by index) and their norms (reference order)
        {
            const int s = tid >> 5, c4 = tid & 31;  // 32 x 32 float4 chunks
            if (s < nq) {
                const int qi = p.lq[c * kMaxBatch + s];
                *reinterpret_cast<f32x4*>(q_s + s * kDim + 4 * c4) = *reinterpret_cast<const f32x4*>(p.q + (int64_t)qi * kDim + 4 * c4);
                if (c4 == 0) base_s[s] = p.lbase[c * kMaxBatch + s];
            }
        }
        __syncthreads();
        if (tid < 256) {
            const int s = tid >> 3, j = tid & 7;
            float acc = 0.f;
            if (s < nq) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float x = q_s[s * kDim + 8 * i + j];
                    acc = fmaf(x, x, acc);
                }
            }
            const int b8 = lane & ~7;
            float sum = __shfl(acc, b8);
#pragma unroll
            for (int u = 1; u < 8; ++u) sum = sum + __shfl(acc, b8 + u);
            if (j == 0 && s < nq) qn_s[s] = sum;
        }
        __syncthreads();
        constexpr int U = 2;  // row groups in flight per wave
        for (int row0 = r_begin + wave * 8; row0 < r_end; row0 += 128 * U) {
            f32x4 v[U][4];
            float vn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + 128 * u + rr;
                const int rowc = row < r_end ? row : r_end - 1;
                const float* src = p.vecs + (int64_t)rowc * kDim + 4 * s8;
#pragma unroll
                for (int m = 0; m < 4; ++m) v[u][m] = *reinterpret_cast<const f32x4*>(src + 32 * m);
                vn[u] = p.vnorm[rowc];
            }
            for (int s = 0; s < nq; ++s) {
                f32x4 qf[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) qf[m] = *reinterpret_cast<const f32x4*>(q_s + s * kDim + 4 * (s8 + 8 * m));
                const float qn = qn_s[s];
                float* dst = p.cand + base_s[s];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float acc = 0.f;
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc = fmaf(v[u][m][i], qf[m][i], acc);
                    acc = dpp_add_xor1(acc);
                    acc = dpp_add_xor2(acc);
                    acc = dpp_add_half_mirror(acc);
                    const float d = p.metric ? -acc : fmaf(-2.0f, acc, qn + vn[u]);
                    const int row = row0 + 128 * u + rr;
                    if (s8 == 0 && row < r_end) dst[row - list_start] = d;
                }
            }
        }
    }
}

// Planned variant: the batch's queries are staged once per workgroup; after that every wave works alone on
// 32-row units of the plan (no barriers, no per-list staging): 4 row groups of 8 rows are loaded (8 lanes per
// row, whole 128-byte lines per wave-instruction) and scored against every query that probes the unit's list.
constexpr int kIvfUnitThreads = 256;
__global__ __launch_bounds__(kIvfUnitThreads) void ivf_unit_scan_kernel(IvfListScanParams p, const int32_t* __restrict__ units,
                                                                      const int32_t* __restrict__ n_units_ptr, int B) {
    {   // multi-batch launch: this workgroup's batch
        const long long y = blockIdx.y;
        p.q = mb_adv(p.q, y * p.mb.q);
        p.lcnt = mb_adv(p.lcnt, y * p.mb.zslab);
        p.lq = mb_adv(p.lq, y * p.mb.slab);
        p.lbase = mb_adv(p.lbase, y * p.mb.slab);
        p.cand = mb_adv(p.cand, y * p.mb.slab);
        p.slotmin = mb_adv(p.slotmin, y * p.mb.zslab);
        p.bkt = mb_adv(p.bkt, y * p.mb.zslab);
        units = mb_adv(units, y * p.mb.slab);
        n_units_ptr = mb_adv(n_units_ptr, y * p.mb.zslab);
    }
    __shared__ __attribute__((aligned(16))) float q_s[kMaxBatch * kDim];
    __shared__ float qn_s[kMaxBatch];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = lane >> 3, s8 = lane & 7;
    const int n_units = *n_units_ptr;
    if ((int)blockIdx.x * 4 >= n_units) return;  // workgroup-uniform
    for (int i = tid; i < B * 32; i += kIvfUnitThreads)
        *reinterpret_cast<f32x4*>(q_s + 4 * i) = *reinterpret_cast<const f32x4*>(p.q + 4 * (int64_t)i);
    __syncthreads();
    {   // ||q||^2 in the reference's AVX2 order (cpu_baseline.cpp:95-114): 8 lanes per query
        const int s = tid >> 3, j = tid & 7;
        float acc = 0.f;
        if (s < B) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float x = q_s[s * kDim + 8 * i + j];
                acc = fmaf(x, x, acc);
            }
        }
        const int b8 = lane & ~7;
        float sum = __shfl(acc, b8);
#pragma unroll
        for (int u = 1; u < 8; ++u) sum = sum + __shfl(acc, b8 + u);
        if (j == 0 && s < B) qn_s[s] = sum;
    }
    __syncthreads();
    const int nw = (int)gridDim.x * 4;
    // ---- int8 path: rows stored as (x - 128) bytes, the batch's queries converted once per workgroup; a unit is two
    //      16-row MFMA tiles against 16-query column blocks of its list's query set; distances are the same integers
    //      the fp32 path computes (scan_kernel PREC = 1).  A batch with a non-integer query uses the fp32 rows below.
    if (p.vecs_u8 && p.metric == 0) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        typedef int i32x4_u __attribute__((ext_vector_type(4), aligned(4)));
        typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
        __shared__ __attribute__((aligned(16))) int q8_s[kMaxBatch * 32];  // [query][128 bytes]
        __shared__ int qsum_s[kMaxBatch];
        if (tid < kMaxBatch) qsum_s[tid] = 0;
        __syncthreads();
        bool q_ok = true;
        for (int i = tid; i < B * 32; i += kIvfUnitThreads) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(q_s + 4 * i);
            unsigned word = 0;
            int part = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int xi = (int)v[e];
                q_ok = q_ok && ((float)xi == v[e]) && xi >= 0 && xi <= 255;
                part += xi - 128;
                word |= ((unsigned)((xi - 128) & 0xff)) << (8 * e);
            }
            q8_s[i] = (int)word;
            atomicAdd(&qsum_s[i >> 5], part);
        }
        if (__syncthreads_and(q_ok ? 1 : 0)) {
            const int r = lane & 15, g = lane >> 4;
            const int4* recs = reinterpret_cast<const int4*>(units);
            int u = (int)blockIdx.x * 4 + wave;
            int4 rec = u < n_units ? recs[u] : make_int4(0, 0, 0, 0);
            for (; u < n_units; u += nw) {
                const int r0 = __builtin_amdgcn_readfirstlane(rec.x);
                const int r_end = __builtin_amdgcn_readfirstlane(rec.y);
                const int c = __builtin_amdgcn_readfirstlane(rec.z) & 0xffff;
                const int nq = __builtin_amdgcn_readfirstlane(rec.z) >> 16;
                const int list_start = __builtin_amdgcn_readfirstlane(rec.w);
                if (u + nw < n_units) rec = recs[u + nw];  // the next unit's record, in flight during this unit
                // A operands of the two tiles: bytes 16 g .. and 64 + 16 g .. of row r0 + 16 t + r (rows past the chunk
                // read the chunk's last row and are never written)
                i32x4 a0[2], a1[2], rt[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int row = min(r0 + 16 * t + r, r_end - 1);
                    a0[t] = *reinterpret_cast<const i32x4*>(p.vecs_u8 + (int64_t)row * kDim + 16 * g);
                    a1[t] = *reinterpret_cast<const i32x4*>(p.vecs_u8 + (int64_t)row * kDim + 64 + 16 * g);
                    // row terms of rows 4 g .. 4 g + 3 of the tile: one 4-byte-aligned vector load (rows past the chunk
                    // belong to the next list or to the array's 64 spare entries: readable, never used)
                    rt[t] = *reinterpret_cast<const i32x4_u*>(p.rterm + r0 + 16 * t + 4 * g);
                }
                for (int cb = 0; cb < nq; cb += 16) {
                    const int sq = cb + r;  // this lane's query slot in the list's query set
                    const bool live = sq < nq;
                    const int qi = live ? p.lq[c * kMaxBatch + sq] : 0;
                    const long long cbase = live ? p.lbase[c * kMaxBatch + sq] : 0;
                    const i32x4 b0 = *reinterpret_cast<const i32x4*>(q8_s + qi * 32 + 4 * g);
                    const i32x4 b1 = *reinterpret_cast<const i32x4*>(q8_s + qi * 32 + 16 + 4 * g);
                    const int qterm = (int)qn_s[qi] - 256 * qsum_s[qi] - 4194304;
                    float umin = VS_INF;  // this unit's minimum score for the lane's query
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        i32x4 acc = {0, 0, 0, 0};
                        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[t], b0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[t], b1, acc, 0, 0, 0);
                        const int row = r0 + 16 * t + 4 * g;
                        f32x4 dv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) dv[j] = (float)(qterm + rt[t][j] - 2 * acc[j]);
                        float* dst = p.cand + cbase + (row - list_start);
                        if (live && row + 3 < r_end) {
                            *reinterpret_cast<f32x4_u*>(dst) = dv;  // 4 consecutive scores of the query's window
                            umin = fminf(umin, fminf(fminf(dv[0], dv[1]), fminf(dv[2], dv[3])));
                        } else if (live) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (row + j < r_end) {
                                    dst[j] = dv[j];
                                    umin = fminf(umin, dv[j]);
                                }
                        }
                    }
                    if (p.slotmin || p.bkt) {  // fold the 4 lanes of the query column, one atomic per (unit, query)
                        umin = fminf(umin, __shfl_xor(umin, 16));
                        umin = fminf(umin, __shfl_xor(umin, 32));
                        if (live && g == 0 && umin < VS_INF) {
                            if (p.bkt)  // block of the query's candidate array in which this unit's scores start
                                atomicMax(p.bkt + (long long)qi * p.nbk + (int)((cbase - (long long)qi * p.cand_stride + (r0 - list_start)) >> 5),
                                          ~f32_ordered(umin));
                            else
                                atomicMax(p.slotmin + qi * kIvfSlots + (u & (kIvfSlots - 1)), ~f32_ordered(umin));
                        }
                    }
                }
            }
            return;
        }
    }
    for (int u = (int)blockIdx.x * 4 + wave; u < n_units; u += nw) {
        const int4 rec = reinterpret_cast<const int4*>(units)[u];
        const int r0 = __builtin_amdgcn_readfirstlane(rec.x);
        const int r_end = __builtin_amdgcn_readfirstlane(rec.y);
        const int c = __builtin_amdgcn_readfirstlane(rec.z) & 0xffff;
        const int nq = __builtin_amdgcn_readfirstlane(rec.z) >> 16;
        const int list_start = __builtin_amdgcn_readfirstlane(rec.w);
        f32x4 v[4][4];
        float vn[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = r0 + 8 * g + rr;
            const int rowc = row < r_end ? row : r_end - 1;
            const float* src = p.vecs + (int64_t)rowc * kDim + 4 * s8;
#pragma unroll
            for (int m = 0; m < 4; ++m) v[g][m] = *reinterpret_cast<const f32x4*>(src + 32 * m);
            vn[g] = p.vnorm[rowc];
        }
        for (int s = 0; s < nq; ++s) {
            const int qi = p.lq[c * kMaxBatch + s];
            float* dst = p.cand + p.lbase[c * kMaxBatch + s] + (r0 - list_start);
            f32x4 qf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) qf[m] = *reinterpret_cast<const f32x4*>(q_s + qi * kDim + 4 * (s8 + 8 * m));
            const float qn = qn_s[qi];
            float umin = VS_INF;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc = fmaf(v[g][m][i], qf[m][i], acc);
                acc = dpp_add_xor1(acc);
                acc = dpp_add_xor2(acc);
                acc = dpp_add_half_mirror(acc);
                const float d = p.metric ? -acc : fmaf(-2.0f, acc, qn + vn[g]);
                if (s8 == 0 && r0 + 8 * g + rr < r_end) {
                    dst[8 * g + rr] = d;
                    umin = fminf(umin, d);
                }
            }
            if (p.slotmin || p.bkt) {  // the unit's minimum score for this query: fold the row lanes, one atomic
                umin = fminf(umin, __shfl_xor(umin, 8));
                umin = fminf(umin, __shfl_xor(umin, 16));
                umin = fminf(umin, __shfl_xor(umin, 32));
                if (lane == 0 && umin < VS_INF) {
                    if (p.bkt)
                        atomicMax(p.bkt + (long long)qi * p.nbk + (int)((p.lbase[c * kMaxBatch + s] - (long long)qi * p.cand_stride + (r0 - list_start)) >> 5),
                                  ~f32_ordered(umin));
                    else
                        atomicMax(p.slotmin + qi * kIvfSlots + (u & (kIvfSlots - 1)), ~f32_ordered(umin));
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// k-means++ seeding.  kpp_update_kernel: one workgroup per kKppBlockRows rows, 8 lanes per row (as in the IVF
// scans); kpp_pick_kernel: one workgroup finds the block, then the row, where the running sum passes u * total.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kpp_update_kernel(const float* __restrict__ x, const float* __restrict__ xnorm, int64_t rows,
                                                         const float* __restrict__ centre, float* __restrict__ d2,
                                                         double* __restrict__ block_sums) {
    __shared__ double wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rr = lane >> 3, s8 = lane & 7;
    f32x4 cf[4];
    float cn = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        cf[m] = *reinterpret_cast<const f32x4*>(centre + 4 * (s8 + 8 * m));
#pragma unroll
        for (int i = 0; i < 4; ++i) cn = fmaf(cf[m][i], cf[m][i], cn);
    }
    cn = dpp_add_xor1(cn);
    cn = dpp_add_xor2(cn);
    cn = dpp_add_half_mirror(cn);
    const int64_t row_begin = (int64_t)blockIdx.x * kKppBlockRows;
    double acc = 0.0;
    for (int r0 = wave * 8; r0 < kKppBlockRows; r0 += 32) {
        const int64_t row = row_begin + r0 + rr;
        const bool ok = row < rows;
        const float* src = x + (ok ? row : 0) * kDim + 4 * s8;
        float dot = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 32 * m);
#pragma unroll
            for (int i = 0; i < 4; ++i) dot = fmaf(v[i], cf[m][i], dot);
        }
        dot = dpp_add_xor1(dot);
        dot = dpp_add_xor2(dot);
        dot = dpp_add_half_mirror(dot);
        if (ok && s8 == 0) {
            const float d = fmaxf(fmaf(-2.0f, dot, xnorm[row] + cn), 0.f);
            const float nd = fminf(d2[row], d);
            d2[row] = nd;
            acc += (double)nd;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (tid == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void kpp_pick_kernel(const float* __restrict__ x, int64_t rows, const float* __restrict__ d2,
                                                        const double* __restrict__ block_sums, int n_blocks, double u,
                                                        float* __restrict__ out_centre) {
    __shared__ double s_part[1024];
    __shared__ double s_target, s_before;
    __shared__ int s_block;
    __shared__ long long s_row;
    const int tid = threadIdx.x;
    // total and the block where the running sum passes the target: block sums through LDS, 1024 at a time
    if (tid == 0) {
        s_before = 0.0;
        s_block = -1;
    }
    __syncthreads();
    double total = 0.0;
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
        s_part[tid] = b0 + tid < n_blocks ? block_sums[b0 + tid] : 0.0;
        __syncthreads();
        for (int t = 0; t < 1024 && b0 + t < n_blocks; ++t) total += s_part[t];  // every thread: the same order, the same sum
        __syncthreads();
    }
    const double target = u * total;
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
        s_part[tid] = b0 + tid < n_blocks ? block_sums[b0 + tid] : 0.0;
        __syncthreads();
        if (tid == 0 && s_block < 0) {
            double run = s_before;
            for (int t = 0; t < 1024 && b0 + t < n_blocks; ++t) {
                if (run + s_part[t] > target) {
                    s_block = b0 + t;
                    break;
                }
                run += s_part[t];
            }
            s_before = run;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (s_block < 0) s_block = n_blocks - 1;
        s_target = target;
        s_row = -1;
    }
    __syncthreads();
    const int64_t row = (int64_t)s_block * kKppBlockRows + tid;  // kKppBlockRows == blockDim.x
    s_part[tid] = row < rows ? (double)d2[row] : 0.0;
    __syncthreads();
    if (tid == 0) {
        double run = s_before;
        long long pick = -1;
        const int64_t last = min<int64_t>(rows, ((int64_t)s_block + 1) * kKppBlockRows) - 1;
        for (int t = 0; t < 1024; ++t) {
            run += s_part[t];
            if (run > s_target && s_part[t] > 0.0) {
                pick = (long long)s_block * kKppBlockRows + t;
                break;
            }
        }
        if (pick < 0) {  // rounding at the very end of the range (or an all-zero block): last row with d2 > 0, else the last row
            pick = last;
            for (int t = 1023; t >= 0; --t)
                if (s_part[t] > 0.0) {
                    pick = (long long)s_block * kKppBlockRows + t;
                    break;
                }
        }
        s_row = pick;
    }
    __syncthreads();
    if (tid < kDim) out_centre[tid] = x[s_row * kDim + tid];
}

hipError_t launch_kpp_step(const float* x, const float* xnorm, int64_t rows, float* cents, int c, float* d2, double* block_sums,
                           int n_blocks, double u, hipStream_t s) {
    hipLaunchKernelGGL(kpp_update_kernel, dim3(n_blocks), dim3(256), 0, s, x, xnorm, rows, cents + (size_t)(c - 1) * kDim, d2, block_sums);
    hipLaunchKernelGGL(kpp_pick_kernel, dim3(1), dim3(1024), 0, s, x, rows, d2, block_sums, n_blocks, u, cents + (size_t)c * kDim);
    return hipGetLastError();
}

hipError_t launch_ivf_unit_scan(const IvfListScanParams& p, const int32_t* units, const int32_t* n_units, int B, int num_cus,
                                hipStream_t s, int n_batches) {
    static int per_cu = 0;
    if (!per_cu) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ivf_unit_scan_kernel, kIvfUnitThreads, 0) != hipSuccess || nb < 1) nb = 4;
        per_cu = nb > 8 ? 8 : nb;
    }
    // one resident grid for a single batch; a multi-batch launch shares about two resident grids among its batches
    const int wgs = n_batches > 1 ? std::max(16, 2 * num_cus * per_cu / n_batches) : num_cus * per_cu;
    hipLaunchKernelGGL(ivf_unit_scan_kernel, dim3(wgs, n_batches), dim3(kIvfUnitThreads), 0, s, p, units, n_units, B);
    return hipGetLastError();
}

hipError_t launch_ivf_list_scan(const IvfListScanParams& p, int n_chunks, hipStream_t s) {
    if (n_chunks <= 0) return hipSuccess;
    hipLaunchKernelGGL(ivf_list_scan_kernel, dim3(n_chunks < 512 ? n_chunks : 512), dim3(kIvfScanThreads), 0, s, p, n_chunks);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Top-k of a query's candidate scores (the heap loop of IVFIndex.cpp:750-767 as a two-pass
// selection): pass 1 finds the 16 wave minima, whose k-th smallest bounds the k-th best score;
// pass 2 collects the few scores under the bound, maps them back to reordered positions and ranks
// them by counting in (dist, position) order; ids go through reorder_to_original (:774-779).
// ------------------------------------------------------------------------------------------------
constexpr int kSelSplit = 2;      // workgroups per query
constexpr int kSelCap = 4096;     // global candidate slots per query
constexpr int kSelBkPer = 16;     // block minima a thread keeps in registers (covers 4096 blocks = 131072 candidate scores)

// Pass 1 of the selection: every workgroup takes 1/8 of a query's candidate scores, computes its 256
// thread minima and their k-th smallest -- a bound backed by k distinct candidates -- and folds it
// into the query's bound with an atomic (complemented ordered floats: 0 = no bound yet, atomicMax).
__global__ __launch_bounds__(256) void ivf_bound_kernel(IvfSelectParams p) {
    {   // multi-batch launch: this workgroup's batch
        const long long y = blockIdx.y;
        p.cand = mb_adv(p.cand, y * p.mb.slab);
        p.qoff = mb_adv(p.qoff, y * p.mb.slab);
        p.probes = mb_adv(p.probes, y * p.mb.slab);
        p.gcand_d = mb_adv(p.gcand_d, y * p.mb.slab);
        p.gcand_p = mb_adv(p.gcand_p, y * p.mb.slab);
        p.tq = mb_adv(p.tq, y * p.mb.zslab);
        p.slotmin = mb_adv(p.slotmin, y * p.mb.zslab);
        p.bkt = mb_adv(p.bkt, y * p.mb.zslab);
        p.gcnt = mb_adv(p.gcnt, y * p.mb.zslab);
        p.gdone = mb_adv(p.gdone, y * p.mb.zslab);
        p.govf = mb_adv(p.govf, y * p.mb.zslab);
        p.out_d = mb_adv(p.out_d, y * p.mb.out_d);
        p.out_i = mb_adv(p.out_i, y * p.mb.out_i);
    }
    __shared__ float mn[256];
    const int q = blockIdx.x / kSelSplit, part = blockIdx.x % kSelSplit;
    const int tid = threadIdx.x;
    const int S = p.qoff[(int64_t)q * (kIvfMaxProbe + 1) + p.nprobe];
    const float* sc = p.cand + (int64_t)q * p.cand_stride;
    const int S4 = S >> 2;
    const int per = (S4 + kSelSplit - 1) / kSelSplit;
    const int a4 = part * per, b4 = min(S4, a4 + per);
    const f32x4* sc4 = reinterpret_cast<const f32x4*>(sc);
    float m = VS_INF;
    for (int i0 = a4 + tid; i0 < b4; i0 += 1024) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            v[u] = i < b4 ? sc4[i] : (f32x4){VS_INF, VS_INF, VS_INF, VS_INF};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fminf(m, fminf(fminf(v[u][0], v[u][1]), fminf(v[u][2], v[u][3])));
    }
    if (part == kSelSplit - 1)
        for (int i = 4 * S4 + tid; i < S; i += 256) m = fminf(m, sc[i]);
    mn[tid] = m;
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < 256; ++j) rank += (mn[j] < m || (mn[j] == m && j < tid)) ? 1 : 0;
    if (rank == min(p.k, 256) - 1 && m < VS_INF) atomicMax(p.tq + q, ~f32_ordered(m));
}


__global__ __launch_bounds__(256) void ivf_select_kernel(IvfSelectParams p) {
    {   // multi-batch launch: this workgroup's batch
        const long long y = blockIdx.y;
        p.cand = mb_adv(p.cand, y * p.mb.slab);
        p.qoff = mb_adv(p.qoff, y * p.mb.slab);
        p.probes = mb_adv(p.probes, y * p.mb.slab);
        p.gcand_d = mb_adv(p.gcand_d, y * p.mb.slab);
        p.gcand_p = mb_adv(p.gcand_p, y * p.mb.slab);
        p.tq = mb_adv(p.tq, y * p.mb.zslab);
        p.slotmin = mb_adv(p.slotmin, y * p.mb.zslab);
        p.bkt = mb_adv(p.bkt, y * p.mb.zslab);
        p.gcnt = mb_adv(p.gcnt, y * p.mb.zslab);
        p.gdone = mb_adv(p.gdone, y * p.mb.zslab);
        p.govf = mb_adv(p.govf, y * p.mb.zslab);
        p.out_d = mb_adv(p.out_d, y * p.mb.out_d);
        p.out_i = mb_adv(p.out_i, y * p.mb.out_i);
    }
    __shared__ float s_t;
    __shared__ int s_cnt, s_base, s_last;
    __shared__ float cd[1024];
    __shared__ int cpos[1024];
    __shared__ int s_off[257];
    __shared__ int s_probe[256];
    const int split = p.split;  // workgroups per query: 1 when the block minima filter the candidates, else kSelSplit
    const int q = blockIdx.x / split, part = blockIdx.x % split;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int S = p.qoff[(int64_t)q * (kIvfMaxProbe + 1) + p.nprobe];
    const float* sc = p.cand + (int64_t)q * p.cand_stride;
    for (int pp = tid; pp <= p.nprobe; pp += 256) s_off[pp] = p.qoff[(int64_t)q * (kIvfMaxProbe + 1) + pp];
    for (int pp = tid; pp < p.nprobe; pp += 256) s_probe[pp] = p.probes[(int64_t)q * p.nprobe + pp];
    __shared__ float s_slot[kIvfSlots];
    if (tid == 0) {
        s_cnt = 0;
        const unsigned u = p.slotmin ? 0u : p.tq[q];  // bound from ivf_bound_kernel (0 = none)
        s_t = u ? f32_unordered(~u) : VS_INF;
    }
    __shared__ __attribute__((aligned(16))) float s_mn[256];
    __shared__ int s_blk[1024];
    __shared__ int s_nblk;
    const unsigned* bk = p.bkt ? p.bkt + (long long)q * p.nbk : nullptr;
    const int nb_used = min(p.nbk, (S + 31) / 32 + 1);
    if (bk) {
        // per-block minima of the unit scan: the k-th smallest of the 256 thread minima is backed by k distinct candidates
        // (fixed trip count: the loads of a thread are independent and go out together; beyond 4096 blocks a plain loop)
        float mine = VS_INF;
        if (nb_used <= 256 * kSelBkPer) {
            unsigned bv[kSelBkPer];
#pragma unroll
            for (int i = 0; i < kSelBkPer; ++i) bv[i] = tid + 256 * i < nb_used ? bk[tid + 256 * i] : 0u;
#pragma unroll
            for (int i = 0; i < kSelBkPer; ++i)
                if (bv[i]) mine = fminf(mine, f32_unordered(~bv[i]));
        } else {
            for (int j = tid; j < nb_used; j += 256) {
                const unsigned u = bk[j];
                if (u) mine = fminf(mine, f32_unordered(~u));
            }
        }
        s_mn[tid] = mine;
        if (tid == 0) s_nblk = 0;
    }
    if (!bk && p.slotmin && tid < kIvfSlots) {
        const unsigned u = p.slotmin[q * kIvfSlots + tid];  // minima of disjoint sets of units (0 = empty slot)
        s_slot[tid] = u ? f32_unordered(~u) : VS_INF;
    }
    __syncthreads();
    if (bk) {
        const float v = s_mn[tid];
        int rank = 0;
        const f32x4* m4 = reinterpret_cast<const f32x4*>(s_mn);
#pragma unroll 8
        for (int j4 = 0; j4 < 64; ++j4) {
            const f32x4 w = m4[j4];
#pragma unroll
            for (int e = 0; e < 4; ++e) rank += (w[e] < v || (w[e] == v && 4 * j4 + e < tid)) ? 1 : 0;
        }
        if (rank == min(p.k, 256) - 1) s_t = v;
    }
    if (!bk && p.slotmin && tid < kIvfSlots) {
        // k-th smallest of the slot minima: k distinct candidates are at least that good
        const float v = s_slot[tid];
        int rank = 0;
        for (int j = 0; j < kIvfSlots; ++j) rank += (s_slot[j] < v || (s_slot[j] == v && j < tid)) ? 1 : 0;
        if (rank == min(p.k, kIvfSlots) - 1) s_t = v;
    }
    __syncthreads();
    const float T = s_t;
    auto pos_of = [&](int i) {  // candidate index -> probe (windows are in probe order) -> reordered position
        int lo = 0, hi = p.nprobe - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= i) lo = mid; else hi = mid - 1;
        }
        return p.offsets[s_probe[lo]] + (i - s_off[lo]);
    };
    auto consider = [&](int i, float d) {
        if (d <= T) {
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < 1024) {
                cd[pos] = d;
                cpos[pos] = pos_of(i);
            }
        }
    };
    if (bk) {
        // only the 32-score blocks that can hold a score under the bound are read: block j is needed when a unit that
        // starts in block j or j - 1 has a minimum <= T (a unit's 32 scores span at most two blocks)
        if (part == 0) {
            auto need_block = [&](int j, unsigned u0, unsigned u1) {
                const bool need = (u0 && f32_unordered(~u0) <= T) || (u1 && f32_unordered(~u1) <= T);
                if (need) {
                    const int pos = atomicAdd(&s_nblk, 1);
                    if (pos < 1024) s_blk[pos] = j;
                }
            };
            if (nb_used <= 256 * kSelBkPer) {
                unsigned b0[kSelBkPer], b1[kSelBkPer];
#pragma unroll
                for (int i = 0; i < kSelBkPer; ++i) {
                    const int j = tid + 256 * i;
                    b0[i] = j < nb_used ? bk[j] : 0u;
                    b1[i] = (j < nb_used && j > 0) ? bk[j - 1] : 0u;
                }
#pragma unroll
                for (int i = 0; i < kSelBkPer; ++i)
                    if (tid + 256 * i < nb_used) need_block(tid + 256 * i, b0[i], b1[i]);
            } else {
                for (int j = tid; j < nb_used; j += 256) need_block(j, bk[j], j > 0 ? bk[j - 1] : 0u);
            }
            __syncthreads();
            const int nblk = s_nblk;
            if (nblk > 1024) {
                if (tid == 0) s_cnt = 2048;  // too many blocks under the bound: the exact slow path below
            } else {
                for (int e = tid; e < nblk * 32; e += 256) {
                    const int i = 32 * s_blk[e >> 5] + (e & 31);
                    if (i < S) consider(i, sc[i]);
                }
            }
        }
    } else {
    // this workgroup's slice of the candidate array (16-byte loads, four in flight per thread)
    const int S4 = S >> 2;
    const int per = (S4 + split - 1) / split;
    const int a4 = part * per, b4 = min(S4, a4 + per);
    const f32x4* sc4 = reinterpret_cast<const f32x4*>(sc);  // the candidate array is 64-float aligned
    for (int i0 = a4 + tid; i0 < b4; i0 += 1024) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            v[u] = i < b4 ? sc4[i] : (f32x4){VS_INF, VS_INF, VS_INF, VS_INF};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            if (i < b4 && (v[u][0] <= T || v[u][1] <= T || v[u][2] <= T || v[u][3] <= T)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) consider(4 * i + e, v[u][e]);
            }
        }
    }
    if (part == split - 1)
        for (int i = 4 * S4 + tid; i < S; i += 256) consider(i, sc[i]);
    }
    __syncthreads();
    const int C = s_cnt;
    bool slow;
    if (split == 1) {
        // one workgroup per query: the candidates are ranked where they are (LDS), no hand-off
        slow = C > 1024;
        if (!slow) {
            // (dist, position) as one 64-bit key: two keys per 16-byte LDS read, one compare each
            typedef unsigned long long u64;
            typedef u64 u64x2 __attribute__((ext_vector_type(2)));
            __shared__ __attribute__((aligned(16))) u64 ck[1024 + 2];
            for (int c = tid; c < C + 2; c += 256) ck[c] = c < C ? (((u64)f32_ordered(cd[c]) << 32) | (unsigned)cpos[c]) : ~0ull;
            __syncthreads();
            for (int c = tid; c < C; c += 256) {
                const u64 key = ck[c];
                int rank = 0;
                const u64x2* p2 = reinterpret_cast<const u64x2*>(ck);
                for (int j = 0; j < (C + 1) / 2; ++j) {
                    const u64x2 w = p2[j];
                    rank += (w.x < key ? 1 : 0) + (w.y < key ? 1 : 0);
                }
                if (rank < p.k) {
                    const int id = cpos[c];
                    p.out_d[(int64_t)q * p.k + rank] = cd[c];
                    p.out_i[(int64_t)q * p.k + rank] = p.id_map ? p.id_map[id] : id;
                }
            }
            for (int c = C + tid; c < p.k; c += 256) {
                p.out_d[(int64_t)q * p.k + c] = VS_INF;
                p.out_i[(int64_t)q * p.k + c] = -1;
            }
            return;
        }
    } else {
    // append to the query's global candidate list (write-through), then take an arrival ticket
    if (tid == 0) s_base = atomicAdd(p.gcnt + q, C);
    __syncthreads();
    const int base = s_base;
    const bool overflow = C > 1024 || base + C > kSelCap;
    float* gd = p.gcand_d + (int64_t)q * kSelCap;
    int* gp = p.gcand_p + (int64_t)q * kSelCap;
    if (!overflow)
        for (int c = tid; c < C; c += 256) {
            __hip_atomic_store(gd + base + c, cd[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gp + base + c, cpos[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (overflow) __hip_atomic_store(p.govf + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(p.gdone + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == split - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the last workgroup of the query ranks the gathered candidates ----
    const int total = __hip_atomic_load(p.gcnt + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ovf = __hip_atomic_load(p.govf + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!ovf && total <= kSelCap) {
        for (int c = tid; c < total; c += 256) {
            const float d = gd[c];
            const int id = gp[c];
            int rank = 0;
            for (int j = 0; j < total; ++j) rank += lex_lt(gd[j], gp[j], d, id) ? 1 : 0;
            if (rank < p.k) {
                p.out_d[(int64_t)q * p.k + rank] = d;
                p.out_i[(int64_t)q * p.k + rank] = p.id_map ? p.id_map[id] : id;
            }
        }
        for (int c = total + tid; c < p.k; c += 256) {
            p.out_d[(int64_t)q * p.k + c] = VS_INF;
            p.out_i[(int64_t)q * p.k + c] = -1;
        }
        return;
    }
    }
    // Too many scores under the bound (massive ties / no usable bound): exact but slow path -- k rounds of a
    // workgroup-wide minimum in (dist, position) order over everything not emitted yet.
    {
        __shared__ float r_d[4];
        __shared__ int r_p[4];
        float last_d = -VS_INF;
        int last_p = -1;
        for (int round = 0; round < p.k; ++round) {
            float bd = VS_INF;
            int bp = 0x7fffffff;
            for (int i = tid; i < S; i += 256) {
                const float d = sc[i];
                if (!(d == d) || d < last_d || d > bd) continue;
                const int ps = pos_of(i);
                if (d == last_d && ps <= last_p) continue;
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
                p.out_d[(int64_t)q * p.k + round] = none ? VS_INF : bd;
                p.out_i[(int64_t)q * p.k + round] = none ? -1 : (p.id_map ? p.id_map[bp] : bp);
            }
            last_d = none ? VS_INF : bd;
            last_p = none ? 0x7fffffff : bp;
            __syncthreads();
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Wide IVF pipeline (see IvfWideParams).
// ------------------------------------------------------------------------------------------------
// Bound of a query = k-th smallest distance among the first kIvfTauRows rows of each of its two nearest resident lists
// (those rows are candidates, so k of them at most that far bound the k-th best of all candidates; two lists because the
// query's own neighbourhood is not always in the nearest one).  One wave per query: 16-row MFMA tiles with the query in
// column 0 of the B operand, distances through LDS, k rounds of a wave minimum.  Fewer than k rows: tau = +inf and the
// query is marked for the exact slow path.
__device__ __forceinline__ void ivf_tau_body(const IvfWideParams& p, const int wg) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int NSEG = 2;                      // lists sampled per query
    constexpr int SEGR = kIvfTauRows;            // rows per list
    constexpr int NR = NSEG * SEGR;              // distance slots per query
    // a workgroup = 2 queries x NSEG waves: wave (slot, sgm) scores segment sgm of its query, the segment-0 wave selects
    __shared__ __attribute__((aligned(16))) float dist[2][NR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef VS_STAMPS
#define TAU_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(7168 + wg) * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff); } while (0)
#else
#define TAU_STAMP(i)
#endif
    TAU_STAMP(0);
    const int slot = wave >> 1, myseg = wave & 1;
    const int qg = wg * 2 + slot;
    const int batch = qg >> 5, qi = qg & 31;
    const bool valid = batch < p.n_batches && qi < p.B;  // wave-uniform
    const int r = lane & 15, g = lane >> 4;
    int seg_start[NSEG], seg_rows[NSEG], seg_td[NSEG];
    int nseg = 0, total_rows = 0;
#pragma unroll
    for (int sgm = 0; sgm < NSEG; ++sgm) seg_start[sgm] = seg_rows[sgm] = seg_td[sgm] = 0;
    if (valid) {
        const int32_t* pr = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.probes) + (long long)batch * p.probes_batch_bytes) + qi * p.nprobe;
        // the first NSEG probed lists that are resident here: their first SEGR rows each (k rows in all are needed).  The
        // first LOOK probes and their lists' extents are fetched together (one after the other: a chain of cache round trips)
        constexpr int LOOK = 4;
        int cs[LOOK], o0[LOOK], o1[LOOK], td[LOOK];
#pragma unroll
        for (int i = 0; i < LOOK; ++i) cs[i] = i < p.nprobe ? pr[i] : -1;
#pragma unroll
        for (int i = 0; i < LOOK; ++i) {
            const int c = max(cs[i], 0);
            o0[i] = p.offsets[c];
            o1[i] = p.offsets[c + 1];
            td[i] = p.tdelta ? p.tdelta[c] : 0;
        }
        auto take_list = [&](int start, int len, int delta) {
            const int take = min(len, SEGR);
#pragma unroll
            for (int sgm = 0; sgm < NSEG; ++sgm)
                if (sgm == nseg) {
                    seg_start[sgm] = start;
                    seg_rows[sgm] = take;
                    seg_td[sgm] = delta;
                }
            ++nseg;
            total_rows += take;
        };
#pragma unroll
        for (int i = 0; i < LOOK; ++i)
            if (cs[i] >= 0 && o1[i] > o0[i] && nseg < NSEG) take_list(o0[i], o1[i] - o0[i], td[i]);
        for (int pp = LOOK; pp < p.nprobe && nseg < NSEG; ++pp) {  // (rare: lists without rows here among the nearest)
            const int c = pr[pp];
            if (c < 0) continue;
            const int len = p.offsets[c + 1] - p.offsets[c];
            if (len > 0) take_list(p.offsets[c], len, p.tdelta ? p.tdelta[c] : 0);
        }
    }
    const bool usable = valid && total_rows >= p.k;
    TAU_STAMP(1);
    for (int i = lane; i < SEGR; i += 64) dist[slot][myseg * SEGR + i] = VS_INF;
    if (usable) {
    const bool i8 = p.vecs_u8 && p.metric == 0 && p.invalid[batch] == 0;
    i32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    f32x4 qf[8];
    int qt = 0;
    float qn = 0.f;
    if (i8) {
        if (r == 0) {
            b0 = *reinterpret_cast<const i32x4*>(p.q8 + (int64_t)qg * kDim + 16 * g);
            b1 = *reinterpret_cast<const i32x4*>(p.q8 + (int64_t)qg * kDim + 64 + 16 * g);
        }
        qt = p.qterm[qg];
    } else {
        const float* qsrc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.q) + (long long)batch * p.q_batch_bytes) + qi * kDim;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            qf[c] = *reinterpret_cast<const f32x4*>(qsrc + 16 * c + 4 * g);
            if (r != 0) qf[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        qn = p.qnorm[qg];
    }
    {
        const int start = myseg ? seg_start[1] : seg_start[0], rows = myseg ? seg_rows[1] : seg_rows[0];
        const int tiles = (rows + 15) >> 4;
        float* dseg = &dist[slot][myseg * SEGR];
        if (i8 && p.vecs_t8) {
            // the tiled copy (see IvfWideParams): a list starts on a tile boundary there and a load is 1 KB in one piece
            const int tstart = start + (myseg ? seg_td[1] : seg_td[0]);
            const int8_t* rows_t = p.vecs_t8 + (int64_t)tstart * kDim + 16 * lane;
            constexpr int U = 8;  // tiles whose loads go out together
            for (int t0 = 0; t0 < tiles; t0 += U) {
                i32x4 a0[U], a1[U], rt[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = min(t0 + u, tiles - 1);
                    a0[u] = *reinterpret_cast<const i32x4*>(rows_t + (int64_t)t * 16 * kDim);
                    a1[u] = *reinterpret_cast<const i32x4*>(rows_t + (int64_t)t * 16 * kDim + 1024);
                    rt[u] = *reinterpret_cast<const i32x4*>(p.rterm_t + tstart + 16 * t + 4 * g);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u;
                    if (t >= tiles) break;
                    i32x4 acc = {0, 0, 0, 0};
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[u], b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[u], b1, acc, 0, 0, 0);
                    if (r == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (16 * t + 4 * g + j < rows) dseg[16 * t + 4 * g + j] = (float)(qt + rt[u][j] - 2 * acc[j]);
                    }
                }
            }
        } else if (i8) {
            constexpr int U = 4;  // tiles whose loads go out together
            for (int t0 = 0; t0 < tiles; t0 += U) {
                i32x4 a0[U], a1[U], rt[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = min(t0 + u, tiles - 1);
                    const int row = min(start + 16 * t + r, start + rows - 1);
                    a0[u] = *reinterpret_cast<const i32x4*>(p.vecs_u8 + (int64_t)row * kDim + 16 * g);
                    a1[u] = *reinterpret_cast<const i32x4*>(p.vecs_u8 + (int64_t)row * kDim + 64 + 16 * g);
                    rt[u] = *reinterpret_cast<const i32x4_u*>(p.rterm + start + 16 * t + 4 * g);  // padded by 64
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u;
                    if (t >= tiles) break;
                    i32x4 acc = {0, 0, 0, 0};
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[u], b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[u], b1, acc, 0, 0, 0);
                    if (r == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (16 * t + 4 * g + j < rows) dseg[16 * t + 4 * g + j] = (float)(qt + rt[u][j] - 2 * acc[j]);
                    }
                }
            }
        } else {
            for (int t = 0; t < tiles; ++t) {
                const int row = min(start + 16 * t + r, start + rows - 1);
                f32x4 a[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) a[c] = *reinterpret_cast<const f32x4*>(p.vecs + (int64_t)row * kDim + 16 * c + 4 * g);
                const f32x4 bn = *reinterpret_cast<const f32x4_u*>(p.vnorm + start + 16 * t + 4 * g);  // padded by 64
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[c][i], acc, 0, 0, 0);
                if (r == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (16 * t + 4 * g + j < rows) dseg[16 * t + 4 * g + j] = p.metric ? -acc[j] : fmaf(-2.0f, acc[j], qn + bn[j]);
                }
            }
        }
    }
    }
    __syncthreads();
    TAU_STAMP(2);
    if (!valid || myseg != 0) return;
    if (!usable) {
        if (lane == 0) {
            p.tau[qg] = VS_INF;
            p.slow[qg] = 1;
        }
        return;
    }
    const bool i8 = p.vecs_u8 && p.metric == 0 && p.invalid[batch] == 0;
    // k-th smallest of the NR slots: k rounds of a wave minimum over the lanes' private values
    float v[NR / 64];
#pragma unroll
    for (int i = 0; i < NR / 64; ++i) v[i] = dist[slot][i * 64 + lane];
    float kth = VS_INF;
    for (int round = 0; round < p.k; ++round) {
        float m = v[0];
#pragma unroll
        for (int i = 1; i < NR / 64; ++i) m = fminf(m, v[i]);
        const float wm = wave_min_f32(m);
        kth = wm;
        if (!(wm < VS_INF)) break;
        const unsigned long long mask = __ballot(m == wm);
        if (lane == __builtin_ctzll(mask)) {  // drop exactly one instance
            bool done = false;
#pragma unroll
            for (int i = 0; i < NR / 64; ++i)
                if (!done && v[i] == wm) {
                    v[i] = VS_INF;
                    done = true;
                }
        }
    }
    TAU_STAMP(3);
    if (lane == 0) {
        // integer distances (int8 path) are exact: the bound may sit right above the k-th value; fp32 rows are scored
        // with the same MFMA chain as the scan here, but leave slack anyway (the bound only filters)
        const float t = i8 ? next_up(kth) : kth + 1e-4f * fabsf(kth) + 1e-30f;
        p.tau[qg] = kth < VS_INF ? t : VS_INF;
        if (!(kth < VS_INF)) p.slow[qg] = 1;
    }
}

// Work plan of one super-batch (blockIdx.y), several workgroups each (see ivf_group_plan_kernel): records of bounded cost.
// A record is one kIvfWideUnit-row unit of a chunk whose list is probed, times one range of at most S of the slots of
// the list's query table: (first row, chunk end, list, first slot | end slot << 16).  S = 256 (a record costs between 1
// and 16 column blocks beside its rows; every further record of a unit reads the unit's rows again) unless the plan would
// not fit `units_cap`, then the next power of two that does (1024 = no split always fits).
constexpr int kIvfWideTiles = kIvfWideUnit / 16;  // 16-row MFMA tiles per unit
constexpr int kIvfWideSplits = 3;  // S = 256 << i
constexpr int kPlanThreads = 256, kPlanWaves = kPlanThreads / 64, kPlanClasses = 16;
__device__ __forceinline__ void ivf_plan_body(const IvfWideParams& p, const int sb, const int slice, const int nsl) {
    __shared__ int cnt_s[kIvfFastNlist];
    __shared__ int s_carry;
    __shared__ int s_tot[kPlanWaves][kIvfWideSplits];
    __shared__ int s_ctot[kPlanClasses + 1], s_cpre[kPlanClasses + 1], s_cpos[kPlanClasses + 1];
    __shared__ int s_shift;
    const int tid = threadIdx.x;
#ifdef VS_STAMPS
#define PLAN_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[(6144 + slice) * 16 + (i)] = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff); } while (0)
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
    constexpr int EARLY = 8;
    int e_list[EARLY], e_rows[EARLY];
#pragma unroll
    for (int i = 0; i < EARLY; ++i) {
        const int chunk = tid + kPlanThreads * i;
        e_list[i] = chunk < p.n_chunks ? p.chunk_list[chunk] : 0;
        e_rows[i] = chunk < p.n_chunks ? p.chunk_rows[chunk] : 0;
    }
    const int own = c0 + tid;
    const int o_list = own < c1 ? p.chunk_list[own] : 0, o_rows = own < c1 ? p.chunk_rows[own] : 0, o_row0 = own < c1 ? p.chunk_trow0[own] : 0;
    long long cand = 0;
    {
        constexpr int CPT = kIvfFastNlist / kPlanThreads;  // counters per thread: loaded together, then stored
        int n[CPT], len[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + kPlanThreads * i;
            n[i] = c < p.nlist ? p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)c * kIvfWideCntStride] : 0;
            len[i] = (p.cand_count && slice == 0 && c < p.nlist) ? p.offsets[c + 1] - p.offsets[c] : 0;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + kPlanThreads * i;
            if (c < p.nlist) {
                const int nq = min(n[i], kIvfWideQ);
                cnt_s[c] = nq;
                // the scan takes a list's slot table 16 entries at a time without looking at the count: the last block
                // is filled up with the dummy slot (one workgroup does it; the entries are stale otherwise)
                if (slice == 0)
                    for (int sl = nq; sl < ((nq + 15) & ~15); ++sl) p.lq[((int64_t)sb * p.nlist + c) * kIvfWideQ + sl] = kIvfWideQ * kDim;
            }
            cand += (long long)min(n[i], kIvfWideQ) * len[i];
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

// Bounds and plan in ONE launch (both need the pick kernel's output only and take about 10 us each: side by side instead
// of one after the other).  The first n_plan * n_sb workgroups plan, the rest compute bounds, two queries each.
__global__ __launch_bounds__(256) void ivf_tau_plan_kernel(const IvfWideParams p, const int n_plan, const int n_sb) {
    const int wg = blockIdx.x;
    if (wg < n_plan * n_sb) ivf_plan_body(p, wg / n_plan, wg % n_plan, n_plan);
    else ivf_tau_body(p, wg - n_plan * n_sb);
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
    const int b0 = sb * kIvfWideBatches, b1 = min(p.n_batches, b0 + kIvfWideBatches);
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
        const bool live = s < nslots && (s & 31) < p.B && p.slow[qg] == 0;  // a query without a bound goes through the slow path only
        const float t0 = live ? p.tau[qg] : -VS_INF;
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
#pragma unroll 1
        for (int t = 0; t < kIvfWideTiles; ++t) {
            if (r0 + 16 * t >= r_end) break;  // wave-uniform
            const int row = min(r0 + 16 * t + r, r_end - 1);
            f32x4 a[8];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) a[c8] = *reinterpret_cast<const f32x4*>(p.vecs + (int64_t)row * kDim + 16 * c8 + 4 * g);
            const f32x4 bn = *reinterpret_cast<const f32x4_u*>(p.vnorm + r0 + 16 * t + 4 * g);
            for (int cb = 0; cb < nq; cb += 16) {
                const int sq = cb + r;
                const bool live = sq < nq;
                const int ql = live ? lqc[sq] >> 7 : 0;  // (the table holds slot * 128)
                const int qg = qbase + ql;
                const float* qsrc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.q) + (long long)(qg >> 5) * p.q_batch_bytes) + (qg & 31) * kDim;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const f32x4 qf = *reinterpret_cast<const f32x4*>(qsrc + 16 * c8 + 4 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c8][i], qf[i], acc, 0, 0, 0);
                }
                const float qn = qn_s[ql];
                const float tq = live ? tau_s[ql] : -VS_INF;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = p.metric ? -acc[j] : fmaf(-2.0f, acc[j], qn + bn[j]);
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
    }
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

hipError_t launch_ivf_wide_bounds_plan(const IvfWideParams& p, hipStream_t s) {
    if (p.nlist > kIvfFastNlist || p.nprobe > kIvfMaxProbe || p.k > 16) return hipErrorInvalidValue;
    const int n_sb = (p.n_batches + kIvfWideBatches - 1) / kIvfWideBatches;
    const int n_plan = std::max(4, 16 / n_sb);  // (every planning workgroup reads all pair counters, a cache line each)
    hipLaunchKernelGGL(ivf_tau_plan_kernel, dim3(n_plan * n_sb + (p.n_batches * kMaxBatch + 1) / 2), dim3(256), 0, s, p, n_plan, n_sb);
    return hipGetLastError();
}

hipError_t launch_ivf_wide_scan(const IvfWideParams& p, int num_cus, hipStream_t s) {
    if (p.nlist > kIvfFastNlist || p.nprobe > kIvfMaxProbe || p.k > 16) return hipErrorInvalidValue;
    const int n_sb = (p.n_batches + kIvfWideBatches - 1) / kIvfWideBatches;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_scan_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kIvfWideLds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
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
    else merge_compact_body(m, L, cd, ci);
    // Last kernel of the launch group: it leaves the group's counters zeroed for the next group (no memset launch per
    // group).  Every workgroup clears what belongs to its query and a share of the lists' pair counters.
    __syncthreads();
#ifdef VS_STAMPS
    if (p.diag & 128) return;  // diagnostics read the counters afterwards (the API then memsets before every group)
#endif
    const int tid = threadIdx.x;
    if (tid < p.sink.nsub) p.sink.cnt[(int64_t)tid * p.sink.cnt_sub_stride + qg] = 0;
    if (tid == 0) p.slow[qg] = 0;
    const int sb = (q / p.B) / kIvfWideBatches;
    for (int c = q + tid * (int)gridDim.x; c < p.nlist; c += 256 * (int)gridDim.x)
        p.zero[sb * ivf_wide_plan_words(p.nlist) + (int64_t)c * kIvfWideCntStride] = 0;
    // (the words every workgroup reads: `overflow` is cleared by the next group's coarse kernel, the batches' "not byte
    // valued" flags are written as 0 or 1 there; a counter of finished workgroups here would be one contended atomic
    // per query and cost more than the memset it saves -- measured)
}

hipError_t launch_ivf_wide_rank(const MergeParams& m, int64_t stride_g, int64_t stride_q, const IvfWideParams& p, hipStream_t s) {
    if (m.kout < 1 || m.G < 1 || m.nq != p.n_batches * p.B || (int64_t)m.G * m.kin > kCompactCap || m.q_group_out != p.B || m.q_group_in != kMaxBatch)
        return hipErrorInvalidValue;
    MergeLayout L{stride_g, stride_q};
    hipLaunchKernelGGL(ivf_wide_rank_kernel, dim3(m.nq), dim3(256), 0, s, m, L, p);
    return hipGetLastError();
}

hipError_t launch_ivf_select(const IvfSelectParams& p, int B, hipStream_t s, int n_batches) {
    if (p.k > 16 || p.nprobe > 256) return hipErrorInvalidValue;
    IvfSelectParams pp = p;
    pp.split = p.bkt ? 1 : kSelSplit;  // with per-block minima one workgroup reads the few blocks under the bound
    if (!p.slotmin && !p.bkt) hipLaunchKernelGGL(ivf_bound_kernel, dim3(B * kSelSplit, n_batches), dim3(256), 0, s, pp);
    hipLaunchKernelGGL(ivf_select_kernel, dim3(B * pp.split, n_batches), dim3(256), 0, s, pp);
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
