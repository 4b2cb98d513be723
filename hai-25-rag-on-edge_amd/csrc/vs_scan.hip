// vs_scan.hip   -- hand-written gfx950 (CDNA4) kernels of the distance + top-k hot path.
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
//   ivf_coarse_mfma_kernel, ivf_pick_kernel, ivf_bounds_plan_kernel, ivf_tau_combine_kernel, ivf_scan_wide_kernel, ivf_wide_rank4_kernel :
//                          IVFIndex::searchBatch (IVFIndex.cpp:640-859) as a list-major pipeline over launch groups of up
//                          to 256 batches (one pass over the probed lists per super-batch of 32 batches);
//                          pick_probes_kernel / ivf_scan_kernel: the per-batch fallback (nlist > 4096, k > 15).
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
#include "vs_sink.h"

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

}  // namespace vs
