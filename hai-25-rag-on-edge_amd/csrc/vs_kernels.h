// vs_kernels.h -- host-callable launchers of the gfx950 kernels (internal to libvsearch_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

constexpr int kDim = 128;          // SIFT descriptor length; the scan kernels are specialised for it
constexpr int kTileRows = 16;      // base rows per MFMA tile (v_mfma_f32_16x16x4_f32, base = A operand)
constexpr int kScanThreads = 512;  // 8 waves: 2 per SIMD
constexpr int kScanWaves = 8;
constexpr int kMaxBatch = 32;      // queries per scan pass (two 16-query MFMA column blocks)
constexpr int kScanPadRows = 64;  // every row array handed to the scan is allocated with this many spare rows (tile DMAs are not clamped)
constexpr int kSlotStride = 256;   // threshold-exchange slots: [32 queries][256 workgroups]

enum ScanMode { kModeTopK = 0, kModeStore = 1, kModeAssign = 2, kModeFilter = 3 };

struct ScanParams {
    const float* base;       // [n_rows][128] row-major (the flat fvecs payload, cpu_baseline.cpp:48-49)
    const float* bnorm;      // [n_rows (+64 pad)] squared norms (cpu_baseline.cpp:116-125)
    const int8_t* base_u8;   // int8 path: [n_rows][128] bytes (x - 128), or nullptr = fp32 path
    const int32_t* rterm;    // int8 path: [n_rows (+64 pad)] ||b||^2 - 256 * sum(b - 128)
    int32_t* invalid;        // int8 path: [n_batches], set to 1 for a batch whose queries are not integers in [0, 255]
    const float* q;          // [n_batches][nq_valid][128] raw queries; rows up to 32 are zero-padded in-kernel (main.cpp:206-211)
    int n_batches;           // query batches served by this one persistent launch (1 for kModeStore)
    int64_t q_batch_stride;  // floats between consecutive batches in q
    float* slots_cur;        // [n_batches][32][kSlotStride] per-workgroup minima (pre-set to +inf) or nullptr = no exchange
    const float* tau0;       // [n_batches][32] bounds computed beforehand by launch_seed (then no exchange, no warm-up), or nullptr
    int k1;                  // the exchange bounds the k1-th best distance; also entries kept per partial list
    int* dbg;                // optional debug counters [grid][16]
    const int32_t* run_if;   // optional [1]: the launch does nothing unless *run_if != 0 (fallback behind a streaming scan)
    int64_t row_begin;       // multiple of 16
    int64_t row_end;         // exclusive
    int tiles_per_wg;
    int metric;              // 0 = L2, 1 = IP (scores negated so that smallest wins)
    int32_t id_offset;       // added to the row number
    int nq_valid;            // queries beyond this are padding
    // kModeTopK outputs: per-workgroup sorted partial lists, query-major (ranked by merge_compact_kernel)
    float* part_d;           // [n_batches][32][kSlotStride][KCAP]
    int32_t* part_i;
    // kModeStore output
    float* store;            // [nq_valid][store_ld], column = row - row_begin
    int64_t store_ld;
    // kModeAssign (k-means): running nearest "query" (centroid) per base row, id = assign_base + batch*32 + column
    float* best_d;           // [n_rows] (pre-set to +inf)
    int32_t* best_i;         // [n_rows] (pre-set to -1)
    int assign_base;
    // kModeFilter (tie resolver): every (row, dist) with dist < tau0[query] is appended to the query's candidate list --
    // the rows select_topk (cpu_baseline.cpp:139-150) can still act on once its buffer maximum is below tau0
    int32_t* f_cnt;          // [32] entries appended per query (pre-set to 0; may exceed f_cap: overflow)
    int32_t* f_row;          // [32][f_cap] row numbers (relative to the index, unordered)
    float* f_d;              // [32][f_cap]
    int f_cap;
};

// Brute-force / coarse scan.  kcap in {8, 16}; nqh = 1 (<=16 queries) or 2.
hipError_t launch_scan(const ScanParams& p, int grid, int kcap, int nqh, int mode, hipStream_t s);

// Bounds for a multi-batch scan, computed up front on a sample of the rows: kSeedWaves 16-row tiles spread evenly
// over the shard, scored against every batch; tau0[batch][q] = next_up of the k1-th smallest of 64 group minima
// (groups of 32 tiles) -- an upper bound of the k1-th best distance, because k1 distinct rows are at least that
// close.  Three small launches per call.
constexpr int kSeedWaves = 2048;  // sample tiles
struct SeedParams {
    const float* base;       // [n_rows (+pad)][128]
    const float* bnorm;
    // optional (launch_seed_sample): the kSeedWaves sample tiles once more, compact and in MFMA A-fragment order -- the seed
    // reads 1 KB per load instruction out of 4 MB (bytes) / 16 MB (fp32) that stay in L2, instead of 64 pieces per
    // instruction scattered over the whole shard (which made the address unit, not the arithmetic, set its 24 us)
    const float* sample_f32;     // [kSeedWaves][8][64][4]   fp32 tile: chunk c, lane (r, g) -> row r, floats 16 c + 4 g ..
    const float* sample_bnorm;   // [kSeedWaves][16]
    const int8_t* sample_u8;     // [kSeedWaves][2][64][16]  byte tile: half, lane (r, g) -> row r, bytes 64 half + 16 g ..
    const int32_t* sample_rterm; // [kSeedWaves][16]
    const int8_t* base_u8;   // optional exact int8 copy (x - 128) + row terms: the seed then runs on v_mfma_i32_16x16x64_i8
    const int32_t* rterm;    //   (16x fewer MFMA cycles) for batches whose queries are byte valued too (q8 / qterm / invalid below are then required)
    int64_t n_rows;
    const float* q;          // [n_batches][nq_valid][128]
    int n_batches;
    int64_t q_batch_stride;
    int nq_valid;
    int metric;
    int k1;
    float* qnorm;            // scratch [n_batches][32]
    float* wmin;             // scratch [n_batches][64 groups][32]
    float* tau0;             // out [n_batches][32]
    int32_t* zero;           // optional: zero_words words cleared by the query-preparation launch (except [16, 48) when `invalid` is set)
    int zero_words;
    // optional outputs of the query-preparation launch for the wide int8 scan (launch_scan_i8_wide): the queries as
    // bytes (x - 128), qterm = ||q||^2 - 256 sum(q - 128) - 2 * 128^3 per query, invalid[batch] = 1 when a query of the
    // batch is not an integer in [0, 255] (pre-set to 0)
    int8_t* q8;              // [n_batches][32][128]
    int32_t* qterm;          // [n_batches][32]
    int32_t* invalid;        // [n_batches]
    // optional: the queries in MFMA B-fragment order for the fp32 streaming scan, qfrag[batch][h][c][lane] = the four floats
    // Q[16 h + (lane & 15)][16 c + 4 (lane >> 4) ..] (zeros for padding queries): a wave's operand load is then 1 KB in one
    // piece (as fragments straight from the row-major queries it is 64 pieces 512 bytes apart per instruction, and eight
    // waves entering a pass kept a CU's address unit busy for 8 us)
    float* qfrag;            // [n_batches][2][8][64][4]
    int8_t* q8frag;          // [n_batches][2][2][64][16] the byte queries likewise: (h, half, lane) -> bytes [64 half + 16 (lane >> 4) ..] of query 16 h + (lane & 15)
};
hipError_t launch_seed(const SeedParams& p, hipStream_t s);
// gathers the sample tiles of a shard into the compact arrays above (once, at index creation); u8 outputs optional
hipError_t launch_seed_sample(const float* base, const float* bnorm, const int8_t* base_u8, const int32_t* rterm, int64_t n_rows,
                              float* sample_f32, float* sample_bnorm, int8_t* sample_u8, int32_t* sample_rterm, hipStream_t s);

// Wide exact-int8 brute-force scan: ONE pass over the byte rows serves NQH 16-query column blocks = NQH / bpb query
// batches (bpb = 1 for batches of <= 16 queries, else 2), with the bounds of launch_seed in force from the first tile.
// No lane lists, no workgroup merge: a distance under its query's bound is appended to that query's candidate list in
// global memory (a few hundred per query per million rows); launch_merge_flat ranks the lists.  Waves are independent:
// tiles of ALL passes are handed out through one monotonic ticket counter per workgroup, so nothing synchronises
// between batches.
// Where the streaming scans put what they find: wave-private buffers filled with plain stores (positions from a wave
// ballot -- a returning atomic in the tile loop would have to be waited for with vmcnt(0), i.e. drain the tile queue on
// every hit), binned into per-query lists by a small launch after the scan.
struct CandSink {
    int4* wbuf;              // [grid * 8][wcap] x (query = batch * 32 + q, dist bits, id, 0)
    int wcap;
    int32_t* overflow;       // [1] (pre-set to 0): a wave buffer or a query list overflowed -> the launch's fallback kernels
                             //     (the per-batch scan + its merge, enqueued behind with run_if = overflow) produce the result
    // per-query lists, split into nsub sub-lists (wave buffer w appends to sub-list w % nsub) so that the appending
    // atomics of one query spread over nsub counters
    int32_t* cnt;            // [n_batches][32][nsub] entries per sub-list (pre-set to 0)
    float* cand_d;           // [n_batches][32][nsub][cap]
    int32_t* cand_i;
    int cap;                 // per sub-list
    int nsub;
    int32_t* slow;           // optional [queries]: a query whose lists overflow is marked here instead of raising `overflow`
    // XCD-local sub-lists (0 = off): sub-list = xcd * xcd_subs + hash % xcd_subs (nsub = 8 * xcd_subs) and the counters
    // are sub-list major, cnt[sub * cnt_sub_stride + query]: every counter line and every list is then written by the
    // workgroups of ONE XCD and stays in that XCD's L2 (lines appended to from several XCDs travel between the L2s
    // on every atomic and every store)
    int xcd_subs;
    int cnt_sub_stride;
};

struct WideParams {
    const int8_t* base_u8;   // [n_rows (+64)][128] bytes (x - 128)
    const int32_t* rterm;    // [n_rows + 64]
    int64_t n_rows;
    const int8_t* q8;        // [n_batches][32][128] from launch_seed
    const int8_t* q8frag;    // [n_batches][2][2][64][16] the same in B-fragment order (SeedParams::q8frag)
    const int32_t* qterm;    // [n_batches][32]
    const float* tau0;       // [n_batches][32]
    const int32_t* invalid;  // [n_batches]
    int n_batches, nq_valid, bpb;
    int32_t id_offset;
    CandSink sink;
};
hipError_t launch_scan_i8_wide(const WideParams& p, int grid, int nqh, hipStream_t s);  // nqh in {4, 8}

// Streaming fp32 scan: the arithmetic of scan_kernel (v_mfma_f32_16x16x4_f32 chain in k order, fma(-2, dot, qn + bn)
// epilogue: bit-identical distances), one batch per pass, but organised like the wide int8 scan: bounds from
// launch_seed, queries and their norms straight from global memory, survivors to the CandSink, tiles of all batches
// handed out through one monotonic ticket per workgroup -- no barrier, no workgroup merge, no per-batch prologue.
struct StreamParams {
    const float* base;       // [n_rows (+64)][128]
    const float* bnorm;      // [n_rows + 64]
    int64_t n_rows;
    const float* q;          // [n_batches][nq_valid][128] raw queries
    int64_t q_batch_stride;
    const float* qfrag;      // [n_batches][2][8][64][4] the same queries in B-fragment order (launch_seed)
    const float* qnorm;      // [n_batches][32] from launch_seed
    const float* tau0;       // [n_batches][32]
    int n_batches, nq_valid, metric;
    int32_t id_offset;
    CandSink sink;
    int batches_per_pass;    // 1 (HBM bound) or 2 (two batches share a pass over the rows: MFMA bound)
};
hipError_t launch_scan_f32_stream(const StreamParams& p, int grid, hipStream_t s);

// Single-call scan (vs_scan_one.hip): one launch per batch of <= 32 queries, no seed launches, no merge launch -- a lane
// keeps its own sorted lists, a workgroup ranks them into one partial list per query, the workgroup that arrives last
// merges the partial lists.  fp32 rows; k1 <= 16.
struct OneParams {
    const float* base;       // [n_rows (+64)][128]
    const float* bnorm;      // [n_rows + 64]
    int64_t n_rows;
    const float* q;          // [nq_valid][128] raw queries
    int nq_valid, k1, metric;
    int reverse;             // walk the base from its end (calls alternate: see scan_one_kernel)
    int32_t id_offset;
    float* part_d;           // scratch [32][grid][k1]
    int32_t* part_i;
    int32_t* done;           // [1] arrival counter: 0 at launch, left 0 by the last workgroup
    float* out_d;            // [nq_valid][k1] ascending (dist, id), padded with (+inf, -1)
    int32_t* out_i;
    int32_t* flags;          // optional [nq_valid]: 1 = two equal distances among the k1
    int* dbg;                // diagnostic builds (-DVS_STAMPS) only: time stamps [grid][16]
};
int scan_one_grid(int64_t n_rows, int num_cus);
hipError_t launch_scan_one(const OneParams& p, int grid, hipStream_t s);

// Cross-workgroup merge of sorted partial lists -> [nq][kout] + tie flags (+ seed thresholds).
struct MergeParams {
    const float* part_d;     // [G][nq_stride][kin]
    const int32_t* part_i;
    int G;
    int nq_stride;           // queries per partial block (32 for scan partials)
    int kin;
    int nq;                  // queries to produce
    int kout;                // entries to write per query (<= kin)
    float* out_d;            // [nq][kout] or nullptr
    int32_t* out_i;
    int32_t* flags;          // [nq] or nullptr: adjacent equal distances among the first kout
    float* tau_out;          // [32] or nullptr: nextafter(kout-th best) used as scan seed
    const int32_t* id_map;   // optional: out id = id_map[id] (IVF reorder_to_original)
    int q_group_out, q_group_in;  // output query q reads input query (q / out) * in + q % out (0 = identity)
    const int32_t* invalid;  // optional [nq / q_group_out]: batches skipped by the int8 scan -> flags = 2
    int flag_empty;          // flags = 2 for a query with no finite entry at all (cross-GPU merge: every shard skipped its batch)
    const int32_t* shard_flags;  // optional (cross-GPU merge): shard g's own flag of output query q at g * shard_flags_stride + q;
    int64_t shard_flags_stride;  //   a shard that skipped the query's batch (flag 2) makes the merged flag 2: its rows are missing
    const int32_t* run_if;   // optional [1] with run_mode: 1 = run only if *run_if != 0, 2 = only if *run_if == 0 (the other
    int run_mode;            //   launch of the pair writes the outputs)
    const int32_t* flat_len; // optional [queries][G]: list (q, g) holds flat_len[q * G + g] <= kin UNSORTED candidates
    int flat_len_sub_stride; // != 0: the lengths are list major instead, flat_len[g * flat_len_sub_stride + q]
    int32_t* dbg;            // diagnostic builds (-DVS_STAMPS) only
};
hipError_t launch_merge(const MergeParams& p, hipStream_t s);  // scan-partial layout [G][nq_stride][kin]
// general layout: entry (g, q, j) at g*stride_g + q*stride_q + j
hipError_t launch_merge_layout(const MergeParams& p, int64_t stride_g, int64_t stride_q, hipStream_t s);

// One Lloyd update: deterministic fixed-point cluster sums -> new centroids; shift[c] = ||new - old||^2.
hipError_t launch_kmeans_update(const float* x, const int32_t* assign, int64_t rows, int nlist, float* cents,
                                unsigned long long* acc, int32_t* counts, double* shift, hipStream_t s);

// k-means++ seeding (Arthur & Vassilvitskii D^2 sampling; sklearn's KMeans default init, create_ivf_model_reordered.py:
// 97-103).  Step c: d2[i] = min(d2[i], ||x_i - cents[c-1]||^2) with per-block sums, then the row where the running sum
// of d2 passes u * total becomes cents[c].  Two launches per centre, no host round trip.
hipError_t launch_kpp_step(const float* x, const float* xnorm, int64_t rows, float* cents, int c, float* d2, double* block_sums,
                           int n_blocks, double u, hipStream_t s);
constexpr int kKppBlockRows = 1024;

// ||v||^2 per row in the reference's AVX2 summation order (cpu_baseline.cpp:95-114).
hipError_t launch_row_sqnorm(const float* v, int64_t rows, int dim, float* out, hipStream_t s);

// ---- IVF ----
// Per query: the nprobe nearest centroids (ascending (dist, id)) out of a [B][ld] score matrix.
hipError_t launch_pick_probes(const float* scores, int64_t ld, int B, int nlist, int nprobe,
                              int32_t* probes /*[B][nprobe]*/, hipStream_t s);

constexpr int kIvfMaxProbe = 256;

// Multi-batch IVF launches: blockIdx.y = batch.  Per-batch pointers point at batch 0's copy; batch y's coarse scores lie
// y * slab bytes, its probes y * probes bytes, its queries y * q bytes further on.
struct IvfMulti {
    long long slab, probes, q;
};

// What the coarse / pick kernels of the wide pipeline do beside scores and probes (every pointer optional)
struct IvfGroup {
    IvfMulti mb;                     // batch y's slab (coarse scores, probes) lies mb.slab bytes, its queries mb.q bytes further on
    int sb_batches;                  // batches per super-batch (<= kIvfWideBatches)
    // the coarse kernel also prepares the queries for the int8 paths (block x = 0 of every batch)
    float* w_qnorm;                  // [n_batches][32] ||q||^2
    int8_t* w_q8;                    // [n_batches][32][128] queries as bytes (x - 128)
    int32_t* w_qterm;                // [n_batches][32]
    int32_t* w_invalid;              // [n_batches]: a query of the batch is not byte valued (written as 0 or 1)
    int32_t* w_overflow;             // [1]: cleared (the wide pipeline's candidate-buffer overflow word)
    int32_t* w_glist;                // [1]: cleared (count of the ranking's left-over list, see launch_ivf_wide_rank)
    // the pick kernel also enters every (query, probe) pair in its list's slot table
    int32_t* w_cnt;                  // [n_sb][ivf_wide_plan_words] pair counters (pre-set to 0), list c's at word c * kIvfWideCntStride
    int32_t* w_lq;                   // [n_sb][nlist][w_q] slots: 128 * (slot of the query in its super-batch)
    int w_q;
    // ... and enters the query in the bound tables of its two nearest lists that hold rows (ivf_bounds_list_body)
    int32_t* w_tq;                   // [nlist][w_tq_cap] entries: query of the launch group | segment << 16
    int w_tq_cap;
    int32_t* w_tcnt;                 // list c's entry counter at word c * kIvfWideCntStride + 1 (pre-set to 0)
    int32_t* w_nseg;                 // [n_batches][32] segments the query has (0..kBoundSegs)
    const int32_t* t_offsets;        // [nlist + 1] extents of the rows the bounds are taken from
    int32_t* dbg;                    // diagnostic builds (-DVS_STAMPS) only: time stamps of the pick kernel
};

// Coarse stage: Q x C^T + L2 epilogue on MFMA into scores [n_batches][32][ld] (ld >= nlist rounded up to 64; batch y's
// copy lies grp.mb.slab bytes further on), then the nprobe nearest lists per query (nlist <= kIvfFastNlist).
constexpr int kIvfFastNlist = 4096;
hipError_t launch_ivf_coarse_pick(const float* q, int B, const float* cents, const float* cnorm, int nlist, int nprobe,
                                  int metric, float* scores, int ld, int32_t* probes, const IvfGroup& grp, hipStream_t s,
                                  int n_batches = 1);
hipError_t launch_ivf_prep_queries(const float* q, int B, const float* cents, const float* cnorm, int nlist, const IvfGroup& grp,
                                   hipStream_t s, int n_batches);
hipError_t launch_ivf_fill(const int32_t* gathered, long long blk_words, int B, int nprobe, int nlist, const int32_t* offsets,
                           int32_t* probes_out, float* tau_out, int32_t* slow_out, const IvfGroup& grp, hipStream_t s, int n_batches);

// ---- wide IVF pipeline: launch groups are cut into super-batches of kIvfWideBatches batches (<= 1024 queries) that
// share ONE list-major pass: a list probed by any of them is read once and scored against all its queries (MFMA
// column blocks of 16).  Bounds first (ivf_bounds_plan_kernel, list-major too: the k-th best among the first rows of the
// query's two nearest lists), survivors to a CandSink, ranking by a wave per query -- no candidate-score arrays, no
// selection kernel.
constexpr int kIvfWideBatches = 32;  // batches per super-batch at most: every resident list is read once per 1024 queries
constexpr int kIvfWideQ = kIvfWideBatches * kMaxBatch;  // query slots per super-batch
#ifndef VS_BOUND_SEGS
#define VS_BOUND_SEGS 2
#endif
constexpr int kBoundSegs = VS_BOUND_SEGS;               // lists a query takes its bound from (list-major bounds; <= 4)
#ifndef VS_TAU_ROWS
#define VS_TAU_ROWS 256
#endif
constexpr int kIvfTauRows = VS_TAU_ROWS;                 // rows of a list that seed its queries' bounds (a multiple of 64)
#ifndef VS_WIDE_UNIT
#define VS_WIDE_UNIT 32
#endif
constexpr int kIvfWideUnit = VS_WIDE_UNIT;  // rows per unit of the wide scan's plan (lists are padded to multiples of it)
constexpr int kIvfWideCntStride = 32;
constexpr long long ivf_wide_plan_words(int nlist) { return (long long)nlist * kIvfWideCntStride + 16; }
struct IvfWideParams {
    const float* vecs;        // [n_rows (+64)][128] cluster-reordered
    const float* vnorm;       // [n_rows + 64]
    const int8_t* vecs_u8;    // optional exact int8 copy (x - 128) + row terms
    const int32_t* rterm;
    // the scan's own copy: lists padded to multiples of 32 rows ("padded rows"), 16-row tiles [half][chunk][row][16 B]
    const int8_t* vecs_t8;
    const int32_t* nrh_t;     // [padded rows + 64] -(rterm >> 1): the MFMA C operand
    const int32_t* rterm_t;
    const int32_t* tdelta;    // [nlist] padded row - row (null: 0)
    const int32_t* chunk_trow0;  // [n_chunks] first padded row of a chunk: the plan's records and the candidates are padded rows
    const int32_t* offsets;   // [nlist+1] local list offsets (lists not resident here are empty)
    const int32_t* chunk_list;
    const int32_t* chunk_row0;
    const int32_t* chunk_rows;
    int n_chunks, nlist, nprobe, k, metric;
    const float* q;           // batch b's [B][128] at (char*)q + b * q_batch_bytes
    long long q_batch_bytes;
    int n_batches, B;
    int sb_batches;           // batches per super-batch (<= kIvfWideBatches): super-batch sb = batches [sb * sb_batches, ...)
    const float* qnorm;       // [n_batches][32]   | from ivf_coarse_mfma_kernel
    const int8_t* q8;         // [n_batches][32][128]
    const int32_t* qterm;     // [n_batches][32]
    const int32_t* invalid;   // [n_batches] (pre-set to 0) a query of the batch is not byte valued
    const int32_t* probes;    // batch b's [B][nprobe] at (char*)probes + b * probes_batch_bytes
    long long probes_batch_bytes;
    int32_t* lq;              // [n_sb][nlist][kIvfWideQ] slot tables: 128 * ((batch % 32) * 32 + q) of the queries probing each list
                              //   (= where the scan's LDS keeps the query), padded to a multiple of 16 entries with the dummy slot
    int32_t* zero;            // [n_sb][ivf_wide_plan_words(nlist)] (pre-set to 0): list c's pair counter at word c * kIvfWideCntStride
                              //   (a 128-byte line each: the pick kernel's atomics come from all XCDs), the super-batch's record
                              //   count at word nlist * kIvfWideCntStride
    unsigned long long* cand_count;  // optional: += rows the probed lists hold per (query, probe) pair
    int32_t* units;           // [n_sb][units_cap][4] records (first row, chunk end, list, first slot | end slot << 16)
    long long units_sb_stride;  // in int32 (= 4 units_cap)
    int units_cap;            // records per super-batch the plan may hold (>= the index's units: the unsplit plan fits)
    // bounds, list-major: entries of the pick kernel (see IvfGroup), entry counter of list c at zero[c * kIvfWideCntStride + 1]
    const int32_t* tq;        // [nlist][tq_cap]
    int tq_cap;
    float* tk;                // [n_batches][32][kBoundSegs][16] the k smallest distances per (query, segment), ascending
    const int32_t* nseg;      // [n_batches][32]
    int tau_inline;           // the scan works a query's bound out of tk / nseg itself (no ivf_tau_combine_kernel launch, tau unused)
    float* tau;               // [n_batches][32] bounds
    int32_t* slow;            // [n_batches][32] (pre-set to 0): no bound could be had -> exact slow path
    CandSink sink;
    float* out_d;             // [n_batches][B][k]   (slow path writes here directly)
    int32_t* out_i;
    const int32_t* id_map;    // reorder_to_original (local), by row: the slow path's
    int32_t* dbg;             // diagnostic builds (-DVS_STAMPS) only: per-workgroup time stamps, dbg[0..] see ivf_scan_wide_kernel
    int diag;                 // diagnostic builds only: bit 0 no column blocks, bit 1 every unit = the wave's first, bit 2 no binning
};
hipError_t launch_ivf_wide_bounds_plan(const IvfWideParams& p, hipStream_t s, int what = 3);  // bounds (what & 1) and the plan (what & 2), one launch
hipError_t launch_ivf_wide_scan(const IvfWideParams& p, int num_cus, hipStream_t s);  // the list-major scan
int ivf_wide_grid_x(int num_cus, int n_sb);  // grid.x of the scan
int ivf_wide_waves(int num_cus, int n_sb);   // its waves = candidate buffers
// merge of the candidate lists (flat layout, see launch_merge_layout) or the exact slow path, per query
// glist (optional): 1 + queries int32 words; with it launches of more than 2048 queries rank one query per wave and leave the
// few that need a workgroup to a second launch (the list's count, word 0, is cleared by the next group's coarse / prep kernel)
hipError_t launch_ivf_wide_rank(const MergeParams& m, int64_t stride_g, int64_t stride_q, const IvfWideParams& p, hipStream_t s,
                                int32_t* glist = nullptr);

struct IvfScanParams {
    const float* vecs;        // [n_rows][128] cluster-reordered (vectors_reordered.npy)
    const float* vnorm;       // [n_rows]
    const int32_t* offsets;   // [nlist+1] (cluster_offsets.npy)
    const uint8_t* owned;     // [nlist] 1 = this shard scans the list, or nullptr = all
    const float* q;           // [B][128] raw queries (norms are computed in-kernel)
    const int32_t* probes;    // [B][nprobe]
    int B, nprobe, kcap;
    int metric;
    float* part_d;            // [B][nprobe][kcap]
    int32_t* part_i;          // reordered row positions
    unsigned long long* cand_count;  // total rows scanned (IVFIndex::searchBatch return value)
};
hipError_t launch_ivf_scan(const IvfScanParams& p, hipStream_t s);

}  // namespace vs
