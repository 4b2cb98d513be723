// vs_api.hip -- C ABI of libvsearch_hip.so: index objects, device memory, launch orchestration.
//
// One vs_index owns: the base (or cluster-reordered) vectors and their squared norms in HBM,
// a small scratch arena (padded queries, thresholds, per-workgroup partial lists, result
// staging) and a HIP stream.  Nothing here computes distances on the host: without a HIP device
// every create/search call fails (VS_ERR_DEVICE).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vsearch.h"
#include "vs_host.h"
#include "vs_kernels.h"

using vs::set_error;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                              \
            return VS_ERR_DEVICE;                                                                      \
        }                                                                                              \
    } while (0)

namespace {

constexpr int kMaxEvents = 8192;
constexpr int kKcapMax = 16;       // largest per-lane list the scan kernels are compiled for
constexpr int kMaxNprobe = 256;
constexpr int kMaxLanes = 4;
constexpr int kMaxMulti = 32;  // batches per persistent scan launch
constexpr int kOneMaxBatches = 4;  // calls of fewer batches take one single-call scan launch per batch (fp32 rows)
constexpr int kOneMaxQueries = 16;  // ... when a batch holds at most this many queries
constexpr int kPairMinTiles = 96;   // tiles per workgroup and pass from which the fp32 streaming scan pairs batches
constexpr int kIvfGroupDefault = 256; // batches per launch group of an unsharded IVF index (VSEARCH_IVF_GROUP): 46 us per 1024 queries against 79 with groups of 32
constexpr int kIvfGroupMax = 256;     // ... at most (8 super-batches of 32): also the group of an index sharded 8 ways
constexpr int kIvfShardMaxWorld = 16; // ranks the cluster-sharded pipeline is compiled for (one super-batch per rank)
constexpr int64_t kIvfHostChunk = 4 * 32 * 32;  // queries per chunk of the host-buffer IVF call (one upload, one download)
constexpr int kWideLanesMax = 4;  // streams (and scratch sets) the launch groups of one wide IVF call may be dealt to
constexpr int kWideWaveCap = 1024;  // entries per wave buffer of the wide int8 scan (expected fill: about 100 per launch)
constexpr int kWideSub = 16;      // sub-lists per query of the streaming scans' candidate lists
constexpr int kWideCap = 128;     // entries per sub-list (2048 per query; more: the per-batch scan behind takes over)
constexpr int kIvfWideWaveCap = 512;  // entries per wave buffer of the wide IVF scan (expected fill: a few dozen per group)
constexpr int kIvfWideSubCap = 256;   // entries per candidate sub-list (16 per query: 4096 candidates)
constexpr int kTieDense = 4096;   // rows whose distances the tie resolver takes densely
constexpr int kTieCap = 8192;     // candidate slots per flagged query (more: full-row fallback)

struct ProfSlot {
    std::vector<hipEvent_t> ev;  // pairs
    int used = 0;
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct vs_index {
    int kind = 0;  // 0 = brute force, 1 = IVF
    int device = 0;
    int dim = 0;
    int metric = VS_METRIC_L2;
    int64_t n_rows = 0;   // rows resident on this GPU
    int64_t n_total = 0;  // rows of the whole (unsharded) index
    int64_t id_offset = 0;
    int batch = vs::kMaxBatch;
    int num_cus = 256;

    float* d_vecs = nullptr;   // [n_rows][128]
    float* d_norm = nullptr;   // [n_rows + 64]
    // int8 data path: only when every base value is an integer in [0, 255]
    int8_t* d_vecs_u8 = nullptr;   // [n_rows][128] bytes (x - 128)
    int32_t* d_rterm = nullptr;    // [n_rows + 64] ||b||^2 - 256 * sum(b - 128)
    // the seed's sample tiles, compact and in MFMA fragment order (SeedParams::sample_*; brute-force indexes)
    float* d_seed_f32 = nullptr;
    float* d_seed_bnorm = nullptr;
    int8_t* d_seed_u8 = nullptr;
    int32_t* d_seed_rterm = nullptr;
    // wide IVF scan: the byte rows once more, every list padded to a multiple of 32 rows ("padded rows") and stored as
    // 16-row MFMA tiles of 2 KB, [half of the row][16-byte chunk][row][16 bytes] -- a wave's A-operand load is then 1 KB
    // of consecutive bytes (from the row-major copy the same load touches 16 B in each of 64 places)
    int8_t* d_vecs_t8 = nullptr;   // [n_padded + 64 rows]
    int32_t* d_nrh_t = nullptr;    // [n_padded + 64] -(rterm >> 1) by padded row: the MFMA C operand
    int32_t* d_rterm_t = nullptr;  // [n_padded + 64]
    int32_t* d_r2o_t = nullptr;    // [n_padded + 64] padded row -> original id (-1: padding)
    int32_t* d_tdelta = nullptr;   // [nlist] padded row - row, per list
    int32_t* d_chunk_trow0 = nullptr;  // [n_chunks] first padded row of a chunk
    int32_t* d_invalid = nullptr;  // [kMaxMulti] batches the int8 scan had to skip
    int precision = 0;             // 0 = auto (int8 when possible), 1 = fp32, 2 = int8

    // IVF
    int nlist = 0;
    int rank = 0, world = 1;
    float* d_centroids = nullptr;  // [nlist][128]
    float* d_cnorm = nullptr;      // [nlist + 16]
    int32_t* d_offsets = nullptr;  // [nlist + 1] offsets into the LOCAL d_vecs (non-owned lists are empty)
    int32_t* d_r2o = nullptr;      // [n_rows] local position -> original id
    std::vector<int32_t> h_offsets_global;  // as loaded (for save / stats)
    double avg_cluster_size = 0;

    // scratch
    float* d_q = nullptr;        // staging for host queries [32][128]
    // Pipeline lanes: consecutive batches of a multi-batch call run on different internal streams
    // so that the start-up / tail of one scan overlaps the streaming phase of its neighbours
    // (inside one launch all workgroups go through those phases in lock-step and HBM idles).
    struct Lane {
        hipStream_t s = nullptr;
        hipEvent_t done_ev = nullptr;
        float* slots = nullptr;     // [kMaxMulti][32][kSlotStride] threshold-exchange slots, reset per launch
        int* done = nullptr;        // [kMaxMulti] arrival counters, reset per launch
        float* part_d = nullptr;    // [kMaxMulti][32][kSlotStride][16]
        int32_t* part_i = nullptr;
        float* seed_qnorm = nullptr; // [kMaxMulti][32]   scratch of launch_seed
        float* qfrag = nullptr;      // [kMaxMulti][2][8][64][4] queries in MFMA B-fragment order (fp32 streaming scan)
        int8_t* q8frag = nullptr;    // [kMaxMulti][2][2][64][16] byte queries in B-fragment order (wide int8 scan)
        float* seed_wmin = nullptr;  // [kMaxMulti][kSeedWaves][32]
        float* tau0 = nullptr;       // [kMaxMulti][32]   bounds of the current multi-batch launch
        // wide int8 scan (several batches per pass over the rows): prepared queries + per-query candidate lists
        int8_t* q8 = nullptr;        // [kMaxMulti][32][128]
        int32_t* qterm = nullptr;    // [kMaxMulti][32]
        int32_t* wcnt = nullptr;     // [16] (word 0: overflow) + [kMaxMulti][32][kWideSub]
        float* wcand_d = nullptr;    // [kMaxMulti][32][kWideCap]
        int32_t* wcand_i = nullptr;
        int4* wbuf = nullptr;        // [256 * 8][kWideWaveCap] wave-private candidate buffers of the scan
    };
    Lane lane[kMaxLanes];
    int n_lanes = 1;
    hipEvent_t fork = nullptr;
    float* d_out_d = nullptr;    // [32][64]
    int32_t* d_out_i = nullptr;
    int32_t* d_flags = nullptr;  // [32]
    float* d_scores = nullptr;   // IVF query-major fallback: coarse scores [32][nlist_pad]; tie fallback rows
    int64_t scores_cap = 0;      // floats
    int32_t* d_probes = nullptr; // [32][kMaxNprobe]
    float* d_ipart_d = nullptr;  // [32][kMaxNprobe][16]
    int32_t* d_ipart_i = nullptr;
    unsigned long long* d_cand = nullptr;
    // list-major IVF scan (grouped path)
    int32_t* d_chunk_list = nullptr;   // [n_chunks] (list, 1024-row chunk) work items over the resident lists
    int32_t* d_chunk_row0 = nullptr;
    int32_t* d_chunk_rows = nullptr;
    int ivf_gb = 32;                   // batches per launch group (multiple of 32) the wide pipeline's scratch is sized for
    int ivf_nsb = 1;                   // ... in at most this many super-batches (sharded: one per rank)
    int ivf_lanes = 2;                 // streams the launch groups of one device call are dealt to
    int64_t ivf_host_cap = 0;          // queries per chunk the host-buffer call's staging slots hold
    // sharded index: the first kIvfTauRows rows of EVERY list (resident or not), replicated on every rank: a query's bound
    // then comes from its two nearest lists wherever they live -- the bounds of the unsharded index (a bound from the
    // nearest RESIDENT lists of an eighth of the lists let ten times the candidates through)
    float* d_head_vecs = nullptr;      // [head rows + 64][128], lists packed
    float* d_head_norm = nullptr;      // [head rows + 64]
    int32_t* d_head_off = nullptr;     // [nlist + 1]
    int8_t* d_head_t8 = nullptr;       // byte-valued heads: 16-row tiles, every list padded to a multiple of 16 rows
    int32_t* d_head_rterm_t = nullptr;
    int32_t* d_head_tdelta = nullptr;  // [nlist] padded row - row
    int32_t* d_sh = nullptr;           // sharded brute force, host-buffer calls: the gathered scratch of the owner (see ShBuf)
    size_t sh_words = 0;
    float* d_sh_row = nullptr;         // ... and full distance rows (tie fallback)
    size_t sh_row_words = 0;
    int32_t* vsh_blk = nullptr;        // virtual ranks (vs_ivf_search_dev_vshards): the gathered blocks / top-k lists, owned by shard 0
    int32_t* vsh_loc = nullptr;
    size_t vsh_loc_words = 0;
    // wide IVF pipeline (a launch group of up to 32 batches shares one list-major pass): slot tables, zeroed counters,
    // plans, bounds, prepared queries, candidate sink
    struct IvfWide {
        int32_t* lq = nullptr;      // [ivf_nsb][nlist][kIvfWideQ]
        int32_t* zero = nullptr;    // one zeroed block per launch group (see wide_zero)
        size_t zero_words = 0;
        int32_t* units = nullptr;   // [n_sb_max][units_cap][4]
        int units_cap = 0;
        float* tau = nullptr;       // [1024]
        int32_t* tq = nullptr;      // [nlist][group queries] bound tables (ivf_bounds_list_body)
        float* tk = nullptr;        // [group queries][kBoundSegs][16]
        int32_t* nseg = nullptr;    // [group queries]
        float* qnorm = nullptr;     // [1024]
        int8_t* q8 = nullptr;       // [1024][128]
        int32_t* qterm = nullptr;   // [1024]
        int4* wbuf = nullptr;       // [waves][kIvfWideWaveCap]
        int n_waves = 0;
        float* cand_d = nullptr;    // [1024][16][kIvfWideSubCap]
        int32_t* rank_list = nullptr;   // groups of more than 2048 queries: [0] count, [1..] the queries the wave-per-query ranking left over
        int32_t* cand_i = nullptr;
        bool dirty = false;         // the zeroed block may hold a failed call's counts: memset before the next group
        char* slab = nullptr;       // per batch: probes [32][kMaxNprobe] | coarse scores [32][nlist padded]
        long long slab_stride = 0, off_scores = 0;
    } wide[kWideLanesMax];          // scratch sets: consecutive launch groups of one call run on different streams
    hipStream_t wide_stream[kWideLanesMax] = {};
    hipEvent_t wide_fork = nullptr, wide_join[kWideLanesMax] = {};
    int64_t n_units_max = 0;
    int n_chunks = 0;
    int32_t max_list = 0;              // longest resident list
    int max_grid = 0;
    unsigned one_calls = 0;            // single-call scans issued so far (they alternate the direction of their pass)

    // host-buffer API (vs_bf_search / vs_ivf_search): two slots of pinned staging + device I/O buffers, so that chunk
    // c + 1's query upload and chunk c - 1's result download run beside chunk c's kernels (copy streams + events)
    struct PipeSlot {
        float* pin_q = nullptr;    // [kMaxMulti * 32][128]
        char* pin_out = nullptr;   // dists | ids | flags of one chunk
        float* d_q = nullptr;
        float* d_out_d = nullptr;  // [kMaxMulti * 32][64]
        int32_t* d_out_i = nullptr;
        int32_t* d_flags = nullptr;
        hipEvent_t ev_h2d = nullptr, ev_comp = nullptr, ev_d2h = nullptr;
        int64_t q0 = -1, n = 0;    // the chunk in flight in this slot (q0 < 0: free)
    } pipe[2];
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    // vs_ivf_search through the wide pipeline: larger chunks (up to kIvfHostGroups launch groups each, dealt to the two
    // lanes), ONE upload and ONE download per chunk -- every hipMemcpyAsync costs the host tens of microseconds
    struct IvfHostSlot {
        float* pin_q = nullptr;   // [kIvfHostChunk][128]
        float* pin_out = nullptr; // dists [n][k] | ids [n][k] of one chunk
        float* d_q = nullptr;
        float* d_out = nullptr;   // same layout on the device
        hipEvent_t ev_h2d = nullptr, ev_comp[2] = {nullptr, nullptr}, ev_d2h = nullptr;
        int64_t q0 = -1, n = 0;
    } ihs[2];
    // tie resolver (flagged queries of vs_bf_search): distances to the first kTieDense rows, bound, filtered candidates
    float* d_tie_dense = nullptr;  // [32][kTieDense]
    char* pin_tie = nullptr;       // pinned staging of the tie resolver: dense [32][kTieDense] f32 | cnt [32] | rows [32][kTieCap] | dists [32][kTieCap]
    float* d_tie_tau = nullptr;    // [32]
    int32_t* d_tie_cnt = nullptr;  // [32]
    int32_t* d_tie_row = nullptr;  // [32][kTieCap]
    float* d_tie_d = nullptr;      // [32][kTieCap]

    // SearchTiming split of vs_ivf_search (IVFIndex.h:31-36): HIP events between the stages of every launch group
    std::vector<hipEvent_t> stage_ev;  // quadruples: start, after coarse + pick, after grouping, after scan + select
    int stage_used = 0;
    bool stage_on = false;
    double stage_ms[3] = {0, 0, 0};

    hipStream_t stream = nullptr;
    hipEvent_t ev_busy = nullptr;  // a call on another stream than the previous call's waits for that stream through this event
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    bool prof = false;
    ProfSlot prof_slot[2];
};

namespace {

int set_device(const vs_index* h) {
    HIPCHK(hipSetDevice(h->device));
    return VS_OK;
}

template <typename T>
int dev_alloc(T** p, size_t n) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T)));
    return VS_OK;
}

void free_all(vs_index* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (auto& L : h->lane) {
        if (L.slots) (void)hipFree(L.slots);
        if (L.done) (void)hipFree(L.done);
        if (L.part_d) (void)hipFree(L.part_d);
        if (L.part_i) (void)hipFree(L.part_i);
        if (L.seed_qnorm) (void)hipFree(L.seed_qnorm);
        if (L.qfrag) (void)hipFree(L.qfrag);
        if (L.q8frag) (void)hipFree(L.q8frag);
        if (L.seed_wmin) (void)hipFree(L.seed_wmin);
        if (L.tau0) (void)hipFree(L.tau0);
        void* wide[] = {L.q8, L.qterm, L.wcnt, L.wcand_d, L.wcand_i, L.wbuf};
        for (void* w : wide)
            if (w) (void)hipFree(w);
        if (L.done_ev) (void)hipEventDestroy(L.done_ev);
        if (L.s) (void)hipStreamDestroy(L.s);
    }
    if (h->fork) (void)hipEventDestroy(h->fork);
    void* ptrs[] = {h->d_vecs, h->d_norm, h->d_vecs_u8, h->d_rterm, h->d_seed_f32, h->d_seed_bnorm, h->d_seed_u8, h->d_seed_rterm, h->d_vecs_t8, h->d_nrh_t, h->d_rterm_t, h->d_r2o_t, h->d_tdelta, h->d_chunk_trow0, h->d_invalid, h->d_centroids, h->d_cnorm, h->d_offsets, h->d_r2o, h->d_q,
                    h->d_out_d, h->d_out_i,
                    h->d_flags, h->d_scores, h->d_probes, h->d_ipart_d, h->d_ipart_i, h->d_cand,
                    h->d_chunk_list, h->d_chunk_row0, h->d_chunk_rows, h->vsh_blk, h->vsh_loc,
                    h->d_head_vecs, h->d_head_norm, h->d_head_off, h->d_head_t8, h->d_head_rterm_t, h->d_head_tdelta, h->d_sh, h->d_sh_row};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    {
        for (auto& W : h->wide) {
            void* wd[] = {W.lq, W.zero, W.units, W.tau, W.qnorm, W.q8, W.qterm, W.wbuf, W.cand_d, W.cand_i, W.slab, W.rank_list};
            for (void* w : wd)
                if (w) (void)hipFree(w);
        }
        for (int i = 0; i < kWideLanesMax; ++i) {
            if (h->wide_stream[i]) (void)hipStreamDestroy(h->wide_stream[i]);
            if (h->wide_join[i]) (void)hipEventDestroy(h->wide_join[i]);
        }
        if (h->wide_fork) (void)hipEventDestroy(h->wide_fork);
    }
    for (auto& S : h->ihs) {
        if (S.pin_q) (void)hipHostFree(S.pin_q);
        if (S.pin_out) (void)hipHostFree(S.pin_out);
        if (S.d_q) (void)hipFree(S.d_q);
        if (S.d_out) (void)hipFree(S.d_out);
        hipEvent_t evs[] = {S.ev_h2d, S.ev_comp[0], S.ev_comp[1], S.ev_d2h};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
    }
    for (auto& ps : h->prof_slot)
        for (auto e : ps.ev) (void)hipEventDestroy(e);
    for (auto e : h->stage_ev) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        vs_index::PipeSlot& S = h->pipe[i];
        if (S.pin_q) (void)hipHostFree(S.pin_q);
        if (S.pin_out) (void)hipHostFree(S.pin_out);
        if (i > 0) {  // slot 0 aliases d_q / d_out_* / d_flags, freed above
            void* dp[] = {S.d_q, S.d_out_d, S.d_out_i, S.d_flags};
            for (void* q : dp)
                if (q) (void)hipFree(q);
        }
        hipEvent_t evs[] = {S.ev_h2d, S.ev_comp, S.ev_d2h};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
    }
    if (h->pin_tie) (void)hipHostFree(h->pin_tie);
    void* tie[] = {h->d_tie_dense, h->d_tie_tau, h->d_tie_cnt, h->d_tie_row, h->d_tie_d};
    for (void* q : tie)
        if (q) (void)hipFree(q);
    if (h->s_h2d) (void)hipStreamDestroy(h->s_h2d);
    if (h->s_d2h) (void)hipStreamDestroy(h->s_d2h);
    if (h->ev_busy) (void)hipEventDestroy(h->ev_busy);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return VS_ERR_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device index out of range");
        return VS_ERR_INVALID;
    }
    return VS_OK;
}

// scan geometry for n rows: persistent-style grid of at most one workgroup per CU
// min_tp > 0: a smaller shard uses fewer workgroups so that each still has min_tp (16-row) tiles, the amount from
// which the threshold exchange pays -- as long as that keeps at least half of the CUs busy
void scan_geometry(int64_t rows, int num_cus, int& grid, int& tiles_per_wg, int min_tp = 0) {
    const int64_t tiles = (rows + vs::kTileRows - 1) / vs::kTileRows;
    int64_t g = std::min<int64_t>(num_cus, (tiles + vs::kScanWaves - 1) / vs::kScanWaves);
    if (min_tp > 0 && tiles / g < min_tp && tiles / min_tp >= num_cus / 2) g = tiles / min_tp;
    g = std::max<int64_t>(std::min<int64_t>(g, vs::kSlotStride), 1);
    tiles_per_wg = (int)((tiles + g - 1) / g);
    grid = (int)((tiles + tiles_per_wg - 1) / tiles_per_wg);
    grid = std::max(grid, 1);
}

// scratch of the query-major IVF fallback (one batch at a time)
int alloc_ivf_scratch(vs_index* h) {
    int rc;
    h->scores_cap = (int64_t)32 * ((h->nlist + 63) & ~63);
    if ((rc = dev_alloc(&h->d_scores, (size_t)h->scores_cap))) return rc;
    if ((rc = dev_alloc(&h->d_probes, 32 * kMaxNprobe))) return rc;
    if ((rc = dev_alloc(&h->d_ipart_d, (size_t)32 * kMaxNprobe * kKcapMax))) return rc;
    if ((rc = dev_alloc(&h->d_ipart_i, (size_t)32 * kMaxNprobe * kKcapMax))) return rc;
    return VS_OK;
}

int alloc_scratch(vs_index* h) {
    int rc;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("VSEARCH_GRID_CUS")) h->num_cus = std::max(1, std::min(h->num_cus, atoi(e)));  // tuning knob
    int tp;
    scan_geometry(std::max<int64_t>(h->n_rows, 1), h->num_cus, h->max_grid, tp);
    h->max_grid = std::max(h->max_grid, h->num_cus);
    if ((rc = dev_alloc(&h->d_q, (size_t)kMaxMulti * 32 * vs::kDim))) return rc;
    {
        const char* e = getenv("VSEARCH_LANES");
        h->n_lanes = e ? std::max(1, std::min(kMaxLanes, atoi(e))) : 1;
        const size_t nslot = (size_t)kMaxMulti * 32 * vs::kSlotStride;
        const size_t part = (size_t)kMaxMulti * 32 * vs::kSlotStride * kKcapMax;
        for (int i = 0; i < h->n_lanes; ++i) {
            vs_index::Lane& L = h->lane[i];
            if ((rc = dev_alloc(&L.slots, nslot))) return rc;
            if ((rc = dev_alloc(&L.seed_qnorm, (size_t)kMaxMulti * 32))) return rc;
            if ((rc = dev_alloc(&L.qfrag, (size_t)kMaxMulti * 4096))) return rc;
            if ((rc = dev_alloc(&L.q8frag, (size_t)kMaxMulti * 4096))) return rc;
            if ((rc = dev_alloc(&L.seed_wmin, (size_t)kMaxMulti * vs::kSeedWaves * 32))) return rc;
            if ((rc = dev_alloc(&L.tau0, (size_t)kMaxMulti * 32))) return rc;
            if ((rc = dev_alloc(&L.done, kMaxMulti))) return rc;
            HIPCHK(hipMemset(L.done, 0, kMaxMulti * sizeof(int)));  // arrival counter of the single-call scan: 0 between launches
            if (h->kind == 0) {
                if ((rc = dev_alloc(&L.part_d, part))) return rc;
                if ((rc = dev_alloc(&L.part_i, part))) return rc;
            }
            HIPCHK(hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&L.done_ev, hipEventDisableTiming));
        }
        HIPCHK(hipEventCreateWithFlags(&h->fork, hipEventDisableTiming));
    }
    if ((rc = dev_alloc(&h->d_out_d, (size_t)kMaxMulti * 32 * 64))) return rc;
    if ((rc = dev_alloc(&h->d_out_i, (size_t)kMaxMulti * 32 * 64))) return rc;
    if ((rc = dev_alloc(&h->d_flags, (size_t)kMaxMulti * 32))) return rc;
    if (h->kind == 1) {
        if ((rc = dev_alloc(&h->d_cand, 1))) return rc;
        if ((rc = alloc_ivf_scratch(h))) return rc;
    }
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    return VS_OK;
}

// upload `rows x dim` floats in chunks through the default pageable path and compute norms
int upload_vectors(vs_index* h, const float* host, int64_t rows) {
    int rc;
    if ((rc = dev_alloc(&h->d_vecs, ((size_t)std::max<int64_t>(rows, 1) + vs::kScanPadRows) * vs::kDim))) return rc;
    HIPCHK(hipMemset(h->d_vecs + (size_t)std::max<int64_t>(rows, 1) * vs::kDim, 0, (size_t)vs::kScanPadRows * vs::kDim * sizeof(float)));
    if ((rc = dev_alloc(&h->d_norm, (size_t)rows + 64))) return rc;
    HIPCHK(hipMemset(h->d_norm, 0, ((size_t)rows + 64) * sizeof(float)));
    if (rows > 0) {
        HIPCHK(hipMemcpy(h->d_vecs, host, (size_t)rows * vs::kDim * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(vs::launch_row_sqnorm(h->d_vecs, rows, vs::kDim, h->d_norm, nullptr));
        HIPCHK(hipDeviceSynchronize());
    }
    return VS_OK;
}

// int8 copy of the base when it is exactly representable: bytes (x - 128) and the per-row term
// ||b||^2 - 256 * sum(b - 128); dist = [||q||^2 - 256 sum(q-128) - 2*128^3] + rterm - 2 * sum((q-128)(b-128)).
int build_u8_copy(vs_index* h, const float* host, int64_t rows, std::vector<int8_t>* bytes_out = nullptr, std::vector<int32_t>* rterm_out = nullptr) {
    std::vector<int8_t> bytes(((size_t)rows + vs::kScanPadRows) * vs::kDim, 0);  // spare rows: tile DMAs are not clamped
    std::vector<int32_t> rterm((size_t)rows + 64, 0);
    for (int64_t i = 0; i < rows; ++i) {
        int32_t n2 = 0, sb = 0;
        for (int t = 0; t < vs::kDim; ++t) {
            const float x = host[i * vs::kDim + t];
            const int xi = (int)x;
            if (!((float)xi == x) || xi < 0 || xi > 255) return VS_OK;  // not representable: fp32 path only
            bytes[(size_t)i * vs::kDim + t] = (int8_t)(xi - 128);
            n2 += xi * xi;
            sb += xi - 128;
        }
        rterm[(size_t)i] = n2 - 256 * sb;
    }
    int rc;
    if ((rc = dev_alloc(&h->d_vecs_u8, bytes.size()))) return rc;
    if ((rc = dev_alloc(&h->d_rterm, rterm.size()))) return rc;
    if ((rc = dev_alloc(&h->d_invalid, (size_t)kMaxMulti))) return rc;
    HIPCHK(hipMemcpy(h->d_vecs_u8, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_rterm, rterm.data(), rterm.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (bytes_out) bytes_out->swap(bytes);
    if (rterm_out) rterm_out->swap(rterm);
    return VS_OK;
}

void prof_begin(vs_index* h, int which, hipStream_t s) {
    if (!h->prof) return;
    ProfSlot& ps = h->prof_slot[which];
    if (ps.used + 2 > kMaxEvents) return;
    while ((int)ps.ev.size() < ps.used + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        ps.ev.push_back(e);
    }
    (void)hipEventRecord(ps.ev[ps.used], s);
}
void prof_end(vs_index* h, int which, hipStream_t s) {
    if (!h->prof) return;
    ProfSlot& ps = h->prof_slot[which];
    if ((int)ps.ev.size() < ps.used + 2) return;
    (void)hipEventRecord(ps.ev[ps.used + 1], s);
    ps.used += 2;
}

// stage mark i (0..3) of the current launch group (vs_ivf_search only)
void stage_mark(vs_index* h, int i, hipStream_t s) {
    if (!h->stage_on) return;
    if (i == 0 && h->stage_used + 4 > kMaxEvents) {
        h->stage_on = false;
        return;
    }
    while ((int)h->stage_ev.size() < h->stage_used + 4) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            h->stage_on = false;
            return;
        }
        h->stage_ev.push_back(e);
    }
    (void)hipEventRecord(h->stage_ev[h->stage_used + i], s);
    if (i == 3) h->stage_used += 4;
}

// tuning knob (VSEARCH_XCHG_IT): loop iteration of the first threshold-exchange attempt
int g_xchg_first_it = [] {
    const char* e = getenv("VSEARCH_XCHG_IT");
    return e ? atoi(e) : 1;
}();

int* g_dbg = nullptr;
// tuning knob (VSEARCH_SEED_MIN): multi-batch launches of at least this many batches take their bounds from
// launch_seed instead of the in-kernel exchange (0 = never)
int g_seed_min_batches = [] {
    const char* e = getenv("VSEARCH_SEED_MIN");
    const int v = e ? atoi(e) : 4;
    return v > 0 ? v : 1 << 30;
}();

int g_seed_i8 = [] {
    const char* e = getenv("VSEARCH_SEED_I8");
    return e ? atoi(e) : 1;
}();

// tuning knob (VSEARCH_I8_WIDE): query column blocks per pass of the wide int8 scan (8 = four 32-query batches share one
// pass over the rows, 4 = two, 0 = off: the per-batch int8 scan)
int g_i8_wide = [] {
    const char* e = getenv("VSEARCH_I8_WIDE");
    const int v = e ? atoi(e) : 8;
    return v >= 12 ? 12 : (v >= 8 ? 8 : (v >= 4 ? 4 : 0));
}();

// tuning knob (VSEARCH_F32_PAIR=0): the fp32 streaming scan makes one pass over the rows per batch (HBM bound) instead of
// one per two batches (MFMA bound); 2 = pair on shards of any size (default 1: from kPairMinTiles tiles per workgroup on)
int g_f32_pair = [] {
    const char* e = getenv("VSEARCH_F32_PAIR");
    return e ? atoi(e) : 1;
}();

// tuning knob (VSEARCH_STREAM=0): seeded launches use the per-batch scan kernels (lane lists + workgroup merge) instead of
// the streaming scans
int g_stream = [] {
    const char* e = getenv("VSEARCH_STREAM");
    return e ? atoi(e) : 1;
}();

int ensure_wide(vs_index::Lane& L) {
    if (L.wbuf) return VS_OK;  // (the buffer allocated LAST: a call that ran out of memory half way is repeated in full)
    void* partial[] = {L.q8, L.qterm, L.wcnt, L.wcand_d, L.wcand_i};
    for (void* q : partial)
        if (q) (void)hipFree(q);
    L.q8 = nullptr;
    L.qterm = nullptr;
    L.wcnt = nullptr;
    L.wcand_d = nullptr;
    L.wcand_i = nullptr;
    int rc;
    if ((rc = dev_alloc(&L.q8, (size_t)kMaxMulti * 32 * vs::kDim))) return rc;
    if ((rc = dev_alloc(&L.qterm, (size_t)kMaxMulti * 32))) return rc;
    if ((rc = dev_alloc(&L.wcnt, (size_t)kMaxMulti * 32 * kWideSub + 64))) return rc;  // overflow word | skipped batches | list counters
    if ((rc = dev_alloc(&L.wcand_d, (size_t)kMaxMulti * 32 * kWideSub * kWideCap))) return rc;
    if ((rc = dev_alloc(&L.wcand_i, (size_t)kMaxMulti * 32 * kWideSub * kWideCap))) return rc;
    if ((rc = dev_alloc(&L.wbuf, (size_t)vs::kSlotStride * vs::kScanWaves * kWideWaveCap))) return rc;
    return VS_OK;
}

int pick_kcap(int need) { return need <= 8 ? 8 : (need <= 16 ? 16 : 0); }

// nb <= kMaxMulti batches of B queries in ONE persistent launch on stream s (preceded by the seed launches or the
// slot reset, followed by one merge launch), outputs [nb][B][k1].  Consecutive calls on one lane must be stream-ordered.
int bf_launch(vs_index* h, vs_index::Lane& L, const float* q_dev, int nb, int B, int k1, float* out_d, int32_t* out_i,
              int32_t* flags, hipStream_t s, bool force_f32 = false) {
    const int kcap = pick_kcap(k1);
    if (!kcap) {
        set_error("k too large for the compiled scan kernels (k <= 15)");
        return VS_ERR_UNSUPPORTED;
    }
    const int nqh = B <= 16 ? 1 : 2;
    vs::ScanParams p{};
    p.base = h->d_vecs;
    p.bnorm = h->d_norm;
    p.q = q_dev;
    p.n_batches = nb;
    p.q_batch_stride = (int64_t)B * vs::kDim;
    p.metric = h->metric;
    p.id_offset = (int32_t)h->id_offset;
    p.nq_valid = B;
    p.k1 = k1;
    p.dbg = g_dbg;
    // the exchange pays once every wave has a few tiles left after its warm-up tiles (int8 tiles hold 64 rows)
    const bool u8_path = h->d_vecs_u8 && h->precision != 1 && !force_f32;
    // a multi-batch launch gets its bounds up front from a sample of the rows (three small launches for all
    // batches): every batch then streams from its first tile on, whatever the shard size (as long as the 2048
    // sample tiles are a minority of it); a short call keeps the in-kernel exchange, which needs a few tiles per
    // wave after its warm-up tiles (int8 tiles hold 64 rows)
    const int64_t tiles_total = (h->n_rows + vs::kTileRows - 1) / vs::kTileRows;
    // The streaming scans' candidate buffers are sized for what survives the seeded bounds: about n_rows * k1 / 32768 rows per
    // query (the sample is 32 768 rows), in kWideSub sub-lists of kWideCap entries.  A shard on which that expectation passes
    // half the capacity (5.6 M rows at k = 5) would overflow on nearly every launch and run the fallback scan as well:
    // it takes the per-batch scan directly.
    const bool cap_ok = (double)h->n_rows * k1 <= 0.5 * 32768.0 * kWideSub * kWideCap;
    const bool seeded = nb >= g_seed_min_batches && tiles_total >= 2 * vs::kSeedWaves && g_xchg_first_it >= 0 && cap_ok;
    int grid, tp;
    scan_geometry(h->n_rows, h->num_cus, grid, tp, seeded ? 0 : (u8_path ? 16 : 6) * vs::kScanWaves);
    const bool exchange = !seeded && grid >= 16 && tp >= (u8_path ? 16 : 6) * vs::kScanWaves && g_xchg_first_it >= 0;
    const bool use_u8 = u8_path;
    // With the bounds known up front the batches need nothing from each other: the streaming scans (fp32: one batch per
    // pass; exact int8 rows: several batches per pass over the rows) hand out the tiles of ALL batches through one
    // ticket per workgroup and write survivors to candidate lists -- no barrier, no workgroup merge, no per-batch prologue.
    const bool i8_seed = h->d_vecs_u8 && h->metric == VS_METRIC_L2 && g_seed_i8;
    const bool stream = seeded && g_stream && (use_u8 ? (g_i8_wide > 0 && i8_seed) : true);
    if (!seeded && !use_u8 && nb < kOneMaxBatches && B <= kOneMaxQueries) {
        // a short call on the fp32 rows: one launch per batch (lane lists, workgroup ranking, the last workgroup merges).
        // Batches of more than 16 queries stay with the per-batch scan below: with two column blocks per tile the
        // single-call kernel's lane lists cost more than that kernel's threshold exchange (measured at 1 M rows, B = 32:
        // 154 us against 102 us; B <= 16: 88 - 97 us against 102 - 114 us)
        for (int b = 0; b < nb; ++b) {
            vs::OneParams op{};
            op.base = h->d_vecs;
            op.bnorm = h->d_norm;
            op.n_rows = h->n_rows;
            op.q = q_dev + (size_t)b * B * vs::kDim;
            op.nq_valid = B;
            op.k1 = k1;
            op.metric = h->metric;
            op.reverse = (int)(h->one_calls++ & 1u);
            op.id_offset = (int32_t)h->id_offset;
            op.part_d = L.part_d;
            op.part_i = L.part_i;
            op.done = L.done;
            op.out_d = out_d + (size_t)b * B * k1;
            op.out_i = out_i + (size_t)b * B * k1;
            op.flags = flags ? flags + (size_t)b * B : nullptr;
            op.dbg = g_dbg;
            if (b == 0) prof_begin(h, 0, s);
            HIPCHK(vs::launch_scan_one(op, vs::scan_one_grid(h->n_rows, h->num_cus), s));
            if (b == nb - 1) prof_end(h, 0, s);
        }
        return VS_OK;
    }
    if (seeded || use_u8) {
        int rc = ensure_wide(L);
        if (rc) return rc;
    }
    // one zeroed block per launch: [0] overflow word | [16, 48) batches the int8 path has to skip | [64, ...) list counters
    int32_t* const overflow = L.wcnt;
    int32_t* const invalid = L.wcnt ? L.wcnt + 16 : nullptr;
    // (cleared by the seed's query-preparation launch where there is one and nobody else writes the batches' verdict words)
    const size_t zero_words = 64 + (stream ? (size_t)nb * 32 * kWideSub : 0);
    const bool zero_in_seed = seeded && (!use_u8 || i8_seed);
    if (L.wcnt && !zero_in_seed) HIPCHK(hipMemsetAsync(L.wcnt, 0, zero_words * sizeof(int32_t), s));
    if (seeded) {
        vs::SeedParams sp{};
        sp.base = h->d_vecs;
        sp.bnorm = h->d_norm;
        sp.sample_f32 = h->d_seed_f32;
        sp.sample_bnorm = h->d_seed_bnorm;
        if (i8_seed) {  // exact int8 copy of the rows: 16x cheaper seed
            sp.base_u8 = h->d_vecs_u8;
            sp.rterm = h->d_rterm;
            sp.sample_u8 = h->d_seed_u8;
            sp.sample_rterm = h->d_seed_rterm;
        }
        sp.n_rows = h->n_rows;
        sp.q = q_dev;
        sp.n_batches = nb;
        sp.q_batch_stride = (int64_t)B * vs::kDim;
        sp.nq_valid = B;
        sp.metric = h->metric;
        sp.k1 = k1;
        sp.qnorm = L.seed_qnorm;
        sp.wmin = L.seed_wmin;
        sp.tau0 = L.tau0;
        if (zero_in_seed) {
            sp.zero = L.wcnt;
            sp.zero_words = (int)zero_words;
        }
        sp.qfrag = L.qfrag;
        if (i8_seed) {  // queries as bytes + constant terms + the "not byte valued" verdict: int8 seed and wide int8 scan
            sp.q8 = L.q8;
            sp.q8frag = L.q8frag;
            sp.qterm = L.qterm;
            sp.invalid = invalid;
        }
        HIPCHK(vs::launch_seed(sp, s));
        p.tau0 = L.tau0;
    } else if (exchange) {
        // 0x7f800000 = +inf
        HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(L.slots), 0x7f800000, (size_t)nb * 32 * vs::kSlotStride, s));
        p.slots_cur = L.slots;
    }
    vs::MergeParams m{};
    m.nq = nb * B;
    m.kout = k1;
    m.out_d = out_d;
    m.out_i = out_i;
    m.flags = flags;
    m.q_group_out = B;
    m.q_group_in = vs::kMaxBatch;
    m.invalid = use_u8 ? invalid : nullptr;
    if (stream) {
        vs::CandSink sink{};
        sink.wbuf = L.wbuf;
        sink.wcap = kWideWaveCap;
        sink.overflow = overflow;
        sink.cnt = L.wcnt + 64;
        sink.cand_d = L.wcand_d;
        sink.cand_i = L.wcand_i;
        sink.cap = kWideCap;
        sink.nsub = kWideSub;
        prof_begin(h, 0, s);
        if (use_u8) {
            vs::WideParams wp{};
            wp.base_u8 = h->d_vecs_u8;
            wp.rterm = h->d_rterm;
            wp.n_rows = h->n_rows;
            wp.q8 = L.q8;
            wp.q8frag = L.q8frag;
            wp.qterm = L.qterm;
            wp.tau0 = L.tau0;
            wp.invalid = invalid;
            wp.n_batches = nb;
            wp.nq_valid = B;
            wp.bpb = nqh;
            wp.id_offset = (int32_t)h->id_offset;
            wp.sink = sink;
            const int64_t tiles64 = (h->n_rows + 63) / 64;
            const int wgrid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(h->num_cus, vs::kSlotStride), tiles64));
            HIPCHK(vs::launch_scan_i8_wide(wp, wgrid, g_i8_wide, s));
            prof_end(h, 0, s);
        } else {
            vs::StreamParams sp{};
            sp.base = h->d_vecs;
            sp.bnorm = h->d_norm;
            sp.n_rows = h->n_rows;
            sp.q = q_dev;
            sp.q_batch_stride = (int64_t)B * vs::kDim;
            sp.qfrag = L.qfrag;
            sp.qnorm = L.seed_qnorm;
            sp.tau0 = L.tau0;
            sp.n_batches = nb;
            sp.nq_valid = B;
            sp.metric = h->metric;
            sp.id_offset = (int32_t)h->id_offset;
            sp.sink = sink;
            const int sgrid = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(h->num_cus, vs::kSlotStride), tiles_total));
            // two batches per pass pay where a workgroup has many tiles per pass (1 M rows: 244, + 14 %); on a small shard
            // (125 K rows: 30 tiles) the per-pass operand fetch and drain weigh more than the halved traffic saves (- 15 %)
            sp.batches_per_pass = (nb >= 2 && (g_f32_pair > 1 || (g_f32_pair == 1 && tiles_total / sgrid >= kPairMinTiles))) ? 2 : 1;
            HIPCHK(vs::launch_scan_f32_stream(sp, sgrid, s));
            prof_end(h, 0, s);
        }
        // every query's candidate list (unsorted, a few hundred entries) -> k1 best by (dist, id), tie flags
        vs::MergeParams mf = m;
        mf.part_d = L.wcand_d;
        mf.part_i = L.wcand_i;
        mf.G = kWideSub;
        mf.kin = kWideCap;
        mf.flat_len = L.wcnt + 64;
        mf.run_if = overflow;
        mf.run_mode = 2;
        HIPCHK(vs::launch_merge_layout(mf, kWideCap, (int64_t)kWideSub * kWideCap, s));
        // Fallback, enqueued behind and idle unless a candidate buffer overflowed (thousands of rows under one query's
        // bound: masses of duplicates): the per-batch scan with lane lists, which copes with any data.
        p.run_if = overflow;
        m.run_if = overflow;
        m.run_mode = 1;
    }
    p.row_begin = 0;
    p.row_end = h->n_rows;
    p.tiles_per_wg = tp;
    p.part_d = L.part_d;
    p.part_i = L.part_i;
    if (use_u8) {
        p.base_u8 = h->d_vecs_u8;
        p.rterm = h->d_rterm;
        p.invalid = invalid;
    }
    if (!stream) prof_begin(h, 0, s);
    HIPCHK(vs::launch_scan(p, grid, kcap, nqh, vs::kModeTopK, s));
    if (!stream) prof_end(h, 0, s);
    // one merge launch ranks every (batch, query): lists are [batch*32 + q][workgroup][kcap]
    m.part_d = L.part_d;
    m.part_i = L.part_i;
    m.G = grid;
    m.kin = kcap;
    HIPCHK(vs::launch_merge_layout(m, kcap, (int64_t)vs::kSlotStride * kcap, s));
    return VS_OK;
}

int bf_batch_dev(vs_index* h, vs_index::Lane& L, const float* q_dev, int B, int k1, float* out_d, int32_t* out_i,
                 int32_t* flags, hipStream_t s) {
    return bf_launch(h, L, q_dev, 1, B, k1, out_d, out_i, flags, s);
}

// nb batches of B queries: chunks of kMaxMulti batches per persistent launch, on the caller's stream.
int bf_multi_dev(vs_index* h, const float* q_dev, int nb, int B, int k1, float* out_d, int32_t* out_i, int32_t* flags,
                 hipStream_t user) {
    for (int b0 = 0; b0 < nb; b0 += kMaxMulti) {
        const int n = std::min(kMaxMulti, nb - b0);
        int rc = bf_launch(h, h->lane[0], q_dev + (size_t)b0 * B * vs::kDim, n, B, k1, out_d + (size_t)b0 * B * k1,
                           out_i + (size_t)b0 * B * k1, flags ? flags + (size_t)b0 * B : nullptr, user);
        if (rc) return rc;
    }
    return VS_OK;
}

int scores_dev(vs_index* h, const float* vecs, const float* norms, int64_t rows, const float* q_dev, int B,
               float* scores, int64_t ld, hipStream_t s) {
    vs::ScanParams p{};
    p.base = vecs;
    p.bnorm = norms;
    p.q = q_dev;
    p.metric = h->metric;
    p.nq_valid = B;
    p.n_batches = 1;
    p.row_begin = 0;
    p.row_end = rows;
    int grid, tp;
    scan_geometry(rows, h->num_cus, grid, tp);
    p.tiles_per_wg = tp;
    p.store = scores;
    p.store_ld = ld;
    HIPCHK(vs::launch_scan(p, grid, 8, B <= 16 ? 1 : 2, vs::kModeStore, s));
    return VS_OK;
}

// tuning knob (VSEARCH_IVF_WIDE_LANES=1, read when an index is created): vs_ivf_search_dev_multi keeps all launch groups of
// a call on the caller's stream instead of alternating them between two streams
int ivf_wide_lanes() {
    const char* e = getenv("VSEARCH_IVF_WIDE_LANES");
    return e ? std::max(1, std::min(kWideLanesMax, atoi(e))) : 2;
}
// tuning knob (VSEARCH_IVF_GROUP, read when an index is created): batches per launch group of an unsharded index
// (multiple of 32, <= 256): every kernel of the pipeline is launched once per group
int ivf_group_batches() {
    const char* e = getenv("VSEARCH_IVF_GROUP");
    const int v = e ? atoi(e) : kIvfGroupDefault;
    return std::max(32, std::min(kIvfGroupMax, (v + 31) / 32 * 32));
}

// Is the wide list-major pipeline available for this index?  (nlist <= 4096, rows resident, k <= 16; otherwise the
// query-major fallback: coarse scores on the MFMA scan kernel, pick_probes, one workgroup per (query, probe).)
bool ivf_wide_ok(const vs_index* h, int k) { return h->nlist <= vs::kIvfFastNlist && h->n_chunks > 0 && pick_kcap(k) != 0; }

// Query-major fallback for one batch (nlist > 4096, or a shard without resident rows).
int ivf_fallback_batch_dev(vs_index* h, const float* q_dev, int B, int k, int nprobe, float* out_d, int32_t* out_i, hipStream_t s) {
    const int kcap = pick_kcap(k);
    if (!kcap) {
        set_error("k too large for the compiled IVF kernels (k <= 16)");
        return VS_ERR_UNSUPPORTED;
    }
    const int64_t ld = (h->nlist + 63) & ~63;
    stage_mark(h, 0, s);
    int rc = scores_dev(h, h->d_centroids, h->d_cnorm, h->nlist, q_dev, B, h->d_scores, ld, s);
    if (rc) return rc;
    HIPCHK(vs::launch_pick_probes(h->d_scores, ld, B, h->nlist, nprobe, h->d_probes, s));
    stage_mark(h, 1, s);
    stage_mark(h, 2, s);
    vs::IvfScanParams ip{};
    ip.vecs = h->d_vecs;
    ip.vnorm = h->d_norm;
    ip.offsets = h->d_offsets;
    ip.owned = nullptr;
    ip.q = q_dev;
    ip.probes = h->d_probes;
    ip.B = B;
    ip.nprobe = nprobe;
    ip.kcap = kcap;
    ip.metric = h->metric;
    ip.part_d = h->d_ipart_d;
    ip.part_i = h->d_ipart_i;
    ip.cand_count = h->d_cand;
    prof_begin(h, 1, s);
    HIPCHK(vs::launch_ivf_scan(ip, s));
    prof_end(h, 1, s);
    vs::MergeParams m{};
    m.part_d = h->d_ipart_d;
    m.part_i = h->d_ipart_i;
    m.G = nprobe;
    m.kin = kcap;
    m.nq = B;
    m.kout = k;
    m.out_d = out_d;
    m.out_i = out_i;
    m.id_map = h->d_r2o;
    HIPCHK(vs::launch_merge_layout(m, kcap, (int64_t)nprobe * kcap, s));
    stage_mark(h, 3, s);
    return VS_OK;
}

// Scratch of one lane of the wide pipeline, sized for launch groups of h->ivf_gb batches in up to h->ivf_nsb super-batches.
int ensure_ivf_wide(vs_index* h, int lane) {
    vs_index::IvfWide& W = h->wide[lane];
    if (W.slab) return VS_OK;
    int rc;
    const size_t nq = (size_t)h->ivf_gb * 32;
    const int n_sb_max = h->ivf_nsb;
    W.n_waves = 0;
    for (int n = 1; n <= n_sb_max; ++n) W.n_waves = std::max(W.n_waves, vs::ivf_wide_waves(h->num_cus, n));
    W.zero_words = (size_t)n_sb_max * vs::ivf_wide_plan_words(h->nlist) + nq + 16 + h->ivf_gb + nq * kWideSub;
    if ((rc = dev_alloc(&W.lq, (size_t)n_sb_max * h->nlist * vs::kIvfWideQ))) return rc;
    if ((rc = dev_alloc(&W.zero, W.zero_words))) return rc;
    HIPCHK(hipMemset(W.zero, 0, W.zero_words * sizeof(int32_t)));
    W.units_cap = (int)std::min<int64_t>(2 * h->n_units_max + 4096, 0x7fffffff / 16);
    if ((rc = dev_alloc(&W.units, (size_t)n_sb_max * W.units_cap * 4))) return rc;
    if ((rc = dev_alloc(&W.tau, nq))) return rc;
    if ((rc = dev_alloc(&W.tq, (size_t)h->nlist * nq))) return rc;
    if ((rc = dev_alloc(&W.tk, nq * vs::kBoundSegs * 16))) return rc;
    if ((rc = dev_alloc(&W.nseg, nq))) return rc;
    if ((rc = dev_alloc(&W.qnorm, nq))) return rc;
    if ((rc = dev_alloc(&W.q8, nq * vs::kDim))) return rc;
    if ((rc = dev_alloc(&W.qterm, nq))) return rc;
    if ((rc = dev_alloc(&W.wbuf, (size_t)W.n_waves * kIvfWideWaveCap))) return rc;
    if ((rc = dev_alloc(&W.cand_d, nq * kWideSub * kIvfWideSubCap))) return rc;
    if ((rc = dev_alloc(&W.cand_i, nq * kWideSub * kIvfWideSubCap))) return rc;
    if (nq > 2048) {
        if ((rc = dev_alloc(&W.rank_list, nq + 1))) return rc;
        HIPCHK(hipMemset(W.rank_list, 0, sizeof(int32_t)));
    }
    W.off_scores = (32ll * kMaxNprobe * 4 + 255) & ~255ll;
    W.slab_stride = (W.off_scores + 32ll * ((h->nlist + 63) & ~63) * 4 + 255) & ~255ll;
    if ((rc = dev_alloc(&W.slab, (size_t)W.slab_stride * h->ivf_gb))) return rc;
    return VS_OK;
}

// The zeroed block of a lane: plan words per super-batch | slow [nq] | overflow (16) | invalid [batches] | list counters [16][nq]
struct WideZero {
    int32_t *plan, *slow, *ovf, *invalid, *cnt;
};
WideZero wide_zero(const vs_index* h, const vs_index::IvfWide& W) {
    const size_t nq = (size_t)h->ivf_gb * 32;
    WideZero z;
    z.plan = W.zero;
    z.slow = z.plan + (size_t)h->ivf_nsb * vs::ivf_wide_plan_words(h->nlist);
    z.ovf = z.slow + nq;
    z.invalid = z.ovf + 16;
    z.cnt = z.invalid + h->ivf_gb;
    return z;
}

// Parameters of the wide pipeline's kernels for a launch group of nb batches in super-batches of sbb.
vs::IvfWideParams wide_params(vs_index* h, vs_index::IvfWide& W, const float* q_dev, int nb, int sbb, int B, int k, int nprobe,
                              float* out_d, int32_t* out_i) {
    const WideZero z = wide_zero(h, W);
    const size_t nq = (size_t)h->ivf_gb * 32;
    vs::IvfWideParams wp{};
#ifdef VS_STAMPS
    wp.dbg = g_dbg;
    wp.diag = getenv("VSEARCH_DIAG") ? atoi(getenv("VSEARCH_DIAG")) : 0;
#endif
    wp.vecs = h->d_vecs;
    wp.vnorm = h->d_norm;
    if (h->d_vecs_u8 && h->precision != 1 && h->metric == VS_METRIC_L2) {
        wp.vecs_u8 = h->d_vecs_u8;
        wp.rterm = h->d_rterm;
        wp.vecs_t8 = h->d_vecs_t8;
        wp.nrh_t = h->d_nrh_t;
        wp.rterm_t = h->d_rterm_t;
    }
    // (without the byte path the records and candidates are plain rows: no padded-row offsets)
    wp.tdelta = wp.vecs_t8 ? h->d_tdelta : nullptr;
    wp.chunk_trow0 = wp.vecs_t8 ? h->d_chunk_trow0 : h->d_chunk_row0;
    wp.offsets = h->d_offsets;
    wp.chunk_list = h->d_chunk_list;
    wp.chunk_row0 = h->d_chunk_row0;
    wp.chunk_rows = h->d_chunk_rows;
    wp.n_chunks = h->n_chunks;
    wp.nlist = h->nlist;
    wp.nprobe = nprobe;
    wp.k = k;
    wp.metric = h->metric;
    wp.q = q_dev;
    wp.q_batch_bytes = (long long)B * vs::kDim * sizeof(float);
    wp.n_batches = nb;
    wp.B = B;
    wp.sb_batches = sbb;
    wp.qnorm = W.qnorm;
    wp.q8 = W.q8;
    wp.qterm = W.qterm;
    wp.invalid = z.invalid;
    wp.probes = reinterpret_cast<int32_t*>(W.slab);
    wp.probes_batch_bytes = W.slab_stride;
    wp.lq = W.lq;
    wp.zero = z.plan;
    wp.cand_count = h->d_cand;
    wp.units = W.units;
    wp.units_sb_stride = (long long)W.units_cap * 4;
    wp.units_cap = W.units_cap;
    wp.tq = W.tq;
    wp.tq_cap = (int)nq;
    wp.tk = W.tk;
    wp.nseg = W.nseg;
    wp.tau = W.tau;
    wp.slow = z.slow;
    wp.sink.wbuf = W.wbuf;
    wp.sink.wcap = kIvfWideWaveCap;
    wp.sink.overflow = z.ovf;
    wp.sink.cnt = z.cnt;
    wp.sink.cand_d = W.cand_d;
    wp.sink.cand_i = W.cand_i;
    wp.sink.cap = kIvfWideSubCap;
    wp.sink.nsub = kWideSub;
    wp.sink.slow = z.slow;
    wp.sink.xcd_subs = kWideSub / 8;
    wp.sink.cnt_sub_stride = (int)nq;
    wp.out_d = out_d;
    wp.out_i = out_i;
    wp.id_map = h->d_r2o;
    return wp;
}

vs::IvfGroup wide_group(vs_index* h, vs_index::IvfWide& W, int sbb, int B) {
    const WideZero z = wide_zero(h, W);
    vs::IvfGroup grp{};
    grp.mb.slab = W.slab_stride;
    grp.mb.probes = W.slab_stride;  // (the probes sit at the head of a batch's slab)
    grp.mb.q = (long long)B * vs::kDim * sizeof(float);
    grp.sb_batches = sbb;
    grp.w_qnorm = W.qnorm;
    grp.w_q8 = W.q8;
    grp.w_qterm = W.qterm;
    grp.w_invalid = z.invalid;
    grp.w_overflow = z.ovf;
    grp.w_glist = W.rank_list;
    grp.w_cnt = z.plan;
    grp.w_lq = W.lq;
    grp.w_q = vs::kIvfWideQ;
    grp.w_tq = W.tq;
    grp.w_tq_cap = h->ivf_gb * 32;
    grp.w_tcnt = z.plan;
    grp.w_nseg = W.nseg;
    grp.t_offsets = h->d_offsets;
#ifdef VS_STAMPS
    grp.dbg = g_dbg ? g_dbg + 4096 * 16 : nullptr;
#endif
    return grp;
}

// scan + rank of a launch group whose slot tables, bounds and plan are in place
int wide_scan_rank(vs_index* h, vs_index::IvfWide& W, const vs::IvfWideParams& wp, hipStream_t s) {
    const WideZero z = wide_zero(h, W);
    const size_t nq = (size_t)h->ivf_gb * 32;
    prof_begin(h, 1, s);  // (the profiling window holds the scan kernel alone)
    HIPCHK(vs::launch_ivf_wide_scan(wp, h->num_cus, s));
    prof_end(h, 1, s);
    vs::MergeParams m{};
    m.part_d = W.cand_d;
    m.part_i = W.cand_i;
    m.G = kWideSub;
    m.kin = kIvfWideSubCap;
    m.nq = wp.n_batches * wp.B;
    m.kout = wp.k;
    m.out_d = wp.out_d;
    m.out_i = wp.out_i;
    m.q_group_out = wp.B;
    m.q_group_in = vs::kMaxBatch;
    m.flat_len = z.cnt;
    m.flat_len_sub_stride = (int)nq;
    m.id_map = (wp.vecs_t8 && h->d_r2o_t) ? h->d_r2o_t : h->d_r2o;  // the byte scan's candidates are padded rows
#ifdef VS_STAMPS
    m.dbg = g_dbg ? g_dbg + 8192 * 16 : nullptr;
#endif
    HIPCHK(vs::launch_ivf_wide_rank(m, kIvfWideSubCap, (int64_t)kWideSub * kIvfWideSubCap, wp, s, W.rank_list));
    return VS_OK;
}

// nb <= h->ivf_gb independent batches (one launch group) through the wide pipeline on scratch lane `lane`: coarse (MFMA,
// also prepares the byte queries) + pick (also fills the lists' slot tables), bounds and plan in one launch, ONE list-major
// pass per super-batch of 32 batches with candidates to the sink (binned by the scan's own waves), and the ranking launch
// (merge, or the exact slow path for queries without a bound / for everybody if a candidate buffer overflowed).
int ivf_group_wide_dev(vs_index* h, int lane, const float* q_dev, int nb, int B, int k, int nprobe, float* out_d, int32_t* out_i, hipStream_t s) {
    int rc = ensure_ivf_wide(h, lane);
    if (rc) return rc;
    vs_index::IvfWide& W = h->wide[lane];
    // The zeroed block is left zeroed by the group's last kernel (ivf_wide_rank_kernel): a memset only after a call that
    // did not get as far as that launch.
    if (W.dirty) HIPCHK(hipMemsetAsync(W.zero, 0, W.zero_words * sizeof(int32_t), s));
    W.dirty = true;
    const int sbb = vs::kIvfWideBatches;
    const vs::IvfGroup grp = wide_group(h, W, sbb, B);
    vs::IvfWideParams wp = wide_params(h, W, q_dev, nb, sbb, B, k, nprobe, out_d, out_i);
    wp.tau_inline = 1;  // (bounds launch and scan are in this one pipeline: the scan merges a query's segment lists itself)
    stage_mark(h, 0, s);
    HIPCHK(vs::launch_ivf_coarse_pick(q_dev, B, h->d_centroids, h->d_cnorm, h->nlist, nprobe, h->metric,
                                      reinterpret_cast<float*>(W.slab + W.off_scores), (h->nlist + 63) & ~63,
                                      reinterpret_cast<int32_t*>(W.slab), grp, s, nb));
    stage_mark(h, 1, s);
    HIPCHK(vs::launch_ivf_wide_bounds_plan(wp, s));
    stage_mark(h, 2, s);  // "gather" (IVFIndex.cpp's second stage) = bounds + plan here; the fine search is the scan + ranking
    if ((rc = wide_scan_rank(h, W, wp, s))) return rc;
#ifdef VS_STAMPS
    W.dirty = (wp.diag & 128) != 0;
#else
    W.dirty = false;
#endif
    stage_mark(h, 3, s);
    return VS_OK;
}

// ---- cluster-sharded pipeline (SURVEY 8e / BASELINE configs[4]).  A launch group is cut into `world` slices of sbb batches;
// slice r's per-query stages (coarse scores, probe selection, bound) run on rank r ONLY, their output -- a block of
// int32 words: probes [sbb * 32][nprobe] | tau [sbb * 32] | slow [sbb * 32] -- is exchanged (one all-gather), and every
// rank then scans its resident lists for ALL slices (slice = super-batch of the list-major pass).  What a rank does
// per launch group is therefore what an unsharded index does for ONE slice, except the ranking (all queries, an eighth of
// the candidates each).
long long ivf_block_words(int sbb, int nprobe) { return (long long)sbb * 32 * (nprobe + 2); }
// slice of `rank` in a launch group of nb batches on `world` ranks: sbb batches per slice, its own [b0, b0 + nbs)
void ivf_slice(int nb, int world, int rank, int& sbb, int& b0, int& nbs) {
    sbb = (nb + world - 1) / world;
    b0 = rank * sbb;
    nbs = std::max(0, std::min(sbb, nb - b0));
}

// front half on rank h->rank: prepares ALL queries of the group (bytes, terms, norms: the scan needs them for every
// slice), scores its own slice [b0, b0 + nbs) and writes the slice's block to `blk`
int ivf_shard_front(vs_index* h, int lane, const float* q_dev, int nb, int sbb, int b0, int nbs, int B, int k, int nprobe, int32_t* blk,
                    hipStream_t s) {
    int rc = ensure_ivf_wide(h, lane);
    if (rc) return rc;
    vs_index::IvfWide& W = h->wide[lane];
    if (W.dirty) HIPCHK(hipMemsetAsync(W.zero, 0, W.zero_words * sizeof(int32_t), s));
    W.dirty = true;
    vs::IvfGroup grp = wide_group(h, W, sbb, B);
    HIPCHK(vs::launch_ivf_prep_queries(q_dev, B, h->d_centroids, h->d_cnorm, h->nlist, grp, s, nb));
    const long long sb_q = (long long)sbb * 32;
    HIPCHK(hipMemsetAsync(blk + sb_q * nprobe + sb_q, 0, (size_t)sb_q * sizeof(int32_t), s));  // the slice's `slow` marks
    if (nbs <= 0) return VS_OK;
    // the slice's launches see their own batches only: every per-batch / per-query array is advanced to batch b0
    const size_t q0 = (size_t)b0 * 32;
    vs::IvfGroup fg = grp;
    fg.w_qnorm = nullptr;
    fg.w_q8 = nullptr;
    fg.w_qterm = nullptr;
    fg.w_invalid = nullptr;
    fg.w_overflow = nullptr;
    fg.w_glist = nullptr;
    fg.w_cnt = nullptr;  // (slot tables are filled after the exchange, for all slices)
    fg.w_lq = nullptr;
    fg.mb.probes = (long long)32 * nprobe * sizeof(int32_t);
    fg.t_offsets = h->d_head_vecs ? h->d_head_off : h->d_offsets;  // (the rows the bounds come from, see below)
    const float* qs = q_dev + (size_t)b0 * B * vs::kDim;
    HIPCHK(vs::launch_ivf_coarse_pick(qs, B, h->d_centroids, h->d_cnorm, h->nlist, nprobe, h->metric,
                                      reinterpret_cast<float*>(W.slab + W.off_scores), (h->nlist + 63) & ~63, blk, fg, s, nbs));
    vs::IvfWideParams wp = wide_params(h, W, qs, nbs, sbb, B, k, nprobe, nullptr, nullptr);
    wp.qnorm = W.qnorm + q0;
    wp.q8 = W.q8 + q0 * vs::kDim;
    wp.qterm = W.qterm + q0;
    wp.invalid = wp.invalid + b0;
    wp.probes = blk;
    wp.probes_batch_bytes = fg.mb.probes;
    wp.tau = reinterpret_cast<float*>(blk + sb_q * nprobe);
    wp.slow = blk + sb_q * nprobe + sb_q;
    if (h->d_head_vecs) {  // the bound's rows: the replicated heads of the query's two nearest lists, wherever those live
        wp.vecs = h->d_head_vecs;
        wp.vnorm = h->d_head_norm;
        wp.offsets = h->d_head_off;
        const bool bytes = h->d_head_t8 && h->precision != 1 && h->metric == VS_METRIC_L2;
        wp.vecs_u8 = bytes ? h->d_head_t8 : nullptr;  // (non-null = "byte rows exist"; the tiled copy is what is read)
        wp.rterm = nullptr;
        wp.vecs_t8 = bytes ? h->d_head_t8 : nullptr;
        wp.rterm_t = bytes ? h->d_head_rterm_t : nullptr;
        wp.nrh_t = nullptr;
        wp.tdelta = bytes ? h->d_head_tdelta : nullptr;
    }
    HIPCHK(vs::launch_ivf_wide_bounds_plan(wp, s, 1));  // bounds only
    return VS_OK;
}

// back half: slot tables for all slices from the exchanged blocks, plan, scan, rank -> this rank's top-k of every query
int ivf_shard_back(vs_index* h, int lane, const float* q_dev, int nb, int sbb, int B, int k, int nprobe, const int32_t* gathered,
                   float* out_d, int32_t* out_i, hipStream_t s) {
    vs_index::IvfWide& W = h->wide[lane];
    const WideZero z = wide_zero(h, W);
    const vs::IvfGroup grp = wide_group(h, W, sbb, B);
    const vs::IvfWideParams wp = wide_params(h, W, q_dev, nb, sbb, B, k, nprobe, out_d, out_i);
    HIPCHK(vs::launch_ivf_fill(gathered, ivf_block_words(sbb, nprobe), B, nprobe, h->nlist, h->d_offsets, reinterpret_cast<int32_t*>(W.slab),
                               W.tau, z.slow, grp, s, nb));
    HIPCHK(vs::launch_ivf_wide_bounds_plan(wp, s, 2));  // plan only
    int rc = wide_scan_rank(h, W, wp, s);
    if (rc) return rc;
    W.dirty = false;
    return VS_OK;
}

int ensure_wide_streams(vs_index* h) {
    if (h->wide_fork) return VS_OK;
    for (int i = 0; i < kWideLanesMax; ++i) {
        HIPCHK(hipStreamCreateWithFlags(&h->wide_stream[i], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->wide_join[i], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&h->wide_fork, hipEventDisableTiming));
    return VS_OK;
}

int ensure_ivf_host(vs_index* h) {
    int rc;
    h->ivf_host_cap = std::max<int64_t>(kIvfHostChunk, (int64_t)h->ivf_gb * 32);
    const size_t cap = (size_t)h->ivf_host_cap;
    for (auto& S : h->ihs)
        if (!S.pin_q) {
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&S.pin_q), cap * vs::kDim * sizeof(float), hipHostMallocDefault));
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&S.pin_out), cap * 64 * 2 * sizeof(float), hipHostMallocDefault));
            if ((rc = dev_alloc(&S.d_q, cap * vs::kDim))) return rc;
            if ((rc = dev_alloc(&S.d_out, cap * 64 * 2))) return rc;
            HIPCHK(hipEventCreateWithFlags(&S.ev_h2d, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&S.ev_comp[0], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&S.ev_comp[1], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&S.ev_d2h, hipEventDisableTiming));
        }
    return VS_OK;
}

// nb batches on the device, any number: launch groups of h->ivf_gb batches, alternating between the lanes' streams when
// there is more than one group (a group is a chain of dependent kernels, several of them small: the next group's small
// kernels fill the device beside the current group's scan and ranking)
int ivf_multi_dev(vs_index* h, const float* q_dev, int nb, int B, int k, int nprobe, float* out_d, int32_t* out_i, hipStream_t user) {
    int rc = VS_OK;
    if (!ivf_wide_ok(h, k)) {
        for (int b = 0; b < nb && !rc; ++b)
            rc = ivf_fallback_batch_dev(h, q_dev + (size_t)b * B * vs::kDim, B, k, nprobe, out_d + (size_t)b * B * k, out_i + (size_t)b * B * k, user);
        return rc;
    }
    const int gb = h->ivf_gb;
    const int groups = (nb + gb - 1) / gb;
    const int lanes = std::min({h->ivf_lanes, kWideLanesMax, groups});
    if (lanes <= 1) {
        for (int b0 = 0; b0 < nb && !rc; b0 += gb)
            rc = ivf_group_wide_dev(h, 0, q_dev + (size_t)b0 * B * vs::kDim, std::min(gb, nb - b0), B, k, nprobe, out_d + (size_t)b0 * B * k,
                                    out_i + (size_t)b0 * B * k, user);
        return rc;
    }
    if ((rc = ensure_wide_streams(h))) return rc;
    HIPCHK(hipEventRecord(h->wide_fork, user));
    for (int i = 0; i < lanes; ++i) HIPCHK(hipStreamWaitEvent(h->wide_stream[i], h->wide_fork, 0));
    int g = 0;
    for (int b0 = 0; b0 < nb && !rc; b0 += gb, ++g)
        rc = ivf_group_wide_dev(h, g % lanes, q_dev + (size_t)b0 * B * vs::kDim, std::min(gb, nb - b0), B, k, nprobe, out_d + (size_t)b0 * B * k,
                                out_i + (size_t)b0 * B * k, h->wide_stream[g % lanes]);
    for (int i = 0; i < lanes; ++i) {  // (also after an error: the user's stream must not run ahead of what was enqueued)
        HIPCHK(hipEventRecord(h->wide_join[i], h->wide_stream[i]));
        HIPCHK(hipStreamWaitEvent(user, h->wide_join[i], 0));
    }
    return rc;
}

// no C++ exception leaves the C ABI (vs_status instead)
template <class F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return VS_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error(std::string("internal error: ") + e.what());
        return VS_ERR_INVALID;
    }
}

// Calls on one index may arrive on different caller streams, but its scratch is one set: every enqueue waits for the
// previous call's work (an event), so that two calls never overlap on the device.
int order_begin(vs_index* h, hipStream_t s) {
    // Calls that stay on one stream are ordered by the stream itself: no event traffic (two runtime calls and two
    // barrier packets per search call are what a single-query call's latency is made of).  A call on ANOTHER stream than
    // the previous one waits for everything enqueued on that one so far.
    if (!h->ev_busy) HIPCHK(hipEventCreateWithFlags(&h->ev_busy, hipEventDisableTiming));
    if (h->have_last && h->last_stream != s) {
        HIPCHK(hipEventRecord(h->ev_busy, h->last_stream));
        HIPCHK(hipStreamWaitEvent(s, h->ev_busy, 0));
    }
    h->last_stream = s;  // (set here, not at the end: a call that fails half way has still enqueued work on s)
    h->have_last = true;
    return VS_OK;
}
int order_end(vs_index*, hipStream_t) { return VS_OK; }

int ensure_pipe(vs_index* h) {
    if (h->pipe[0].pin_q) return VS_OK;
    const size_t nqc = (size_t)kMaxMulti * 32;
    for (int i = 0; i < 2; ++i) {
        vs_index::PipeSlot& S = h->pipe[i];
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&S.pin_q), nqc * vs::kDim * sizeof(float), hipHostMallocDefault));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&S.pin_out), nqc * (64 * 2 + 1) * sizeof(float), hipHostMallocDefault));
        if (i == 0) {
            S.d_q = h->d_q;
            S.d_out_d = h->d_out_d;
            S.d_out_i = h->d_out_i;
            S.d_flags = h->d_flags;
        } else {
            int rc;
            if ((rc = dev_alloc(&S.d_q, nqc * vs::kDim))) return rc;
            if ((rc = dev_alloc(&S.d_out_d, nqc * 64))) return rc;
            if ((rc = dev_alloc(&S.d_out_i, nqc * 64))) return rc;
            if ((rc = dev_alloc(&S.d_flags, nqc))) return rc;
        }
        HIPCHK(hipEventCreateWithFlags(&S.ev_h2d, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&S.ev_comp, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&S.ev_d2h, hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&h->s_h2d, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h->s_d2h, hipStreamNonBlocking));
    return VS_OK;
}

// Full-row fallback of the tie resolver: the whole distance row of one query (already in h->d_q at row `b`) -> host replay.
int resolve_dense_full(vs_index* h, const float* q_dev, int B, const std::vector<int>& which, const int64_t* qidx, int k,
                       int32_t* ids, float* dists) {
    const int64_t ld = (h->n_rows + 15) & ~int64_t(15);
    if (h->scores_cap < (int64_t)B * ld) {
        if (h->d_scores) (void)hipFree(h->d_scores);
        h->d_scores = nullptr;
        h->scores_cap = 0;
        int rc = dev_alloc(&h->d_scores, (size_t)32 * ld);
        if (rc) return rc;
        h->scores_cap = (int64_t)32 * ld;
    }
    int rc = scores_dev(h, h->d_vecs, h->d_norm, h->n_rows, q_dev, B, h->d_scores, ld, h->stream);
    if (rc) return rc;
    std::vector<float> row((size_t)ld);
    for (int b : which) {
        HIPCHK(hipMemcpyAsync(row.data(), h->d_scores + (size_t)b * ld, (size_t)h->n_rows * sizeof(float), hipMemcpyDeviceToHost,
                              h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        vs::select_topk_slots_dense(row.data(), h->n_rows, k, (int32_t)h->id_offset, ids + qidx[b] * k, dists + qidx[b] * k);
    }
    return VS_OK;
}

// Exact select_topk (cpu_baseline.cpp:127-153) for queries whose k+1 best distances contain a tie.  The slot replay
// only depends on rows that change the k-slot buffer, i.e. rows whose distance is below the buffer maximum when they
// arrive -- and that maximum never rises.  So: the first kTieDense rows are taken densely (their distance rows), the
// k-th smallest of them bounds the buffer maximum for every later row, and ONE filtered pass over the rest of the base
// emits the few rows under that bound (about N * k / kTieDense of them).  The host replays the slots over
// "dense rows, then candidates in row order".  Exact whenever the distances are (integer-valued SIFT: always).
int resolve_ties(vs_index* h, const float* queries_host, const std::vector<int64_t>& flagged, int k, int32_t* ids, float* dists) {
    int rc;
    const int64_t L0 = std::min<int64_t>(h->n_rows, kTieDense);
    const int64_t L0p = (L0 + 15) & ~int64_t(15);
    if (!h->d_tie_dense) {
        if ((rc = dev_alloc(&h->d_tie_dense, (size_t)32 * kTieDense))) return rc;
        if ((rc = dev_alloc(&h->d_tie_tau, 32))) return rc;
        if ((rc = dev_alloc(&h->d_tie_cnt, 32))) return rc;
        if ((rc = dev_alloc(&h->d_tie_row, (size_t)32 * kTieCap))) return rc;
        if ((rc = dev_alloc(&h->d_tie_d, (size_t)32 * kTieCap))) return rc;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->pin_tie), (size_t)32 * kTieDense * 4 + 128 + (size_t)2 * 32 * kTieCap * 4, hipHostMallocDefault));
    }
    const int group = 32;
    // downloads land in pinned memory (pageable destinations are staged by the runtime: several times slower)
    float* const dense = reinterpret_cast<float*>(h->pin_tie);
    int32_t* const cnt = reinterpret_cast<int32_t*>(h->pin_tie + (size_t)32 * kTieDense * 4);
    int32_t* const cr = cnt + 32;
    float* const cd = reinterpret_cast<float*>(cr + (size_t)32 * kTieCap);
    std::vector<float> qbuf((size_t)group * vs::kDim);
    for (size_t f0 = 0; f0 < flagged.size(); f0 += group) {
        const int B = (int)std::min<size_t>(group, flagged.size() - f0);
        for (int b = 0; b < B; ++b)
            std::memcpy(&qbuf[(size_t)b * vs::kDim], queries_host + flagged[f0 + b] * vs::kDim, vs::kDim * sizeof(float));
        HIPCHK(hipMemcpyAsync(h->d_q, qbuf.data(), (size_t)B * vs::kDim * sizeof(float), hipMemcpyHostToDevice, h->stream));
        if ((rc = scores_dev(h, h->d_vecs, h->d_norm, L0, h->d_q, B, h->d_tie_dense, L0p, h->stream))) return rc;
        const bool sparse = h->n_rows > L0;
        if (sparse) {
            // bound = next_up(k-th smallest of the dense rows): merge kernel over G = L0 one-entry lists
            HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_tie_tau), 0xff800000u, 32, h->stream));  // -inf: padding queries emit nothing
            HIPCHK(hipMemsetAsync(h->d_tie_cnt, 0, 32 * sizeof(int32_t), h->stream));
            vs::MergeParams m{};
            m.part_d = h->d_tie_dense;
            m.G = (int)L0;
            m.kin = 1;
            m.nq = B;
            m.kout = k;
            m.tau_out = h->d_tie_tau;
            HIPCHK(vs::launch_merge_layout(m, 1, L0p, h->stream));
            vs::ScanParams p{};
            p.base = h->d_vecs;
            p.bnorm = h->d_norm;
            p.q = h->d_q;
            p.n_batches = 1;
            p.metric = h->metric;
            p.nq_valid = B;
            p.k1 = k + 1;
            p.tau0 = h->d_tie_tau;
            p.row_begin = L0;  // multiple of 16 (kTieDense)
            p.row_end = h->n_rows;
            p.f_cnt = h->d_tie_cnt;
            p.f_row = h->d_tie_row;
            p.f_d = h->d_tie_d;
            p.f_cap = kTieCap;
            int grid, tp;
            scan_geometry(h->n_rows - L0, h->num_cus, grid, tp);
            p.tiles_per_wg = tp;
            HIPCHK(vs::launch_scan(p, grid, 8, 2, vs::kModeFilter, h->stream));
            HIPCHK(hipMemcpyAsync(cnt, h->d_tie_cnt, 32 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(hipMemcpyAsync(dense, h->d_tie_dense, (size_t)B * L0p * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        int mx = 0;
        std::vector<int> overflow;
        if (sparse) {
            for (int b = 0; b < B; ++b) {
                if (cnt[b] > kTieCap) overflow.push_back(b);
                else mx = std::max(mx, cnt[b]);
            }
            if (mx > 0) {
                HIPCHK(hipMemcpy2DAsync(cr, (size_t)mx * 4, h->d_tie_row, (size_t)kTieCap * 4, (size_t)mx * 4, B,
                                        hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipMemcpy2DAsync(cd, (size_t)mx * 4, h->d_tie_d, (size_t)kTieCap * 4, (size_t)mx * 4, B,
                                        hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
            }
        }
        // the flagged queries are independent: their replays run on a few host threads (a replay walks ~5000 entries)
        auto replay = [&](int b_begin, int b_end) {
            std::vector<int32_t> srow, order;
            std::vector<float> sdist;
            for (int b = b_begin; b < b_end; ++b) {
                const int m = sparse ? cnt[b] : 0;
                if (m > kTieCap) continue;  // full-row fallback below
                srow.resize((size_t)L0 + m);
                sdist.resize((size_t)L0 + m);
                for (int64_t j = 0; j < L0; ++j) {
                    srow[(size_t)j] = (int32_t)(j + h->id_offset);
                    sdist[(size_t)j] = dense[(size_t)b * L0p + j];
                }
                order.resize((size_t)m);
                std::iota(order.begin(), order.end(), 0);
                const int32_t* rr = m ? &cr[(size_t)b * mx] : nullptr;
                const float* dd = m ? &cd[(size_t)b * mx] : nullptr;
                std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return rr[x] < rr[y]; });
                for (int j = 0; j < m; ++j) {
                    srow[(size_t)L0 + j] = rr[order[(size_t)j]] + (int32_t)h->id_offset;
                    sdist[(size_t)L0 + j] = dd[order[(size_t)j]];
                }
                const int64_t qi = flagged[f0 + b];
                vs::select_topk_slots_sparse(srow.data(), sdist.data(), (int64_t)srow.size(), k, ids + qi * k, dists + qi * k);
            }
        };
        const int n_thr = std::min(4, B / 4);
        if (n_thr <= 1) {
            replay(0, B);
        } else {
            std::vector<std::thread> pool;
            std::atomic<bool> failed{false};
            for (int t = 0; t < n_thr; ++t)
                pool.emplace_back([&, t]() {
                    try {
                        replay(B * t / n_thr, B * (t + 1) / n_thr);
                    } catch (...) {
                        failed = true;
                    }
                });
            for (auto& th : pool) th.join();
            if (failed) {
                set_error("out of host memory");
                return VS_ERR_NOMEM;
            }
        }
        if (!overflow.empty()) {  // massive ties / duplicates: more rows under the bound than the candidate buffer holds
            if ((rc = resolve_dense_full(h, h->d_q, B, overflow, &flagged[f0], k, ids, dists))) return rc;
        }
    }
    return VS_OK;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

const char* vs_version(void) { return "vsearch-hip 0.1 (gfx950)"; }

int vs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t vs_index_rows(const vs_index* h) { return h ? h->n_total : 0; }
int vs_index_dim(const vs_index* h) { return h ? h->dim : 0; }
int vs_index_nlist(const vs_index* h) { return h ? h->nlist : 0; }
void vs_destroy(vs_index* h) { free_all(h); }

int vs_set_batch(vs_index* h, int batch) {
    if (!h || batch < 1 || batch > vs::kMaxBatch) {
        set_error("batch must be in 1..32");
        return VS_ERR_INVALID;
    }
    h->batch = batch;
    return VS_OK;
}

int vs_prof_enable(vs_index* h, int on) {
    if (!h) return VS_ERR_INVALID;
    h->prof = on != 0;
    for (auto& ps : h->prof_slot) ps.used = 0;
    return VS_OK;
}

int vs_prof_read(vs_index* h, int which, double* total_ms, int64_t* launches) {
    if (!h || which < 0 || which > 1) return VS_ERR_INVALID;
    int rc = set_device(h);
    if (rc) return rc;
    ProfSlot& ps = h->prof_slot[which];
    double tot = 0;
    for (int i = 0; i + 1 < ps.used; i += 2) {
        HIPCHK(hipEventSynchronize(ps.ev[i + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ps.ev[i], ps.ev[i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = ps.used / 2;
    return VS_OK;
}

int vs_prof_read_launches(vs_index* h, int which, double* ms_out, int64_t cap, int64_t* launches) {
    if (!h || which < 0 || which > 1 || (cap > 0 && !ms_out)) return VS_ERR_INVALID;
    int rc = set_device(h);
    if (rc) return rc;
    ProfSlot& ps = h->prof_slot[which];
    int64_t n = 0;
    for (int i = 0; i + 1 < ps.used; i += 2, ++n) {
        if (n >= cap) continue;
        HIPCHK(hipEventSynchronize(ps.ev[i + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ps.ev[i], ps.ev[i + 1]));
        ms_out[n] = ms;
    }
    if (launches) *launches = n;
    return VS_OK;
}

// ------------------------------------------------------------------------------------- brute force
static int bf_create_impl(const float* base_host, int64_t n_rows, int dim, int metric, int device, int64_t id_offset, vs_index** out);
int vs_bf_create(const float* base_host, int64_t n_rows, int dim, int metric, int device, int64_t id_offset,
                 vs_index** out) {
    return guarded([&]() -> int { return bf_create_impl(base_host, n_rows, dim, metric, device, id_offset, out); });
}
static int bf_create_impl(const float* base_host, int64_t n_rows, int dim, int metric, int device, int64_t id_offset,
                          vs_index** out) {
    if (!out || !base_host || n_rows <= 0) {
        set_error("vs_bf_create: bad arguments");
        return VS_ERR_INVALID;
    }
    if (dim != vs::kDim) {
        set_error("only dim == 128 is compiled in");
        return VS_ERR_UNSUPPORTED;
    }
    if (metric != VS_METRIC_L2 && metric != VS_METRIC_IP) {
        set_error("unknown metric");
        return VS_ERR_INVALID;
    }
    if (n_rows + id_offset > std::numeric_limits<int32_t>::max()) {
        set_error("ids must fit int32");
        return VS_ERR_UNSUPPORTED;
    }
    int rc = check_device(device);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    vs_index* h = new (std::nothrow) vs_index();
    if (!h) {
        set_error("out of host memory");
        return VS_ERR_NOMEM;
    }
    h->kind = 0;
    h->device = device;
    h->dim = dim;
    h->metric = metric;
    h->n_rows = h->n_total = n_rows;
    h->id_offset = id_offset;
    if ((rc = upload_vectors(h, base_host, n_rows)) || (rc = alloc_scratch(h))) {
        free_all(h);
        return rc;
    }
    if (metric == VS_METRIC_L2 && (rc = build_u8_copy(h, base_host, n_rows))) {
        free_all(h);
        return rc;
    }
    if ((n_rows + vs::kTileRows - 1) / vs::kTileRows >= 2 * vs::kSeedWaves) {  // shards on which launches are seeded (bf_launch)
        auto sample = [&]() -> int {
            int r2;
            if ((r2 = dev_alloc(&h->d_seed_f32, (size_t)vs::kSeedWaves * 2048)) || (r2 = dev_alloc(&h->d_seed_bnorm, (size_t)vs::kSeedWaves * 16))) return r2;
            if (h->d_vecs_u8 && ((r2 = dev_alloc(&h->d_seed_u8, (size_t)vs::kSeedWaves * 2048)) || (r2 = dev_alloc(&h->d_seed_rterm, (size_t)vs::kSeedWaves * 16))))
                return r2;
            HIPCHK(vs::launch_seed_sample(h->d_vecs, h->d_norm, h->d_vecs_u8, h->d_rterm, n_rows, h->d_seed_f32, h->d_seed_bnorm,
                                          h->d_seed_u8, h->d_seed_rterm, nullptr));
            HIPCHK(hipDeviceSynchronize());
            return VS_OK;
        };
        if ((rc = sample())) {
            free_all(h);
            return rc;
        }
    }
    *out = h;
    return VS_OK;
}

int vs_ivf_set_metric(vs_index* h, int metric) {
    if (!h || h->kind != 1 || (metric != VS_METRIC_L2 && metric != VS_METRIC_IP)) {
        set_error("vs_ivf_set_metric: an IVF index and VS_METRIC_L2 or VS_METRIC_IP");
        return VS_ERR_INVALID;
    }
    h->metric = metric;  // (inner product: centroid scores, bounds, list scan and ranking on -q.v over the fp32 rows)
    return VS_OK;
}

int vs_set_precision(vs_index* h, int precision) {
    if (!h || precision < 0 || precision > 2) {
        set_error("vs_set_precision: 0 = auto, 1 = fp32, 2 = int8");
        return VS_ERR_INVALID;
    }
    if (precision == 2 && !h->d_vecs_u8) {
        set_error("int8 path unavailable: the base is not integer valued in [0, 255] (or the index is not brute-force L2)");
        return VS_ERR_UNSUPPORTED;
    }
    h->precision = precision;
    return VS_OK;
}

int vs_bf_search_dev(vs_index* h, const float* queries_dev, int B, int k, int32_t* ids_dev, float* dists_dev,
                     int32_t* flags_dev, void* stream) {
    if (!h || h->kind != 0 || !queries_dev || !ids_dev || !dists_dev || B < 1 || B > vs::kMaxBatch || k < 1) {
        set_error("vs_bf_search_dev: bad arguments");
        return VS_ERR_INVALID;
    }
    int rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = order_begin(h, st))) return rc;
    rc = bf_batch_dev(h, h->lane[0], queries_dev, B, k + 1, dists_dev, ids_dev, flags_dev ? flags_dev : h->d_flags, st);
    return rc ? rc : order_end(h, st);
}

int vs_bf_search_dev_multi(vs_index* h, const float* queries_dev, int n_batches, int B, int k, int32_t* ids_dev,
                           float* dists_dev, int32_t* flags_dev, void* stream) {
    if (!h || h->kind != 0 || !queries_dev || !ids_dev || !dists_dev || n_batches < 1 || B < 1 || B > vs::kMaxBatch || k < 1) {
        set_error("vs_bf_search_dev_multi: bad arguments");
        return VS_ERR_INVALID;
    }
    int rc = set_device(h);
    if (rc) return rc;
    if (!pick_kcap(k + 1)) {
        set_error("k too large for the compiled scan kernels (k <= 15)");
        return VS_ERR_UNSUPPORTED;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = order_begin(h, st))) return rc;
    rc = bf_multi_dev(h, queries_dev, n_batches, B, k + 1, dists_dev, ids_dev, flags_dev, st);
    return rc ? rc : order_end(h, st);
}

int vs_bf_scores_dev(vs_index* h, const float* queries_dev, int B, float* scores_dev_, int64_t ld, void* stream) {
    if (!h || h->kind != 0 || !queries_dev || !scores_dev_ || B < 1 || B > vs::kMaxBatch || ld < h->n_rows) {
        set_error("vs_bf_scores_dev: bad arguments");
        return VS_ERR_INVALID;
    }
    int rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = order_begin(h, st))) return rc;
    rc = scores_dev(h, h->d_vecs, h->d_norm, h->n_rows, queries_dev, B, scores_dev_, ld, st);
    return rc ? rc : order_end(h, st);
}

int vs_bf_search(vs_index* h, const float* queries_host, int64_t nq, int k, int32_t* ids, float* dists,
                 vs_timing* timing) {
    if (!h || h->kind != 0 || !queries_host || !ids || !dists || nq < 0 || k < 1) {
        set_error("vs_bf_search: bad arguments");
        return VS_ERR_INVALID;
    }
    return guarded([&]() -> int {
        int rc = set_device(h);
        if (rc) return rc;
        const double t_start = now_ms();
        vs_timing tm{};
        const int k1 = k + 1;
        if (!pick_kcap(k1)) {
            set_error("k too large for the compiled scan kernels (k <= 15)");
            return VS_ERR_UNSUPPORTED;
        }
        if ((rc = ensure_pipe(h)) || (rc = order_begin(h, h->stream))) return rc;
        // (a call that failed half way leaves chunks marked in flight: nothing of it may be retired into THIS call's buffers)
        if (h->pipe[0].q0 >= 0 || h->pipe[1].q0 >= 0) {
            (void)hipStreamSynchronize(h->stream);
            (void)hipStreamSynchronize(h->s_h2d);
            (void)hipStreamSynchronize(h->s_d2h);
            h->pipe[0].q0 = h->pipe[1].q0 = -1;
        }
        // Queries go through in chunks of kMaxMulti batches: one persistent scan launch + one merge launch per chunk
        // (the harness loop of main.cpp:201-251 collapsed into a call; a ragged tail batch gets its own launch).  Two
        // chunks are in flight: uploads and downloads run on copy streams beside the other chunk's kernels.
        const int64_t chunk = (int64_t)kMaxMulti * h->batch;
        std::vector<int64_t> flagged;
        const float inf = std::numeric_limits<float>::infinity();
        auto launch_chunk = [&](vs_index::PipeSlot& S, bool force_f32) -> int {
            const int full = (int)(S.n / h->batch), rem = (int)(S.n % h->batch);
            int r2 = VS_OK;
            if (full) r2 = bf_launch(h, h->lane[0], S.d_q, full, h->batch, k1, S.d_out_d, S.d_out_i, S.d_flags, h->stream, force_f32);
            if (!r2 && rem) {
                const size_t o = (size_t)full * h->batch;
                r2 = bf_launch(h, h->lane[0], S.d_q + o * vs::kDim, 1, rem, k1, S.d_out_d + o * k1, S.d_out_i + o * k1, S.d_flags + o,
                               h->stream, force_f32);
            }
            return r2;
        };
        auto download = [&](vs_index::PipeSlot& S, hipStream_t st) -> int {
            float* hd = reinterpret_cast<float*>(S.pin_out);
            int32_t* hi = reinterpret_cast<int32_t*>(S.pin_out) + (size_t)chunk * k1;
            int32_t* hf = hi + (size_t)chunk * k1;
            HIPCHK(hipMemcpyAsync(hd, S.d_out_d, (size_t)S.n * k1 * sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(hi, S.d_out_i, (size_t)S.n * k1 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(hf, S.d_flags, (size_t)S.n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            return VS_OK;
        };
        auto enqueue = [&](vs_index::PipeSlot& S, int64_t q0, int64_t n) -> int {
            const double t0 = now_ms();
            S.q0 = q0;
            S.n = n;
            std::memcpy(S.pin_q, queries_host + q0 * vs::kDim, (size_t)n * vs::kDim * sizeof(float));
            HIPCHK(hipMemcpyAsync(S.d_q, S.pin_q, (size_t)n * vs::kDim * sizeof(float), hipMemcpyHostToDevice, h->s_h2d));
            HIPCHK(hipEventRecord(S.ev_h2d, h->s_h2d));
            tm.h2d_ms += now_ms() - t0;
            HIPCHK(hipStreamWaitEvent(h->stream, S.ev_h2d, 0));
            int r2 = launch_chunk(S, false);
            if (r2) return r2;
            HIPCHK(hipEventRecord(S.ev_comp, h->stream));
            HIPCHK(hipStreamWaitEvent(h->s_d2h, S.ev_comp, 0));
            if ((r2 = download(S, h->s_d2h))) return r2;
            HIPCHK(hipEventRecord(S.ev_d2h, h->s_d2h));
            return VS_OK;
        };
        auto retire = [&](vs_index::PipeSlot& S) -> int {
            if (S.q0 < 0) return VS_OK;
            const double t0 = now_ms();
            HIPCHK(hipEventSynchronize(S.ev_d2h));
            const float* hd = reinterpret_cast<const float*>(S.pin_out);
            const int32_t* hi = reinterpret_cast<const int32_t*>(S.pin_out) + (size_t)chunk * k1;
            const int32_t* hf = hi + (size_t)chunk * k1;
            bool rerun = false;
            for (int64_t b = 0; b < S.n; ++b) rerun = rerun || hf[(size_t)b] == 2;
            if (rerun) {
                // a query of this chunk is not an integer in [0, 255]: the int8 scan skipped its batch -> fp32 path
                int r2 = launch_chunk(S, true);
                if (r2) return r2;
                if ((r2 = download(S, h->stream))) return r2;
                HIPCHK(hipStreamSynchronize(h->stream));
            }
            tm.d2h_ms += now_ms() - t0;
            for (int64_t b = 0; b < S.n; ++b) {
                for (int t = 0; t < k; ++t) {
                    const int32_t id = hi[(size_t)b * k1 + t];
                    ids[(S.q0 + b) * k + t] = id;
                    float d = id >= 0 ? hd[(size_t)b * k1 + t] : inf;
                    if (h->metric == VS_METRIC_IP && id >= 0) d = -d;
                    dists[(S.q0 + b) * k + t] = d;
                }
                if (hf[(size_t)b]) flagged.push_back(S.q0 + b);
            }
            S.q0 = -1;
            return VS_OK;
        };
        int c = 0;
        for (int64_t q0 = 0; q0 < nq; q0 += chunk, ++c) {
            vs_index::PipeSlot& S = h->pipe[c & 1];
            if ((rc = retire(S))) return rc;
            if ((rc = enqueue(S, q0, std::min<int64_t>(chunk, nq - q0)))) return rc;
        }
        if ((rc = retire(h->pipe[c & 1]))) return rc;        // the older chunk first
        if ((rc = retire(h->pipe[(c + 1) & 1]))) return rc;
        tm.fine_search_ms = now_ms() - t_start;
        // Ties inside the k+1 best: the reference's order is history dependent (cpu_baseline.cpp:127-153) -> replay
        // select_topk over the rows that can change its buffer (resolve_ties).
        if (!flagged.empty() && h->metric == VS_METRIC_L2) {
            const double t0 = now_ms();
            if ((rc = resolve_ties(h, queries_host, flagged, k, ids, dists))) return rc;
            tm.tie_resolve_ms = now_ms() - t0;
            tm.tie_queries = (int64_t)flagged.size();
        }
        tm.total_ms = now_ms() - t_start;
        if (timing) *timing = tm;
        return order_end(h, h->stream);
    });
}

// --------------------------------------------------------------------------------------------- IVF
int vs_ivf_list_owners(const int32_t* cluster_offsets, int nlist, int world, int32_t* owner_out) {
    if (!cluster_offsets || !owner_out || nlist <= 0 || world < 1) {
        set_error("vs_ivf_list_owners: bad arguments");
        return VS_ERR_INVALID;
    }
    std::vector<int> order(nlist);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return (cluster_offsets[a + 1] - cluster_offsets[a]) > (cluster_offsets[b + 1] - cluster_offsets[b]);
    });
    for (int i = 0; i < nlist; ++i) owner_out[order[i]] = i % world;
    return VS_OK;
}

int vs_ivf_shard_group(int world) {
    if (world < 1) return 0;
    return world > 1 ? std::min(kIvfGroupMax, 32 * std::min(world, kIvfShardMaxWorld)) : kIvfGroupDefault;
}

int vs_ivf_shard_slice(int n_batches, int world, int rank, int32_t* slice_batches, int32_t* first_batch, int32_t* own_batches) {
    if (n_batches < 1 || world < 1 || rank < 0 || rank >= world || !slice_batches || !first_batch || !own_batches) {
        set_error("vs_ivf_shard_slice: bad arguments");
        return VS_ERR_INVALID;
    }
    int sbb, b0, nbs;
    ivf_slice(n_batches, world, rank, sbb, b0, nbs);
    if (sbb > vs::kIvfWideBatches) {
        set_error("vs_ivf_shard_slice: more than 32 batches per slice (a launch group holds at most vs_ivf_shard_group(world) batches)");
        return VS_ERR_INVALID;
    }
    *slice_batches = sbb;
    *first_batch = b0;
    *own_batches = nbs;
    return VS_OK;
}

int64_t vs_ivf_shard_block_words(int slice_batches, int nprobe) { return ivf_block_words(slice_batches, nprobe); }

// the replicated heads of a sharded index (see vs_index::d_head_vecs): rows [offsets[c], offsets[c] + min(len, kIvfTauRows))
// of every list of the WHOLE index, fp32 packed + (when every head row is byte valued) the tiled byte copy
static int build_tau_heads(vs_index* h, const float* vectors, const int32_t* offsets, int nlist) {
    std::vector<int32_t> hoff((size_t)nlist + 1, 0), tdelta((size_t)nlist, 0);
    int64_t t = 0;
    for (int c = 0; c < nlist; ++c) {
        const int32_t len = std::min<int32_t>(offsets[c + 1] - offsets[c], vs::kIvfTauRows);
        hoff[c + 1] = hoff[c] + len;
        tdelta[c] = (int32_t)t - hoff[c];
        t += (len + 15) / 16 * 16;
    }
    const size_t nh = (size_t)hoff[nlist], nt = (size_t)t + 64;
    std::vector<float> hv((nh + vs::kScanPadRows) * vs::kDim, 0.f);
    for (int c = 0; c < nlist; ++c)
        if (hoff[c + 1] > hoff[c])
            std::memcpy(&hv[(size_t)hoff[c] * vs::kDim], vectors + (size_t)offsets[c] * vs::kDim, (size_t)(hoff[c + 1] - hoff[c]) * vs::kDim * sizeof(float));
    int rc;
    if ((rc = dev_alloc(&h->d_head_vecs, hv.size())) || (rc = dev_alloc(&h->d_head_norm, nh + 64)) || (rc = dev_alloc(&h->d_head_off, hoff.size())) ||
        (rc = dev_alloc(&h->d_head_tdelta, tdelta.size())))
        return rc;
    HIPCHK(hipMemcpy(h->d_head_vecs, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(h->d_head_norm, 0, (nh + 64) * sizeof(float)));
    HIPCHK(hipMemcpy(h->d_head_off, hoff.data(), hoff.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_head_tdelta, tdelta.data(), tdelta.size() * 4, hipMemcpyHostToDevice));
    if (nh > 0) HIPCHK(vs::launch_row_sqnorm(h->d_head_vecs, (int64_t)nh, vs::kDim, h->d_head_norm, nullptr));
    HIPCHK(hipDeviceSynchronize());
    if (h->metric != VS_METRIC_L2) return VS_OK;
    std::vector<int8_t> tb(nt * vs::kDim, 0);
    std::vector<int32_t> rterm_t(nt, 0);
    for (int c = 0; c < nlist; ++c)
        for (int32_t j = 0; j < hoff[c + 1] - hoff[c]; ++j) {
            const float* src = &hv[((size_t)hoff[c] + j) * vs::kDim];
            const size_t R = (size_t)hoff[c] + tdelta[c] + j;
            int8_t* tile = &tb[(R >> 4) * 16 * vs::kDim + (R & 15) * 16];
            int32_t n2 = 0, sb = 0;
            for (int e = 0; e < vs::kDim; ++e) {
                const float x = src[e];
                const int xi = (int)x;
                if (!((float)xi == x) || xi < 0 || xi > 255) return VS_OK;  // not byte valued: bounds on the fp32 heads
                tile[(e >> 6) * 1024 + ((e >> 4) & 3) * 256 + (e & 15)] = (int8_t)(xi - 128);
                n2 += xi * xi;
                sb += xi - 128;
            }
            rterm_t[R] = n2 - 256 * sb;
        }
    if ((rc = dev_alloc(&h->d_head_t8, tb.size())) || (rc = dev_alloc(&h->d_head_rterm_t, nt))) return rc;
    HIPCHK(hipMemcpy(h->d_head_t8, tb.data(), tb.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_head_rterm_t, rterm_t.data(), nt * 4, hipMemcpyHostToDevice));
    return VS_OK;
}

static int ivf_create_impl(const float* vectors, int64_t n_rows, int dim, const float* centroids, int nlist,
                           const int32_t* offsets, const int32_t* r2o, int device, int rank, int world,
                           vs_index** out) {
    if (!out || !vectors || !centroids || !offsets || n_rows <= 0 || nlist <= 0 || world < 1 || rank < 0 || rank >= world) {
        set_error("vs_ivf_create: bad arguments");
        return VS_ERR_INVALID;
    }
    if (dim != vs::kDim) {
        set_error("only dim == 128 is compiled in");
        return VS_ERR_UNSUPPORTED;
    }
    if (offsets[0] != 0 || offsets[nlist] != n_rows) {
        set_error("cluster_offsets do not cover the vectors");
        return VS_ERR_INVALID;
    }
    for (int c = 0; c < nlist; ++c)
        if (offsets[c + 1] < offsets[c]) {
            set_error("cluster_offsets not monotone");
            return VS_ERR_INVALID;
        }
    int rc = check_device(device);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    vs_index* h = new (std::nothrow) vs_index();
    if (!h) {
        set_error("out of host memory");
        return VS_ERR_NOMEM;
    }
    h->kind = 1;
    h->device = device;
    h->dim = dim;
    h->metric = VS_METRIC_L2;
    h->n_total = n_rows;
    h->nlist = nlist;
    h->rank = rank;
    h->world = world;
    h->h_offsets_global.assign(offsets, offsets + nlist + 1);
    h->avg_cluster_size = (double)n_rows / nlist;

    // Ownership: lists sorted by length (desc), dealt round-robin -> balanced bytes and probe hits
    // (SURVEY.md 8e).  Only owned lists are made resident; the others become empty ranges.
    std::vector<int32_t> owner(nlist);
    vs_ivf_list_owners(offsets, nlist, world, owner.data());
    std::vector<uint8_t> owned(nlist, 0);
    for (int c = 0; c < nlist; ++c) owned[c] = owner[c] == rank;
    std::vector<int32_t> loc_off(nlist + 1, 0);
    for (int c = 0; c < nlist; ++c) loc_off[c + 1] = loc_off[c] + (owned[c] ? offsets[c + 1] - offsets[c] : 0);
    const int64_t n_local = loc_off[nlist];
    h->n_rows = n_local;
    std::vector<int32_t> loc_r2o((size_t)std::max<int64_t>(n_local, 1));
    const float* up = vectors;
    std::vector<float> packed;
    if (world > 1) {
        packed.resize((size_t)std::max<int64_t>(n_local, 1) * dim);
        for (int c = 0; c < nlist; ++c)
            if (owned[c] && offsets[c + 1] > offsets[c])
                std::memcpy(&packed[(size_t)loc_off[c] * dim], vectors + (size_t)offsets[c] * dim,
                            (size_t)(offsets[c + 1] - offsets[c]) * dim * sizeof(float));
        up = packed.data();
    }
    for (int c = 0; c < nlist; ++c)
        if (owned[c])
            for (int32_t r = offsets[c]; r < offsets[c + 1]; ++r)
                loc_r2o[(size_t)loc_off[c] + (r - offsets[c])] = r2o ? r2o[r] : r;

    auto fail = [&](int code) {
        free_all(h);
        return code;
    };
    if ((rc = upload_vectors(h, up, n_local))) return fail(rc);
    // exact int8 copy of the (reordered, local) rows when they are byte valued: the list scan then moves 4x fewer bytes
    std::vector<int8_t> host_bytes;
    std::vector<int32_t> host_rterm;
    if (h->metric == VS_METRIC_L2 && n_local > 0 && (rc = build_u8_copy(h, up, n_local, &host_bytes, &host_rterm))) return fail(rc);
    if ((rc = dev_alloc(&h->d_centroids, ((size_t)nlist + vs::kScanPadRows) * dim))) return fail(rc);
    if (hipMemset(h->d_centroids + (size_t)nlist * dim, 0, (size_t)vs::kScanPadRows * dim * sizeof(float)) != hipSuccess) return fail(VS_ERR_DEVICE);
    if ((rc = dev_alloc(&h->d_cnorm, (size_t)nlist + 64))) return fail(rc);
    if ((rc = dev_alloc(&h->d_offsets, (size_t)nlist + 1))) return fail(rc);
    if ((rc = dev_alloc(&h->d_r2o, (size_t)std::max<int64_t>(n_local, 1)))) return fail(rc);
    hipError_t e;
    if ((e = hipMemset(h->d_cnorm, 0, ((size_t)nlist + 64) * sizeof(float))) != hipSuccess ||
        (e = hipMemcpy(h->d_centroids, centroids, (size_t)nlist * dim * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(h->d_offsets, loc_off.data(), ((size_t)nlist + 1) * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(h->d_r2o, loc_r2o.data(), (size_t)std::max<int64_t>(n_local, 1) * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = vs::launch_row_sqnorm(h->d_centroids, nlist, dim, h->d_cnorm, nullptr)) != hipSuccess ||
        (e = hipDeviceSynchronize()) != hipSuccess) {
        set_error(std::string("ivf upload: ") + hipGetErrorString(e));
        return fail(VS_ERR_DEVICE);
    }
    {
        // (list, 1024-row chunk) work items of the list-major scan, resident lists only
        std::vector<int32_t> cl, cr0, crn;
        int32_t mx = 0;
        for (int c = 0; c < nlist; ++c) {
            const int32_t n = loc_off[c + 1] - loc_off[c];
            mx = std::max(mx, n);
            for (int32_t r = 0; r < n; r += 1024) {
                cl.push_back(c);
                cr0.push_back(loc_off[c] + r);
                crn.push_back(std::min<int32_t>(1024, n - r));
            }
        }
        h->n_chunks = (int)cl.size();
        h->max_list = mx;
        if (h->n_chunks > 0 && h->d_vecs_u8) {
            // the tiled, padded copy for the wide scan (see vs_index::d_vecs_t8)
            std::vector<int32_t> toff((size_t)nlist + 1), tdelta((size_t)nlist), ctr0;
            int64_t t = 0;
            for (int c = 0; c < nlist; ++c) {
                toff[c] = (int32_t)t;
                tdelta[c] = (int32_t)t - loc_off[c];
                t += ((int64_t)(loc_off[c + 1] - loc_off[c]) + vs::kIvfWideUnit - 1) / vs::kIvfWideUnit * vs::kIvfWideUnit;
            }
            toff[nlist] = (int32_t)t;
            if (t + 64 >= (1ll << 31)) {  // (padded rows are int32 like rows)
                set_error("ivf: too many rows for one shard");
                return fail(VS_ERR_UNSUPPORTED);
            }
            const size_t n_t = (size_t)t + 64;
            std::vector<int8_t> tb(n_t * vs::kDim, 0);
            std::vector<int32_t> nrh_t(n_t, 0), rterm_t(n_t, 0), r2o_t(n_t, -1);
            for (int c = 0; c < nlist; ++c)
                for (int32_t j = 0; j < loc_off[c + 1] - loc_off[c]; ++j) {
                    const size_t R = (size_t)toff[c] + j, row = (size_t)loc_off[c] + j;
                    const int8_t* src = &host_bytes[row * vs::kDim];
                    int8_t* tile = &tb[(R >> 4) * 16 * vs::kDim + (R & 15) * 16];
                    for (int ch = 0; ch < 8; ++ch) std::memcpy(tile + (ch >> 2) * 1024 + (ch & 3) * 256, src + 16 * ch, 16);
                    rterm_t[R] = host_rterm[row];
                    nrh_t[R] = -(host_rterm[row] >> 1);
                    r2o_t[R] = loc_r2o[row];
                }
            for (size_t i = 0; i < cl.size(); ++i) ctr0.push_back(cr0[i] + tdelta[cl[i]]);
            if ((rc = dev_alloc(&h->d_vecs_t8, tb.size())) || (rc = dev_alloc(&h->d_nrh_t, n_t)) || (rc = dev_alloc(&h->d_rterm_t, n_t)) ||
                (rc = dev_alloc(&h->d_r2o_t, n_t)) || (rc = dev_alloc(&h->d_tdelta, tdelta.size())) || (rc = dev_alloc(&h->d_chunk_trow0, ctr0.size())))
                return fail(rc);
            if ((e = hipMemcpy(h->d_vecs_t8, tb.data(), tb.size(), hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(h->d_nrh_t, nrh_t.data(), n_t * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(h->d_rterm_t, rterm_t.data(), n_t * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(h->d_r2o_t, r2o_t.data(), n_t * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(h->d_tdelta, tdelta.data(), tdelta.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(h->d_chunk_trow0, ctr0.data(), ctr0.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) {
                set_error(std::string("ivf tiled copy: ") + hipGetErrorString(e));
                return fail(VS_ERR_DEVICE);
            }
        }
        if (h->n_chunks > 0) {
            if ((rc = dev_alloc(&h->d_chunk_list, cl.size()))) return fail(rc);
            if ((rc = dev_alloc(&h->d_chunk_row0, cl.size()))) return fail(rc);
            if ((rc = dev_alloc(&h->d_chunk_rows, cl.size()))) return fail(rc);
            h->n_units_max = 0;
            for (int32_t r : crn) h->n_units_max += (r + 31) >> 5;
            if ((e = hipMemcpy(h->d_chunk_list, cl.data(), cl.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(h->d_chunk_row0, cr0.data(), cl.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(h->d_chunk_rows, crn.data(), cl.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) {
                set_error(std::string("ivf chunk table: ") + hipGetErrorString(e));
                return fail(VS_ERR_DEVICE);
            }
        }
    }
    if (world > 1 && (rc = build_tau_heads(h, vectors, offsets, nlist))) return fail(rc);
    if ((rc = alloc_scratch(h))) return fail(rc);
    // launch groups: an unsharded index takes VSEARCH_IVF_GROUP batches per group (super-batches of 32); a sharded one a
    // slice of up to 32 batches per rank (ivf_shard_front / ivf_shard_back)
    h->ivf_gb = world > 1 ? std::min(kIvfGroupMax, 32 * std::min(world, kIvfShardMaxWorld)) : ivf_group_batches();
    h->ivf_lanes = ivf_wide_lanes();
    h->ivf_nsb = world > 1 ? std::max(h->ivf_gb / 32, std::min(world, kIvfShardMaxWorld)) : h->ivf_gb / 32;
    if (h->nlist <= vs::kIvfFastNlist && h->n_chunks > 0) {
        // the wide pipeline's scratch (two lanes), streams and host staging now rather than inside the first search:
        // index load is outside every timed region, a first call that allocates 200 MB is not (the reference's harness
        // times every searchBatch call, main_ivf.cpp:157-163)
        if ((rc = ensure_ivf_wide(h, 0)) || (rc = ensure_ivf_wide(h, 1)) || (rc = ensure_wide_streams(h)) || (rc = ensure_ivf_host(h)))
            return fail(rc);
    }
    *out = h;
    return VS_OK;
}

// GPU index builder: Lloyd k-means on the scan kernel (assignment = the brute-force MFMA scan with the
// centroids as "queries", 32 per pass) + deterministic fixed-point update.
static int ivf_build_impl(const float* base_host, int64_t n_rows, int dim, int nlist, int max_iter, double tol, uint64_t seed,
                          int device, float* centroids_out, int32_t* assign_out, int* iters_done);
int vs_ivf_build(const float* base_host, int64_t n_rows, int dim, int nlist, int max_iter, double tol, uint64_t seed,
                 int device, float* centroids_out, int32_t* assign_out, int* iters_done) {
    return guarded([&]() -> int {
        return ivf_build_impl(base_host, n_rows, dim, nlist, max_iter, tol, seed, device, centroids_out, assign_out, iters_done);
    });
}
static int ivf_build_impl(const float* base_host, int64_t n_rows, int dim, int nlist, int max_iter, double tol, uint64_t seed,
                          int device, float* centroids_out, int32_t* assign_out, int* iters_done) {
    if (!base_host || !centroids_out || !assign_out || n_rows <= 0 || nlist <= 0 || nlist > n_rows || max_iter < 0) {
        set_error("vs_ivf_build: bad arguments");
        return VS_ERR_INVALID;
    }
    if (dim != vs::kDim) {
        set_error("only dim == 128 is compiled in");
        return VS_ERR_UNSUPPORTED;
    }
    int rc = check_device(device);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const int num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    float *d_x = nullptr, *d_norm = nullptr, *d_cents = nullptr, *d_best_d = nullptr;
    int32_t *d_best_i = nullptr, *d_counts = nullptr;
    unsigned long long* d_acc = nullptr;
    double* d_shift = nullptr;
    auto cleanup = [&]() {
        void* ptrs[] = {d_x, d_norm, d_cents, d_best_d, d_best_i, d_counts, d_acc, d_shift};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    };
#define BUILD_CHK(expr)                                                                \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));              \
            cleanup();                                                                 \
            return VS_ERR_DEVICE;                                                      \
        }                                                                              \
    } while (0)
    const int nlist_pad = (nlist + 31) & ~31;
    BUILD_CHK(hipMalloc(&d_x, ((size_t)n_rows + vs::kScanPadRows) * dim * sizeof(float)));
    BUILD_CHK(hipMemset(d_x + (size_t)n_rows * dim, 0, (size_t)vs::kScanPadRows * dim * sizeof(float)));
    BUILD_CHK(hipMalloc(&d_norm, ((size_t)n_rows + 64) * sizeof(float)));
    BUILD_CHK(hipMalloc(&d_cents, (size_t)nlist_pad * dim * sizeof(float)));
    BUILD_CHK(hipMalloc(&d_best_d, (size_t)n_rows * sizeof(float)));
    BUILD_CHK(hipMalloc(&d_best_i, (size_t)n_rows * sizeof(int32_t)));
    BUILD_CHK(hipMalloc(&d_counts, (size_t)nlist * sizeof(int32_t)));
    BUILD_CHK(hipMalloc(&d_acc, (size_t)nlist * dim * sizeof(unsigned long long)));
    BUILD_CHK(hipMalloc(&d_shift, (size_t)nlist * sizeof(double)));
    BUILD_CHK(hipMemcpy(d_x, base_host, (size_t)n_rows * dim * sizeof(float), hipMemcpyHostToDevice));
    BUILD_CHK(hipMemset(d_norm, 0, ((size_t)n_rows + 64) * sizeof(float)));
    BUILD_CHK(vs::launch_row_sqnorm(d_x, n_rows, dim, d_norm, nullptr));
    // initial centroids: k-means++ (D^2 sampling), sklearn's default for the reference's KMeans(random_state=42, n_init=1),
    // create_ivf_model_reordered.py:97-103.  sklearn's RNG stream and its greedy multi-trial variant are not
    // reproduced (VSEARCH_KMEANS_INIT=random: nlist distinct random rows instead).
    {
        uint64_t st = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        auto next = [&]() {
            uint64_t z = (st += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        const char* init_env = getenv("VSEARCH_KMEANS_INIT");
        const bool random_init = init_env && std::string(init_env) == "random";
        BUILD_CHK(hipMemset(d_cents, 0, (size_t)nlist_pad * dim * sizeof(float)));
        if (random_init) {
            std::vector<float> init((size_t)nlist_pad * dim, 0.f);
            std::vector<bool> used((size_t)n_rows, false);
            for (int c = 0; c < nlist; ++c) {
                int64_t r;
                do r = (int64_t)(next() % (uint64_t)n_rows); while (used[(size_t)r]);
                used[(size_t)r] = true;
                std::memcpy(&init[(size_t)c * dim], base_host + r * dim, (size_t)dim * sizeof(float));
            }
            BUILD_CHK(hipMemcpy(d_cents, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice));
        } else {
            const int n_blocks = (int)((n_rows + vs::kKppBlockRows - 1) / vs::kKppBlockRows);
            double* d_bsum = nullptr;
            BUILD_CHK(hipMalloc(&d_bsum, (size_t)n_blocks * sizeof(double)));
            const int64_t first = (int64_t)(next() % (uint64_t)n_rows);
            hipError_t e = hipMemcpy(d_cents, d_x + first * dim, (size_t)dim * sizeof(float), hipMemcpyDeviceToDevice);
            // d_best_d doubles as the running min squared distance (+inf to start with)
            if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_best_d), 0x7f800000, (size_t)n_rows, nullptr);
            for (int c = 1; c < nlist && e == hipSuccess; ++c) {
                const double u = (double)(next() >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
                e = vs::launch_kpp_step(d_x, d_norm, n_rows, d_cents, c, d_best_d, d_bsum, n_blocks, u, nullptr);
            }
            if (e == hipSuccess) e = hipDeviceSynchronize();
            (void)hipFree(d_bsum);
            BUILD_CHK(e);
        }
    }
    // sklearn's stopping rule: sum of squared centre shifts <= tol * mean per-feature variance
    double tol_abs = 0.0;
    if (tol > 0) {
        std::vector<double> sum(dim, 0.0), sq(dim, 0.0);
        for (int64_t i = 0; i < n_rows; ++i)
            for (int t = 0; t < dim; ++t) {
                const double v = base_host[i * dim + t];
                sum[t] += v;
                sq[t] += v * v;
            }
        double mv = 0;
        for (int t = 0; t < dim; ++t) {
            const double m = sum[t] / n_rows;
            mv += sq[t] / n_rows - m * m;
        }
        tol_abs = tol * mv / dim;
    }
    int grid, tp;
    scan_geometry(n_rows, num_cus, grid, tp);
    auto assign_pass = [&]() -> hipError_t {
        hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_best_d), 0x7f800000, (size_t)n_rows, nullptr);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(d_best_i, 0xff, (size_t)n_rows * sizeof(int32_t), nullptr);
        if (e != hipSuccess) return e;
        vs::ScanParams p{};
        p.base = d_x;
        p.bnorm = d_norm;
        p.metric = 0;
        p.row_begin = 0;
        p.row_end = n_rows;
        p.tiles_per_wg = tp;
        p.best_d = d_best_d;
        p.best_i = d_best_i;
        p.q_batch_stride = (int64_t)32 * dim;
        const int full = nlist / 32, rem = nlist % 32;
        if (full) {
            p.q = d_cents;
            p.n_batches = full;
            p.nq_valid = 32;
            p.assign_base = 0;
            e = vs::launch_scan(p, grid, 8, 2, vs::kModeAssign, nullptr);
            if (e != hipSuccess) return e;
        }
        if (rem) {
            p.q = d_cents + (size_t)full * 32 * dim;
            p.n_batches = 1;
            p.nq_valid = rem;
            p.assign_base = full * 32;
            e = vs::launch_scan(p, grid, 8, 2, vs::kModeAssign, nullptr);
        }
        return e;
    };
    int it = 0;
    std::vector<double> shift((size_t)nlist);
    for (; it < max_iter; ++it) {
        BUILD_CHK(assign_pass());
        BUILD_CHK(vs::launch_kmeans_update(d_x, d_best_i, n_rows, nlist, d_cents, d_acc, d_counts, d_shift, nullptr));
        BUILD_CHK(hipMemcpy(shift.data(), d_shift, (size_t)nlist * sizeof(double), hipMemcpyDeviceToHost));
        double total = 0;
        for (double v : shift) total += v;
        if (total <= tol_abs) {
            ++it;
            break;
        }
    }
    BUILD_CHK(assign_pass());  // labels consistent with the final centroids
    BUILD_CHK(hipMemcpy(assign_out, d_best_i, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToHost));
    BUILD_CHK(hipMemcpy(centroids_out, d_cents, (size_t)nlist * dim * sizeof(float), hipMemcpyDeviceToHost));
#undef BUILD_CHK
    cleanup();
    if (iters_done) *iters_done = it;
    return VS_OK;
}

// build_ivf_index_reordered (create_ivf_model_reordered.py:82-177) end to end: k-means, reordered layout, resident index.
int vs_ivf_build_index(const float* base_host, int64_t n_rows, int dim, int nlist, int max_iter, double tol, uint64_t seed,
                       int device, vs_index** out, int* iters_done) {
    if (!out || !base_host || n_rows <= 0 || nlist <= 0) {
        set_error("vs_ivf_build_index: bad arguments");
        return VS_ERR_INVALID;
    }
    return guarded([&]() -> int {
        const int nl = vs_ivf_clamp_nlist(n_rows, nlist);
        std::vector<float> cents((size_t)nl * dim);
        std::vector<int32_t> assign((size_t)n_rows), off((size_t)nl + 1), r2o((size_t)n_rows);
        int rc = vs_ivf_build(base_host, n_rows, dim, nl, max_iter, tol, seed, device, cents.data(), assign.data(), iters_done);
        if (rc) return rc;
        if ((rc = vs_ivf_layout(assign.data(), n_rows, nl, off.data(), r2o.data()))) return rc;
        std::vector<float> vr((size_t)n_rows * dim);
        for (int64_t i = 0; i < n_rows; ++i)
            std::memcpy(&vr[(size_t)i * dim], base_host + (size_t)r2o[(size_t)i] * dim, (size_t)dim * sizeof(float));
        return ivf_create_impl(vr.data(), n_rows, dim, cents.data(), nl, off.data(), r2o.data(), device, 0, 1, out);
    });
}

int vs_ivf_create(const float* vectors_reordered, int64_t n_rows, int dim, const float* centroids, int nlist,
                  const int32_t* cluster_offsets, const int32_t* reorder_to_original, int device, int rank, int world,
                  vs_index** out) {
    return guarded([&]() -> int {
        return ivf_create_impl(vectors_reordered, n_rows, dim, centroids, nlist, cluster_offsets, reorder_to_original, device, rank,
                               world, out);
    });
}

int vs_ivf_load(const char* index_dir, int device, int rank, int world, vs_index** out) {
    if (!index_dir || !out) {
        set_error("vs_ivf_load: bad arguments");
        return VS_ERR_INVALID;
    }
    return guarded([&]() -> int {
    const std::string dir(index_dir);
    vs::IvfConfig cfg;
    if (!vs::ivf_config_read(dir + "/ivf_config.json", cfg)) return VS_ERR_IO;
    std::vector<int32_t> offsets, r2o;
    std::vector<float> vecs, cents;
    std::vector<int64_t> shp;
    if (!vs::npy_read_i32(dir + "/cluster_offsets.npy", offsets, shp)) return VS_ERR_IO;  // IVFIndex.cpp:210-213
    if (!vs::npy_read_f32(dir + "/centroids.npy", cents, shp)) return VS_ERR_IO;
    if (shp.size() != 2 || shp[0] != cfg.n_clusters || shp[1] != cfg.dim) {
        set_error("centroids.npy shape does not match ivf_config.json");
        return VS_ERR_IO;
    }
    if ((int64_t)offsets.size() != cfg.n_clusters + 1) {
        set_error("cluster_offsets.npy length != n_clusters + 1");
        return VS_ERR_IO;
    }
    if (cfg.reordered) {
        if (!vs::npy_read_i32(dir + "/reorder_to_original.npy", r2o, shp)) return VS_ERR_IO;  // IVFIndex.cpp:225-228
        if (!vs::npy_read_f32(dir + "/vectors_reordered.npy", vecs, shp)) return VS_ERR_IO;   // IVFIndex.cpp:239-244
    } else {
        // plain mode (IVFIndex.cpp:216-222,247-252): gather into the contiguous layout at load time
        std::vector<int32_t> cidx;
        std::vector<float> plain;
        if (!vs::npy_read_i32(dir + "/cluster_indices.npy", cidx, shp)) return VS_ERR_IO;
        if (!vs::npy_read_f32(dir + "/vectors.npy", plain, shp)) return VS_ERR_IO;
        if (shp.size() != 2 || shp[1] != cfg.dim) {
            set_error("vectors.npy shape mismatch");
            return VS_ERR_IO;
        }
        const int64_t n = (int64_t)cidx.size();
        vecs.resize((size_t)n * cfg.dim);
        r2o = cidx;
        for (int64_t i = 0; i < n; ++i) {
            if (cidx[i] < 0 || cidx[i] >= shp[0]) {
                set_error("cluster_indices.npy out of range");
                return VS_ERR_IO;
            }
            std::memcpy(&vecs[(size_t)i * cfg.dim], &plain[(size_t)cidx[i] * cfg.dim], (size_t)cfg.dim * sizeof(float));
        }
    }
    const int64_t n = (int64_t)vecs.size() / std::max<int64_t>(cfg.dim, 1);
    if (n != cfg.n_vectors || (int64_t)r2o.size() != n) {
        set_error("vector count does not match ivf_config.json");
        return VS_ERR_IO;
    }
    for (int32_t v : r2o)
        if (v < 0 || v >= n) {
            set_error("reorder_to_original.npy holds an id outside [0, n_vectors)");
            return VS_ERR_IO;
        }
    int rc = ivf_create_impl(vecs.data(), n, (int)cfg.dim, cents.data(), (int)cfg.n_clusters, offsets.data(), r2o.data(),
                             device, rank, world, out);
    if (rc == VS_OK && cfg.avg_cluster_size > 0) (*out)->avg_cluster_size = cfg.avg_cluster_size;
    return rc;
    });
}

int vs_ivf_save(vs_index* h, const char* index_dir) {
    if (!h || h->kind != 1 || !index_dir) {
        set_error("vs_ivf_save: bad arguments");
        return VS_ERR_INVALID;
    }
    if (h->world != 1) {
        set_error("vs_ivf_save needs an unsharded index");
        return VS_ERR_UNSUPPORTED;
    }
    int rc = set_device(h);
    if (rc) return rc;
    const std::string dir(index_dir);
    std::vector<float> vecs((size_t)h->n_rows * h->dim), cents((size_t)h->nlist * h->dim);
    std::vector<int32_t> r2o((size_t)h->n_rows);
    HIPCHK(hipMemcpy(vecs.data(), h->d_vecs, vecs.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cents.data(), h->d_centroids, cents.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(r2o.data(), h->d_r2o, r2o.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int32_t> sizes(h->nlist);
    int32_t mn = std::numeric_limits<int32_t>::max(), mx = 0;
    for (int c = 0; c < h->nlist; ++c) {
        sizes[c] = h->h_offsets_global[c + 1] - h->h_offsets_global[c];
        mn = std::min(mn, sizes[c]);
        mx = std::max(mx, sizes[c]);
    }
    vs::IvfConfig cfg;
    cfg.n_vectors = h->n_rows;
    cfg.n_clusters = h->nlist;
    cfg.dim = h->dim;
    cfg.batch_size = h->batch;
    cfg.avg_cluster_size = (double)h->n_rows / h->nlist;
    cfg.min_cluster_size = mn;
    cfg.max_cluster_size = mx;
    cfg.reordered = true;
    if (!vs::ivf_config_write(dir + "/ivf_config.json", cfg)) return VS_ERR_IO;
    if (!vs::npy_write(dir + "/vectors_reordered.npy", vecs.data(), "<f4", {h->n_rows, h->dim}, 4)) return VS_ERR_IO;
    if (!vs::npy_write(dir + "/reorder_to_original.npy", r2o.data(), "<i4", {h->n_rows}, 4)) return VS_ERR_IO;
    if (!vs::npy_write(dir + "/cluster_offsets.npy", h->h_offsets_global.data(), "<i4", {h->nlist + 1}, 4)) return VS_ERR_IO;
    if (!vs::npy_write(dir + "/cluster_sizes.npy", sizes.data(), "<i4", {h->nlist}, 4)) return VS_ERR_IO;
    if (!vs::npy_write(dir + "/centroids.npy", cents.data(), "<f4", {h->nlist, h->dim}, 4)) return VS_ERR_IO;
    return VS_OK;
}

int vs_ivf_search_dev(vs_index* h, const float* queries_dev, int B, int k, int nprobe, int32_t* ids_dev,
                      float* dists_dev, void* stream) {
    if (!h || h->kind != 1 || !queries_dev || !ids_dev || !dists_dev || B < 1 || B > vs::kMaxBatch || k < 1 || nprobe < 1) {
        set_error("vs_ivf_search_dev: bad arguments");
        return VS_ERR_INVALID;
    }
    nprobe = std::min(nprobe, h->nlist);  // IVFIndex.cpp:647
    if (nprobe > kMaxNprobe) {
        set_error("nprobe > 256 not supported");
        return VS_ERR_UNSUPPORTED;
    }
    int rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = order_begin(h, st))) return rc;
    rc = ivf_multi_dev(h, queries_dev, 1, B, k, nprobe, dists_dev, ids_dev, st);
    return rc ? rc : order_end(h, st);
}

int vs_ivf_search_dev_multi(vs_index* h, const float* queries_dev, int n_batches, int B, int k, int nprobe, int32_t* ids_dev,
                            float* dists_dev, void* stream) {
    if (!h || h->kind != 1 || !queries_dev || !ids_dev || !dists_dev || n_batches < 1 || B < 1 || B > vs::kMaxBatch || k < 1 ||
        nprobe < 1) {
        set_error("vs_ivf_search_dev_multi: bad arguments");
        return VS_ERR_INVALID;
    }
    nprobe = std::min(nprobe, h->nlist);  // IVFIndex.cpp:647
    if (nprobe > kMaxNprobe) {
        set_error("nprobe > 256 not supported");
        return VS_ERR_UNSUPPORTED;
    }
    int rc = set_device(h);
    if (rc) return rc;
    hipStream_t user = static_cast<hipStream_t>(stream);
    if ((rc = order_begin(h, user))) return rc;
    rc = ivf_multi_dev(h, queries_dev, n_batches, B, k, nprobe, dists_dev, ids_dev, user);
    return rc ? rc : order_end(h, user);
}

int vs_ivf_search(vs_index* h, const float* queries_host, int64_t nq, int k, int nprobe, int32_t* ids, float* dists,
                  int64_t* total_candidates, vs_timing* timing) {
    if (!h || h->kind != 1 || !queries_host || !ids || !dists || nq < 0 || k < 1 || nprobe < 1) {
        set_error("vs_ivf_search: bad arguments");
        return VS_ERR_INVALID;
    }
    nprobe = std::min(nprobe, h->nlist);
    if (nprobe > kMaxNprobe) {
        set_error("nprobe > 256 not supported");
        return VS_ERR_UNSUPPORTED;
    }
    if (k > 64) {
        set_error("k > 64 not supported by the host-buffer API");
        return VS_ERR_UNSUPPORTED;
    }
    return guarded([&]() -> int {
        int rc = set_device(h);
        if (rc) return rc;
        const double t_start = now_ms();
        vs_timing tm{};
        if ((rc = ensure_pipe(h)) || (rc = ensure_wide_streams(h)) || (rc = ensure_ivf_host(h)) || (rc = order_begin(h, h->stream))) return rc;
        h->stage_on = true;
        h->stage_used = 0;
        HIPCHK(hipMemsetAsync(h->d_cand, 0, sizeof(unsigned long long), h->stream));
        const float inf = std::numeric_limits<float>::infinity();
        // Queries go through in chunks (the harness loop of main_ivf.cpp:150-214 collapsed into a call): a chunk is ONE
        // upload, its launch groups dealt to the two lanes' streams, ONE download -- the host, not the device, is the limit
        // of this call (a hipMemcpyAsync costs about as much host time as a launch group's five launches).  Two chunks in
        // flight; at least two chunks per call where there is enough work, so that the second upload runs beside the
        // first chunk's kernels.  A ragged tail batch gets a launch group of its own.
        const bool wide = ivf_wide_ok(h, k);
        HIPCHK(hipEventRecord(h->wide_fork, h->stream));  // (behind the memset above)
        for (int i = 0; i < 2; ++i) HIPCHK(hipStreamWaitEvent(h->wide_stream[i], h->wide_fork, 0));
        const int64_t group_q = (int64_t)h->ivf_gb * h->batch;
        const int64_t unit_q = (int64_t)vs::kIvfWideBatches * h->batch;  // a super-batch of queries: chunks are cut at these
        const int64_t cap_q = h->ivf_host_cap / unit_q * unit_q;
        const int64_t wchunk = std::min<int64_t>(cap_q, std::max<int64_t>(unit_q, (nq / 2 + unit_q - 1) / unit_q * unit_q));
        int next_lane = 0;
        auto enqueue = [&](vs_index::IvfHostSlot& S, int64_t q0, int64_t n) -> int {
            const double t0 = now_ms();
            S.q0 = q0;
            S.n = n;
            std::memcpy(S.pin_q, queries_host + q0 * vs::kDim, (size_t)n * vs::kDim * sizeof(float));
            HIPCHK(hipMemcpyAsync(S.d_q, S.pin_q, (size_t)n * vs::kDim * sizeof(float), hipMemcpyHostToDevice, h->s_h2d));
            HIPCHK(hipEventRecord(S.ev_h2d, h->s_h2d));
            tm.h2d_ms += now_ms() - t0;
            float* od = S.d_out;
            int32_t* oi = reinterpret_cast<int32_t*>(S.d_out + (size_t)n * k);
            bool used[2] = {false, false};
            for (int64_t g0 = 0; g0 < n; g0 += group_q) {
                const int64_t gn = std::min<int64_t>(group_q, n - g0);
                const int full = (int)(gn / h->batch), rem = (int)(gn % h->batch);
                const int lane = wide ? next_lane : 0;
                next_lane ^= 1;
                const hipStream_t cs = h->wide_stream[lane];
                if (!used[lane]) HIPCHK(hipStreamWaitEvent(cs, S.ev_h2d, 0));
                used[lane] = true;
                int r2 = VS_OK;
                auto run = [&](size_t o, int nb, int B) -> int {
                    if (wide) return ivf_group_wide_dev(h, lane, S.d_q + o * vs::kDim, nb, B, k, nprobe, od + o * k, oi + o * k, cs);
                    int r3 = VS_OK;
                    for (int b = 0; b < nb && !r3; ++b)
                        r3 = ivf_fallback_batch_dev(h, S.d_q + (o + (size_t)b * B) * vs::kDim, B, k, nprobe, od + (o + (size_t)b * B) * k,
                                                    oi + (o + (size_t)b * B) * k, cs);
                    return r3;
                };
                if (full) r2 = run((size_t)g0, full, h->batch);
                if (!r2 && rem) r2 = run((size_t)g0 + (size_t)full * h->batch, 1, rem);  // the call's ragged tail
                if (r2) return r2;
            }
            for (int lane = 0; lane < 2; ++lane)
                if (used[lane]) {
                    HIPCHK(hipEventRecord(S.ev_comp[lane], h->wide_stream[lane]));
                    HIPCHK(hipStreamWaitEvent(h->s_d2h, S.ev_comp[lane], 0));
                }
            HIPCHK(hipMemcpyAsync(S.pin_out, S.d_out, (size_t)n * k * 2 * sizeof(float), hipMemcpyDeviceToHost, h->s_d2h));
            HIPCHK(hipEventRecord(S.ev_d2h, h->s_d2h));
            return VS_OK;
        };
        auto retire = [&](vs_index::IvfHostSlot& S) -> int {
            if (S.q0 < 0) return VS_OK;
            const double t0 = now_ms();
            HIPCHK(hipEventSynchronize(S.ev_d2h));
            tm.d2h_ms += now_ms() - t0;
            const float* hd = S.pin_out;
            const int32_t* hi = reinterpret_cast<const int32_t*>(S.pin_out + (size_t)S.n * k);
            for (int64_t i = 0; i < S.n * k; ++i) {
                const int32_t id = hi[i];
                ids[S.q0 * k + i] = id;
                dists[S.q0 * k + i] = id >= 0 ? (h->metric == VS_METRIC_IP ? -hd[i] : hd[i]) : inf;  // (IP: the score q.v, as IVFIndex returns it)
            }
            S.q0 = -1;
            return VS_OK;
        };
        for (auto& S : h->ihs) S.q0 = -1;  // (a call that failed half way may have left a chunk marked in flight)
        int c = 0;
        for (int64_t q0 = 0; q0 < nq; q0 += wchunk, ++c) {
            vs_index::IvfHostSlot& S = h->ihs[c & 1];
            if ((rc = retire(S))) return rc;
            if ((rc = enqueue(S, q0, std::min<int64_t>(wchunk, nq - q0)))) return rc;
        }
        if ((rc = retire(h->ihs[c & 1]))) return rc;
        if ((rc = retire(h->ihs[(c + 1) & 1]))) return rc;
        for (int i = 0; i < 2; ++i) {  // the index's stream continues behind the lanes
            HIPCHK(hipEventRecord(h->wide_join[i], h->wide_stream[i]));
            HIPCHK(hipStreamWaitEvent(h->stream, h->wide_join[i], 0));
        }
        unsigned long long cand = 0;
        HIPCHK(hipMemcpyAsync(&cand, h->d_cand, sizeof(cand), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (total_candidates) *total_candidates = (int64_t)cand;
        tm.total_ms = now_ms() - t_start;
        // SearchTiming split (IVFIndex.h:31-36): device time of the three stages from the events between their launches,
        // summed over the call's launch groups (uploads and downloads overlap them)
        for (int i = 0; i + 3 < h->stage_used; i += 4) {
            float ms[3] = {0, 0, 0};
            for (int j = 0; j < 3; ++j) (void)hipEventElapsedTime(&ms[j], h->stage_ev[i + j], h->stage_ev[i + j + 1]);
            tm.centroid_search_ms += ms[0];
            tm.gather_ms += ms[1];
            tm.fine_search_ms += ms[2];
        }
        h->stage_on = false;
        h->stage_used = 0;
        if (timing) *timing = tm;
        return order_end(h, h->stream);
    });
}

#ifdef VS_STAMPS
// diagnostic builds only (make EXTRA=-DVS_STAMPS; not part of the ABI): state of the wide IVF pipeline after the last launch group
__attribute__((visibility("default"))) int vs_debug_ivf_wide_stats(vs_index* h, int64_t* out /*[8]*/) {
    if (!h || !h->wide[0].lq) return VS_ERR_INVALID;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    const size_t nq = (size_t)h->ivf_gb * 32;
    const int n_sb_max = h->ivf_nsb;
    std::vector<int32_t> z(h->wide[0].zero_words);
    std::vector<float> tau(nq);
    HIPCHK(hipMemcpy(z.data(), h->wide[0].zero, z.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tau.data(), h->wide[0].tau, nq * 4, hipMemcpyDeviceToHost));
    const int32_t* slow = z.data() + (size_t)n_sb_max * vs::ivf_wide_plan_words(h->nlist);
    const int32_t* ovf = slow + nq;
    const int32_t* cnt = ovf + 16 + h->ivf_gb;
    int64_t nslow = 0, total = 0, maxw = 0, maxsub = 0, ninf = 0;
    for (size_t i = 0; i < nq; ++i) nslow += slow[i] != 0;
    for (size_t i = 0; i < nq; ++i) ninf += !(tau[i] < 3e38f);
    for (size_t i = 0; i < nq * kWideSub; ++i) maxsub = std::max<int64_t>(maxsub, cnt[i]);
    out[0] = ovf[0];
    out[1] = nslow;
    out[3] = maxw;
    out[4] = maxsub;
    out[5] = ninf;
    out[6] = z[(size_t)h->nlist * vs::kIvfWideCntStride];  // records of super-batch 0
    int64_t big = 0, maxq = 0;
    for (size_t i = 0; i < nq; ++i) {
        int64_t t = 0;
        for (int g = 0; g < kWideSub; ++g) t += cnt[(size_t)g * nq + i];
        big += t > 256;
        maxq = std::max(maxq, t);
        total += t;
    }
    out[2] = total;
    out[7] = big * 100000 + maxq;
    return VS_OK;
}

__attribute__((visibility("default"))) int vs_debug_buffer(int* dev_ptr) {
    g_dbg = dev_ptr;
    return VS_OK;
}
#endif

// --------------------------------------------------------------------------------------- multi-GPU
int vs_topk_merge_dev(const float* dists_dev, const int32_t* ids_dev, int G, int B, int kin, int64_t stride_g,
                      int kout, float* out_dists_dev, int32_t* out_ids_dev, int32_t* flags_dev, void* stream) {
    if (!dists_dev || !ids_dev || !out_dists_dev || !out_ids_dev || G < 1 || B < 1 || kin < 1 || kout < 1) {
        set_error("vs_topk_merge_dev: bad arguments");
        return VS_ERR_INVALID;
    }
    vs::MergeParams m{};
    m.part_d = dists_dev;
    m.part_i = ids_dev;
    m.G = G;
    m.kin = kin;
    m.nq = B;
    m.kout = kout;
    m.out_d = out_dists_dev;
    m.out_i = out_ids_dev;
    m.flags = flags_dev;
    HIPCHK(vs::launch_merge_layout(m, stride_g > 0 ? stride_g : (int64_t)B * kin, kin, static_cast<hipStream_t>(stream)));
    return VS_OK;
}

}  // extern "C"

// --------------------------------------------------------------------------------------- multi-GPU
// One process per GPU.  The only data-path collective of the hot path is the all-gather of per-shard top-k lists
// (SURVEY.md 8e); it lives here, behind the ABI: a vs_comm owns an RCCL communicator, a stream for the collective
// and two sets of exchange buffers, so that all-gather + merge of launch group g run beside the scan of group g + 1.
// RCCL is bound at run time (dlopen of librccl.so.1: inside a PyTorch process that is the copy torch already loaded,
// otherwise ROCm's), so single-GPU users of the library do not need it at all.
namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        std::vector<std::string> names = {"librccl.so.1", "librccl.so"};
        if (const char* rp = getenv("ROCM_PATH")) names.push_back(std::string(rp) + "/lib/librccl.so.1");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const std::string& n : names) {
            a.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
            if (const char* e = dlerror()) a.err = e;
        }
        if (!a.lib) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) {
            a.err = "librccl is missing a required symbol";
            a.lib = nullptr;
        }
        return a;
    }();
    return api;
}

int rccl_ready() {
    if (rccl().lib) return VS_OK;
    set_error("RCCL not available: " + rccl().err);
    return VS_ERR_DEVICE;
}

#define NCCLCHK(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess) {                                                                       \
            set_error(std::string(#expr) + ": " + rccl().GetErrorString(_r));                          \
            return VS_ERR_DEVICE;                                                                      \
        }                                                                                              \
    } while (0)

}  // namespace

struct vs_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t s_coll = nullptr;
    hipEvent_t ev_scan[2] = {}, ev_coll[2] = {};
    bool coll_used[2] = {false, false};
    int32_t* d_loc[2] = {};   // this rank's lists of one launch group: [dists n*kin][ids n*kin] as 32-bit words
    int32_t* d_gath[2] = {};  // [world] x the same
    size_t cap_words = 0;     // per-rank capacity of d_loc
    // cluster-sharded IVF: the slices' blocks (probes | tau | slow) of one launch group, exchanged between its two halves
    int32_t* d_blk[2] = {};   // this rank's block
    int32_t* d_blkg[2] = {};  // [world] blocks
    size_t blk_cap = 0;       // words per block
    hipEvent_t ev_front[2] = {}, ev_probe[2] = {};
};

namespace {

int comm_reserve(vs_comm* c, size_t words) {
    if (c->cap_words >= words) return VS_OK;
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < 2; ++i) {
        if (c->d_loc[i]) (void)hipFree(c->d_loc[i]);
        if (c->d_gath[i]) (void)hipFree(c->d_gath[i]);
        c->d_loc[i] = c->d_gath[i] = nullptr;
    }
    c->cap_words = 0;
    for (int i = 0; i < 2; ++i) {
        int rc;
        if ((rc = dev_alloc(&c->d_loc[i], words))) return rc;
        if ((rc = dev_alloc(&c->d_gath[i], words * (size_t)c->world))) return rc;
    }
    c->cap_words = words;
    return VS_OK;
}

// groups of <= kMaxMulti batches: local top-kin of group g on `user` -> event -> (collective stream) all-gather + merge
// into the caller's outputs; the local search of group g + 1 is enqueued on `user` right away and runs meanwhile.
template <class LocalSearch>
int sharded_groups(vs_index* h, vs_comm* c, int n_batches, int B, int kin, int kout, int32_t* ids_dev, float* dists_dev,
                   int32_t* flags_dev, const int32_t* id_map_unused, hipStream_t user, LocalSearch local) {
    (void)id_map_unused;
    if (h->device != c->device) {
        set_error("index and communicator live on different devices");
        return VS_ERR_INVALID;
    }
    int rc = comm_reserve(c, (size_t)(2 * kin + 1) * kMaxMulti * 32);
    if (rc) return rc;
    int g = 0;
    for (int b0 = 0; b0 < n_batches; b0 += kMaxMulti, ++g) {
        const int nb = std::min(kMaxMulti, n_batches - b0);
        const int buf = g & 1;
        const size_t n = (size_t)nb * B;         // queries of the group
        const size_t words = 2 * n * kin + (flags_dev ? n : 0);  // per rank: dists | ids | (brute force) the shard's own flags
        if (c->coll_used[buf]) HIPCHK(hipStreamWaitEvent(user, c->ev_coll[buf], 0));  // group g - 2 is done with the buffers
        float* loc_d = reinterpret_cast<float*>(c->d_loc[buf]);
        int32_t* loc_i = c->d_loc[buf] + n * kin;
        if ((rc = local(b0, nb, loc_d, loc_i, c->d_loc[buf] + 2 * n * kin, user))) return rc;
        HIPCHK(hipEventRecord(c->ev_scan[buf], user));
        HIPCHK(hipStreamWaitEvent(c->s_coll, c->ev_scan[buf], 0));
        const int32_t* src = c->d_loc[buf];
        if (c->world > 1) {
            NCCLCHK(rccl().AllGather(c->d_loc[buf], c->d_gath[buf], words, ncclInt32, c->comm, c->s_coll));
            src = c->d_gath[buf];
        }
        vs::MergeParams m{};
        m.part_d = reinterpret_cast<const float*>(src);
        m.part_i = src + n * kin;
        m.G = c->world;
        m.kin = kin;
        m.nq = (int)n;
        m.kout = kout;
        m.out_d = dists_dev + (size_t)b0 * B * kout;
        m.out_i = ids_dev + (size_t)b0 * B * kout;
        m.flags = flags_dev ? flags_dev + (size_t)b0 * B : nullptr;
        m.flag_empty = 1;
        if (flags_dev) {  // a shard that skipped a batch (int8 rows, non-byte query) must not go unnoticed: its rows are missing
            m.shard_flags = src + 2 * n * kin;
            m.shard_flags_stride = (int64_t)words;
        }
        HIPCHK(vs::launch_merge_layout(m, (int64_t)words, kin, c->s_coll));
        HIPCHK(hipEventRecord(c->ev_coll[buf], c->s_coll));
        c->coll_used[buf] = true;
    }
    // the caller's stream continues after the last two groups' merges
    for (int buf = 0; buf < 2; ++buf)
        if (c->coll_used[buf]) HIPCHK(hipStreamWaitEvent(user, c->ev_coll[buf], 0));
    return VS_OK;
}

}  // namespace

extern "C" {

int vs_comm_unique_id(void* id_out) {
    if (!id_out) {
        set_error("vs_comm_unique_id: id_out is NULL");
        return VS_ERR_INVALID;
    }
    int rc = rccl_ready();
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == VS_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id_out, &id, sizeof(id));
    return VS_OK;
}

int vs_comm_create(const void* unique_id, int rank, int world, int device, vs_comm** out) {
    if (!out || !unique_id || world < 1 || rank < 0 || rank >= world) {
        set_error("vs_comm_create: bad arguments");
        return VS_ERR_INVALID;
    }
    int rc = check_device(device);
    if (rc) return rc;
    if ((rc = rccl_ready())) return rc;
    HIPCHK(hipSetDevice(device));
    vs_comm* c = new (std::nothrow) vs_comm();
    if (!c) {
        set_error("out of host memory");
        return VS_ERR_NOMEM;
    }
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error(std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
        delete c;
        return VS_ERR_DEVICE;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->s_coll, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->ev_scan[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_coll[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_front[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_probe[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        set_error(std::string("vs_comm_create: ") + hipGetErrorString(e));
        vs_comm_destroy(c);
        return VS_ERR_DEVICE;
    }
    *out = c;
    return VS_OK;
}

int vs_comm_rank(const vs_comm* c) { return c ? c->rank : -1; }
int vs_comm_world(const vs_comm* c) { return c ? c->world : 0; }

void vs_comm_destroy(vs_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->s_coll) (void)hipStreamSynchronize(c->s_coll);
    if (c->comm && rccl().lib) (void)rccl().CommDestroy(c->comm);
    for (int i = 0; i < 2; ++i) {
        if (c->d_loc[i]) (void)hipFree(c->d_loc[i]);
        if (c->d_gath[i]) (void)hipFree(c->d_gath[i]);
        if (c->ev_scan[i]) (void)hipEventDestroy(c->ev_scan[i]);
        if (c->ev_coll[i]) (void)hipEventDestroy(c->ev_coll[i]);
        if (c->d_blk[i]) (void)hipFree(c->d_blk[i]);
        if (c->d_blkg[i]) (void)hipFree(c->d_blkg[i]);
        if (c->ev_front[i]) (void)hipEventDestroy(c->ev_front[i]);
        if (c->ev_probe[i]) (void)hipEventDestroy(c->ev_probe[i]);
    }
    if (c->s_coll) (void)hipStreamDestroy(c->s_coll);
    delete c;
}

int vs_bf_search_dev_sharded(vs_index* h, vs_comm* c, const float* queries_dev, int n_batches, int B, int k, int32_t* ids_dev,
                             float* dists_dev, int32_t* flags_dev, void* stream) {
    if (!h || !c || h->kind != 0 || !queries_dev || !ids_dev || !dists_dev || n_batches < 1 || B < 1 || B > vs::kMaxBatch || k < 1) {
        set_error("vs_bf_search_dev_sharded: bad arguments");
        return VS_ERR_INVALID;
    }
    int rc = set_device(h);
    if (rc) return rc;
    const int k1 = k + 1;
    if (!pick_kcap(k1)) {
        set_error("k too large for the compiled scan kernels (k <= 15)");
        return VS_ERR_UNSUPPORTED;
    }
    hipStream_t user = static_cast<hipStream_t>(stream);
    if ((rc = order_begin(h, user))) return rc;
    if (!flags_dev && n_batches <= kMaxMulti) flags_dev = h->d_flags;  // (the shards' flags travel whenever there is room to merge them)
    rc = sharded_groups(h, c, n_batches, B, k1, k1, ids_dev, dists_dev, flags_dev, nullptr, user,
                        [&](int b0, int nb, float* loc_d, int32_t* loc_i, int32_t* loc_f, hipStream_t s) {
                            return bf_launch(h, h->lane[0], queries_dev + (size_t)b0 * B * vs::kDim, nb, B, k1, loc_d, loc_i, loc_f, s);
                        });
    return rc ? rc : order_end(h, user);
}

int vs_ivf_search_dev_sharded(vs_index* h, vs_comm* c, const float* queries_dev, int n_batches, int B, int k, int nprobe,
                              int32_t* ids_dev, float* dists_dev, void* stream) {
    if (!h || !c || h->kind != 1 || !queries_dev || !ids_dev || !dists_dev || n_batches < 1 || B < 1 || B > vs::kMaxBatch || k < 1 ||
        nprobe < 1) {
        set_error("vs_ivf_search_dev_sharded: bad arguments");
        return VS_ERR_INVALID;
    }
    nprobe = std::min(nprobe, h->nlist);  // IVFIndex.cpp:647
    if (nprobe > kMaxNprobe || !pick_kcap(k)) {
        set_error("nprobe > 256 or k > 16 not supported");
        return VS_ERR_UNSUPPORTED;
    }
    if (h->world != c->world || h->rank != c->rank) {
        set_error("the index was sharded for a different (rank, world) than the communicator's");
        return VS_ERR_INVALID;
    }
    if (h->device != c->device) {
        set_error("index and communicator live on different devices");
        return VS_ERR_INVALID;
    }
    int rc = set_device(h);
    if (rc) return rc;
    hipStream_t user = static_cast<hipStream_t>(stream);
    if ((rc = order_begin(h, user))) return rc;
    // The per-query stages can be cut by slice only where EVERY rank runs the wide pipeline (a rank without resident
    // rows, or nlist > 4096, takes the query-major fallback): that is a property of the index, the same on all ranks,
    // except for empty shards -- which only occur with fewer lists than ranks.
    const bool sliced = c->world > 1 && c->world <= kIvfShardMaxWorld && h->nlist <= vs::kIvfFastNlist && h->nlist >= c->world;
    if (!sliced) {
        // every rank runs the whole pipeline on its own lists, one all-gather of top-k lists per launch group
        rc = sharded_groups(h, c, n_batches, B, k, k, ids_dev, dists_dev, nullptr, nullptr, user,
                            [&](int b0, int nb, float* loc_d, int32_t* loc_i, int32_t*, hipStream_t s) -> int {
                                return ivf_multi_dev(h, queries_dev + (size_t)b0 * B * vs::kDim, nb, B, k, nprobe, loc_d, loc_i, s);
                            });
        return rc ? rc : order_end(h, user);
    }
    // ---- sliced pipeline: group g's front half (own slice), the exchange of the slices' blocks, its back half (all
    // slices, own lists), the all-gather of top-k lists and their merge.  Software pipelined over the launch groups:
    // the compute stream runs F(g), then B(g - 1); the collective stream P(g), then T(g - 1) -- the same order on every
    // rank -- so that an exchange is in flight while the neighbouring group computes.  Group g uses lane g & 1.
    const int world = c->world, gb = h->ivf_gb;
    const size_t blk_words_max = (size_t)ivf_block_words(vs::kIvfWideBatches, kMaxNprobe);
    if (c->blk_cap < blk_words_max) {
        HIPCHK(hipDeviceSynchronize());
        for (int i = 0; i < 2; ++i) {
            if (c->d_blk[i]) (void)hipFree(c->d_blk[i]);
            if (c->d_blkg[i]) (void)hipFree(c->d_blkg[i]);
            c->d_blk[i] = c->d_blkg[i] = nullptr;
        }
        c->blk_cap = 0;
        for (int i = 0; i < 2; ++i) {
            if ((rc = dev_alloc(&c->d_blk[i], blk_words_max))) return rc;
            if ((rc = dev_alloc(&c->d_blkg[i], blk_words_max * (size_t)world))) return rc;
        }
        c->blk_cap = blk_words_max;
    }
    if ((rc = comm_reserve(c, (size_t)2 * gb * 32 * k))) return rc;
    struct Grp {
        int b0, nb, sbb;
    };
    auto group_of = [&](int g) {
        Grp G;
        G.b0 = g * gb;
        G.nb = std::min(gb, n_batches - G.b0);
        int b0_, nbs_;
        ivf_slice(G.nb, world, 0, G.sbb, b0_, nbs_);  // batches per slice (<= 32)
        return G;
    };
    const int n_groups = (n_batches + gb - 1) / gb;
    auto back_and_gather = [&](int g) -> int {
        const Grp G = group_of(g);
        const int lane = g & 1;
        const size_t n = (size_t)G.nb * B, words = 2 * n * k;
        float* loc_d = reinterpret_cast<float*>(c->d_loc[lane]);
        int32_t* loc_i = c->d_loc[lane] + n * k;
        HIPCHK(hipStreamWaitEvent(user, c->ev_probe[lane], 0));
        if (c->coll_used[lane]) HIPCHK(hipStreamWaitEvent(user, c->ev_coll[lane], 0));  // group g - 2's merge has read d_gath
        int r2 = ivf_shard_back(h, lane, queries_dev + (size_t)G.b0 * B * vs::kDim, G.nb, G.sbb, B, k, nprobe, c->d_blkg[lane], loc_d, loc_i, user);
        if (r2) return r2;
        HIPCHK(hipEventRecord(c->ev_scan[lane], user));
        HIPCHK(hipStreamWaitEvent(c->s_coll, c->ev_scan[lane], 0));
        NCCLCHK(rccl().AllGather(c->d_loc[lane], c->d_gath[lane], words, ncclInt32, c->comm, c->s_coll));
        vs::MergeParams m{};
        m.part_d = reinterpret_cast<const float*>(c->d_gath[lane]);
        m.part_i = c->d_gath[lane] + n * k;
        m.G = world;
        m.kin = k;
        m.nq = (int)n;
        m.kout = k;
        m.out_d = dists_dev + (size_t)G.b0 * B * k;
        m.out_i = ids_dev + (size_t)G.b0 * B * k;
        HIPCHK(vs::launch_merge_layout(m, (int64_t)words, k, c->s_coll));
        HIPCHK(hipEventRecord(c->ev_coll[lane], c->s_coll));
        c->coll_used[lane] = true;
        return VS_OK;
    };
    for (int g = 0; g < n_groups; ++g) {
        const Grp G = group_of(g);
        const int lane = g & 1;
        int sbb_, sb0, nbs;
        ivf_slice(G.nb, world, c->rank, sbb_, sb0, nbs);
        // (lane's scratch and block buffers: group g - 2's back half is behind on this stream, its exchange was waited for there)
        if ((rc = ivf_shard_front(h, lane, queries_dev + (size_t)G.b0 * B * vs::kDim, G.nb, G.sbb, sb0, nbs, B, k, nprobe, c->d_blk[lane], user)))
            return rc;
        HIPCHK(hipEventRecord(c->ev_front[lane], user));
        HIPCHK(hipStreamWaitEvent(c->s_coll, c->ev_front[lane], 0));
        NCCLCHK(rccl().AllGather(c->d_blk[lane], c->d_blkg[lane], (size_t)ivf_block_words(G.sbb, nprobe), ncclInt32, c->comm, c->s_coll));
        HIPCHK(hipEventRecord(c->ev_probe[lane], c->s_coll));
        if (g >= 1 && (rc = back_and_gather(g - 1))) return rc;
    }
    if ((rc = back_and_gather(n_groups - 1))) return rc;
    for (int lane = 0; lane < 2; ++lane)
        if (c->coll_used[lane]) HIPCHK(hipStreamWaitEvent(user, c->ev_coll[lane], 0));
    return order_end(h, user);
}

// Virtual ranks: the cluster-sharded pipeline of vs_ivf_search_dev_sharded for G shards that live on ONE device, driven by
// one thread, the two collectives replaced by writing every rank's block / top-k lists straight into the gathered
// layout.  For tests (the sliced pipeline must reproduce the unsharded result) and for measuring what a rank of a G-way
// job does per launch group on a single GPU: rank_ms[r] (optional) = device time of rank r's front + back halves.
int vs_ivf_search_dev_vshards(vs_index* const* shards, int G, const float* queries_dev, int n_batches, int B, int k, int nprobe,
                              int32_t* ids_dev, float* dists_dev, double* rank_ms, void* stream) {
    if (!shards || G < 2 || G > kIvfShardMaxWorld || !queries_dev || !ids_dev || !dists_dev || n_batches < 1 || B < 1 || B > vs::kMaxBatch || k < 1 ||
        nprobe < 1) {
        set_error("vs_ivf_search_dev_vshards: bad arguments");
        return VS_ERR_INVALID;
    }
    for (int r = 0; r < G; ++r)
        if (!shards[r] || shards[r]->kind != 1 || shards[r]->world != G || shards[r]->rank != r || shards[r]->device != shards[0]->device ||
            shards[r]->nlist != shards[0]->nlist || !ivf_wide_ok(shards[r], k)) {
            set_error("vs_ivf_search_dev_vshards: shard r must be an IVF index created with (rank r, world G) on one device, rows resident, nlist <= 4096");
            return VS_ERR_INVALID;
        }
    vs_index* h0 = shards[0];
    nprobe = std::min(nprobe, h0->nlist);
    if (nprobe > kMaxNprobe || !pick_kcap(k)) {
        set_error("nprobe > 256 or k > 16 not supported");
        return VS_ERR_UNSUPPORTED;
    }
    return guarded([&]() -> int {
        int rc = set_device(h0);
        if (rc) return rc;
        hipStream_t s = static_cast<hipStream_t>(stream);
        for (int r = 0; r < G; ++r)
            if ((rc = order_begin(shards[r], s))) return rc;
        const int gb = h0->ivf_gb;
        const size_t blk_max = (size_t)ivf_block_words(vs::kIvfWideBatches, kMaxNprobe);
        const size_t loc_max = (size_t)2 * gb * 32 * k;
        if (!h0->vsh_blk || h0->vsh_loc_words < loc_max * G) {
            HIPCHK(hipStreamSynchronize(s));
            if (h0->vsh_blk) (void)hipFree(h0->vsh_blk);
            if (h0->vsh_loc) (void)hipFree(h0->vsh_loc);
            h0->vsh_blk = h0->vsh_loc = nullptr;
            if ((rc = dev_alloc(&h0->vsh_blk, blk_max * G)) || (rc = dev_alloc(&h0->vsh_loc, loc_max * G))) return rc;
            h0->vsh_loc_words = loc_max * G;
        }
        std::vector<hipEvent_t> ev;
        if (rank_ms) {
            ev.resize((size_t)4 * G);
            for (auto& e : ev) HIPCHK(hipEventCreate(&e));
            for (int r = 0; r < G; ++r) rank_ms[r] = 0;
        }
        for (int b0 = 0; b0 < n_batches && !rc; b0 += gb) {
            const int nb = std::min(gb, n_batches - b0), sbb = (nb + G - 1) / G;  // (= ivf_slice's)
            const float* q = queries_dev + (size_t)b0 * B * vs::kDim;
            const size_t n = (size_t)nb * B, words = 2 * n * k;
            const long long bw = ivf_block_words(sbb, nprobe);
            for (int r = 0; r < G && !rc; ++r) {
                int sbb_, sb0, nbs;
                ivf_slice(nb, G, r, sbb_, sb0, nbs);
                if (rank_ms) HIPCHK(hipEventRecord(ev[4 * r], s));
                rc = ivf_shard_front(shards[r], 0, q, nb, sbb, sb0, nbs, B, k, nprobe, h0->vsh_blk + (size_t)r * bw, s);
                if (rank_ms) HIPCHK(hipEventRecord(ev[4 * r + 1], s));
            }
            for (int r = 0; r < G && !rc; ++r) {
                float* loc_d = reinterpret_cast<float*>(h0->vsh_loc + (size_t)r * words);
                int32_t* loc_i = h0->vsh_loc + (size_t)r * words + n * k;
                if (rank_ms) HIPCHK(hipEventRecord(ev[4 * r + 2], s));
                rc = ivf_shard_back(shards[r], 0, q, nb, sbb, B, k, nprobe, h0->vsh_blk, loc_d, loc_i, s);
                if (rank_ms) HIPCHK(hipEventRecord(ev[4 * r + 3], s));
            }
            if (rc) break;
            vs::MergeParams m{};
            m.part_d = reinterpret_cast<const float*>(h0->vsh_loc);
            m.part_i = h0->vsh_loc + n * k;
            m.G = G;
            m.kin = k;
            m.nq = (int)n;
            m.kout = k;
            m.out_d = dists_dev + (size_t)b0 * B * k;
            m.out_i = ids_dev + (size_t)b0 * B * k;
            HIPCHK(vs::launch_merge_layout(m, (int64_t)words, k, s));
            if (rank_ms) {
                HIPCHK(hipStreamSynchronize(s));
                for (int r = 0; r < G; ++r) {
                    float a = 0, b = 0;
                    HIPCHK(hipEventElapsedTime(&a, ev[4 * r], ev[4 * r + 1]));
                    HIPCHK(hipEventElapsedTime(&b, ev[4 * r + 2], ev[4 * r + 3]));
                    rank_ms[r] += a + b;
                }
            }
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
        return rc;
    });
}

// Host-buffer forms of the sharded searches (what the CLIs call with --gpus N): every rank passes the same queries and
// receives the same merged result.  Chunks of kMaxMulti batches: upload, vs_*_search_dev_sharded, download.
}  // extern "C"

namespace {

// ---- brute force over row shards, host buffers in, the reference's answer out (cpu_baseline.cpp:127-153 tie order included).
// The shards this process drives are either ONE shard of a collective job (comm: the other ranks run the same code, the
// exchanges are RCCL all-gathers) or ALL G shards on one device (virtual ranks: an exchange is a no-op, every shard has
// written its part of the gathered buffer in place).  Everything else -- per-shard device steps, merge, host replay -- is
// the same code.  Shards are contiguous row ranges in rank order (vs_bf_create(rows of the shard, id_offset = first row)).
struct BfShards {
    std::vector<vs_index*> hs;  // the shards driven here; hs[i] is global shard first + i
    int G = 1, first = 0;
    vs_comm* c = nullptr;
    vs_index* owner() const { return hs[0]; }
    hipStream_t s() const { return hs[0]->stream; }
    int exchange(int32_t* buf, size_t words) const {  // buf = [G][words], this process's parts in place
        if (!c || c->world == 1) return VS_OK;
        NCCLCHK(rccl().AllGather(buf + (size_t)c->rank * words, buf, words, ncclInt32, c->comm, s()));
        return VS_OK;
    }
};

constexpr size_t kShPackWords = 32 + (size_t)2 * 32 * kTieCap;  // filter output of one shard: cnt [32] | rows [32][kTieCap] | dists [32][kTieCap]

// gathered scratch of the owner: main [G][main] | tau [G][32] | dense [G][32 * kTieDense] | pack [G][kShPackWords] | meta [G][2]
struct ShBuf {
    int32_t *main, *tau, *dense, *pack, *meta;
    size_t main_words;
};
int sh_reserve(vs_index* o, int G, int k1, ShBuf& B) {
    const size_t main_words = (size_t)(2 * k1 + 1) * kMaxMulti * 32;
    const size_t total = (size_t)G * (main_words + 32 + (size_t)32 * kTieDense + kShPackWords + 2);
    if (o->sh_words < total) {
        HIPCHK(hipDeviceSynchronize());
        if (o->d_sh) (void)hipFree(o->d_sh);
        o->d_sh = nullptr;
        o->sh_words = 0;
        int rc = dev_alloc(&o->d_sh, total);
        if (rc) return rc;
        o->sh_words = total;
    }
    B.main_words = main_words;
    B.main = o->d_sh;
    B.tau = B.main + (size_t)G * main_words;
    B.dense = B.tau + (size_t)G * 32;
    B.pack = B.dense + (size_t)G * 32 * kTieDense;
    B.meta = B.pack + (size_t)G * kShPackWords;
    return VS_OK;
}

// exact select_topk for the flagged queries of a sharded search (see resolve_ties for the single-shard form and why a
// row-ordered superset of the rows that change the slot buffer suffices): the dense prefix is shard 0's first rows, its
// k-th smallest distance bounds the buffer maximum for every later row of every shard, each shard filters its rows under
// that bound, and the host replays "dense rows, then the shards' candidates in row order".
int resolve_ties_shards(const BfShards& S, const ShBuf& Bf, const std::vector<int32_t>& meta /*[G][2] rows, id_offset*/,
                        const float* queries_host, const std::vector<int64_t>& flagged, int k, int32_t* ids, float* dists) {
    vs_index* o = S.owner();
    hipStream_t st = S.s();
    const int G = S.G;
    const int64_t L0 = std::min<int64_t>(meta[0], kTieDense);
    const int64_t L0p = (L0 + 15) & ~int64_t(15);
    for (int g = 1; g < G; ++g)
        if ((int64_t)meta[2 * g + 1] != (int64_t)meta[2 * (g - 1) + 1] + meta[2 * (g - 1)]) {
            set_error("sharded tie order needs shards that are contiguous row ranges in rank order");
            return VS_ERR_UNSUPPORTED;
        }
    std::vector<float> qbuf((size_t)32 * vs::kDim), dense((size_t)32 * L0p);
    std::vector<int32_t> cnt((size_t)G * 32), rows;
    std::vector<float> cds;
    int rc;
    for (size_t f0 = 0; f0 < flagged.size(); f0 += 32) {
        const int B = (int)std::min<size_t>(32, flagged.size() - f0);
        for (int b = 0; b < B; ++b)
            std::memcpy(&qbuf[(size_t)b * vs::kDim], queries_host + flagged[f0 + b] * vs::kDim, vs::kDim * sizeof(float));
        HIPCHK(hipMemcpyAsync(o->d_q, qbuf.data(), (size_t)B * vs::kDim * sizeof(float), hipMemcpyHostToDevice, st));
        float* tau0 = reinterpret_cast<float*>(Bf.tau);        // shard 0's part: the bound every shard filters with
        float* dense0 = reinterpret_cast<float*>(Bf.dense);    // shard 0's part: [B][L0p]
        if (S.first == 0) {
            vs_index* h0 = S.hs[0];
            if ((rc = scores_dev(h0, h0->d_vecs, h0->d_norm, L0, o->d_q, B, dense0, L0p, st))) return rc;
            HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tau0), 0xff800000u, 32, st));  // -inf: padding queries emit nothing
            vs::MergeParams m{};
            m.part_d = dense0;
            m.G = (int)L0;
            m.kin = 1;
            m.nq = B;
            m.kout = k;
            m.tau_out = tau0;
            HIPCHK(vs::launch_merge_layout(m, 1, L0p, st));
        }
        if ((rc = S.exchange(Bf.tau, 32)) || (rc = S.exchange(Bf.dense, (size_t)32 * kTieDense))) return rc;
        for (size_t i = 0; i < S.hs.size(); ++i) {
            vs_index* h = S.hs[i];
            const int g = S.first + (int)i;
            int32_t* pk = Bf.pack + (size_t)g * kShPackWords;
            HIPCHK(hipMemsetAsync(pk, 0, 32 * sizeof(int32_t), st));
            const int64_t rb = g == 0 ? L0 : 0;
            if (h->n_rows <= rb) continue;
            vs::ScanParams p{};
            p.base = h->d_vecs;
            p.bnorm = h->d_norm;
            p.q = o->d_q;
            p.n_batches = 1;
            p.metric = h->metric;
            p.nq_valid = B;
            p.k1 = k + 1;
            p.tau0 = tau0;
            p.row_begin = rb;  // a multiple of 16 (kTieDense) where rows follow
            p.row_end = h->n_rows;
            p.f_cnt = pk;
            p.f_row = pk + 32;
            p.f_d = reinterpret_cast<float*>(pk + 32 + (size_t)32 * kTieCap);
            p.f_cap = kTieCap;
            int grid, tp;
            scan_geometry(h->n_rows - rb, h->num_cus, grid, tp);
            p.tiles_per_wg = tp;
            HIPCHK(vs::launch_scan(p, grid, 8, 2, vs::kModeFilter, st));
        }
        if ((rc = S.exchange(Bf.pack, kShPackWords))) return rc;
        HIPCHK(hipMemcpyAsync(dense.data(), dense0, (size_t)B * L0p * sizeof(float), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpy2DAsync(cnt.data(), 32 * 4, Bf.pack, kShPackWords * 4, 32 * 4, G, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int mx = 0;
        std::vector<int> overflow;
        for (int b = 0; b < B; ++b) {
            bool ov = false;
            for (int g = 0; g < G; ++g) {
                const int cg = cnt[(size_t)g * 32 + b];
                if (cg > kTieCap) ov = true;
                else mx = std::max(mx, cg);
            }
            if (ov) overflow.push_back(b);
        }
        rows.assign((size_t)G * 32 * std::max(mx, 1), 0);
        cds.assign((size_t)G * 32 * std::max(mx, 1), 0.f);
        if (mx > 0) {
            for (int g = 0; g < G; ++g) {
                const int32_t* pk = Bf.pack + (size_t)g * kShPackWords;
                HIPCHK(hipMemcpy2DAsync(&rows[(size_t)g * 32 * mx], (size_t)mx * 4, pk + 32, (size_t)kTieCap * 4, (size_t)mx * 4, B, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpy2DAsync(&cds[(size_t)g * 32 * mx], (size_t)mx * 4, pk + 32 + (size_t)32 * kTieCap, (size_t)kTieCap * 4, (size_t)mx * 4, B,
                                        hipMemcpyDeviceToHost, st));
            }
            HIPCHK(hipStreamSynchronize(st));
        }
        std::vector<int32_t> srow, order;
        std::vector<float> sdist;
        for (int b = 0; b < B; ++b) {
            if (std::find(overflow.begin(), overflow.end(), b) != overflow.end()) continue;
            srow.clear();
            sdist.clear();
            for (int64_t j = 0; j < L0; ++j) {
                srow.push_back((int32_t)(j + meta[1]));
                sdist.push_back(dense[(size_t)b * L0p + j]);
            }
            for (int g = 0; g < G; ++g) {
                const int m = cnt[(size_t)g * 32 + b];
                const int32_t* rr = &rows[((size_t)g * 32 + b) * mx];
                const float* dd = &cds[((size_t)g * 32 + b) * mx];
                order.resize((size_t)m);
                std::iota(order.begin(), order.end(), 0);
                std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return rr[x] < rr[y]; });
                for (int j = 0; j < m; ++j) {
                    srow.push_back(rr[order[(size_t)j]] + meta[2 * g + 1]);
                    sdist.push_back(dd[order[(size_t)j]]);
                }
            }
            const int64_t qi = flagged[f0 + b];
            vs::select_topk_slots_sparse(srow.data(), sdist.data(), (int64_t)srow.size(), k, ids + qi * k, dists + qi * k);
        }
        // masses of rows under the bound (duplicates): the query's full distance row, shard after shard, replayed densely
        for (int b : overflow) {
            int64_t ldm = 0, total = 0;
            for (int g = 0; g < G; ++g) {
                ldm = std::max<int64_t>(ldm, ((int64_t)meta[2 * g] + 15) & ~int64_t(15));
                total += meta[2 * g];
            }
            if (o->sh_row_words < (size_t)G * ldm) {
                HIPCHK(hipStreamSynchronize(st));
                if (o->d_sh_row) (void)hipFree(o->d_sh_row);
                o->d_sh_row = nullptr;
                o->sh_row_words = 0;
                if ((rc = dev_alloc(&o->d_sh_row, (size_t)G * ldm))) return rc;
                o->sh_row_words = (size_t)G * ldm;
            }
            for (size_t i = 0; i < S.hs.size(); ++i) {
                vs_index* h = S.hs[i];
                const int g = S.first + (int)i;
                if ((rc = scores_dev(h, h->d_vecs, h->d_norm, h->n_rows, o->d_q + (size_t)b * vs::kDim, 1, o->d_sh_row + (size_t)g * ldm, ldm, st))) return rc;
            }
            if ((rc = S.exchange(reinterpret_cast<int32_t*>(o->d_sh_row), (size_t)ldm))) return rc;
            std::vector<float> row((size_t)total);
            int64_t at = 0;
            for (int g = 0; g < G; ++g) {
                HIPCHK(hipMemcpyAsync(row.data() + at, o->d_sh_row + (size_t)g * ldm, (size_t)meta[2 * g] * sizeof(float), hipMemcpyDeviceToHost, st));
                at += meta[2 * g];
            }
            HIPCHK(hipStreamSynchronize(st));
            const int64_t qi = flagged[f0 + b];
            vs::select_topk_slots_dense(row.data(), total, k, meta[1], ids + qi * k, dists + qi * k);
        }
    }
    return VS_OK;
}

int bf_search_shards(const BfShards& S, const float* queries_host, int64_t nq, int k, int32_t* ids, float* dists, vs_timing* timing) {
    vs_index* o = S.owner();
    hipStream_t st = S.s();
    const int G = S.G;
    const double t_start = now_ms();
    vs_timing tm{};
    const int k1 = k + 1;
    if (!pick_kcap(k1)) {
        set_error("k too large for the compiled scan kernels (k <= 15)");
        return VS_ERR_UNSUPPORTED;
    }
    int rc;
    ShBuf Bf{};
    if ((rc = sh_reserve(o, G, k1, Bf))) return rc;
    // every process needs every shard's (rows, id_offset)
    std::vector<int32_t> meta((size_t)2 * G, 0);
    for (size_t i = 0; i < S.hs.size(); ++i) {
        const int32_t mine[2] = {(int32_t)S.hs[i]->n_rows, (int32_t)S.hs[i]->id_offset};
        HIPCHK(hipMemcpyAsync(Bf.meta + 2 * (S.first + i), mine, sizeof(mine), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));  // (`mine` is a stack buffer)
    }
    if ((rc = S.exchange(Bf.meta, 2))) return rc;
    HIPCHK(hipMemcpyAsync(meta.data(), Bf.meta, meta.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int batch = o->batch;
    const int64_t chunk = (int64_t)kMaxMulti * batch;
    std::vector<float> hd((size_t)chunk * k1);
    std::vector<int32_t> hi((size_t)chunk * k1), hf((size_t)chunk);
    std::vector<int64_t> flagged;
    const float inf = std::numeric_limits<float>::infinity();
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t n = std::min<int64_t>(chunk, nq - q0);
        const int full = (int)(n / batch), rem = (int)(n % batch);
        HIPCHK(hipMemcpyAsync(o->d_q, queries_host + q0 * vs::kDim, (size_t)n * vs::kDim * sizeof(float), hipMemcpyHostToDevice, st));
        // one part of the chunk (its full batches, or the ragged tail batch): local lists of every shard driven here, the
        // exchange, the merge into the owner's output buffers
        auto part = [&](size_t o0, int nb, int B, bool force_f32) -> int {
            const size_t np = (size_t)nb * B, words = 2 * np * k1 + np;
            for (size_t i = 0; i < S.hs.size(); ++i) {
                int32_t* loc = Bf.main + (size_t)(S.first + i) * words;
                int r2 = bf_launch(S.hs[i], S.hs[i]->lane[0], o->d_q + o0 * vs::kDim, nb, B, k1, reinterpret_cast<float*>(loc),
                                   loc + np * k1, loc + 2 * np * k1, st, force_f32);
                if (r2) return r2;
            }
            int r2 = S.exchange(Bf.main, words);
            if (r2) return r2;
            vs::MergeParams m{};
            m.part_d = reinterpret_cast<const float*>(Bf.main);
            m.part_i = Bf.main + np * k1;
            m.G = G;
            m.kin = k1;
            m.nq = (int)np;
            m.kout = k1;
            m.out_d = o->d_out_d + o0 * k1;
            m.out_i = o->d_out_i + o0 * k1;
            m.flags = o->d_flags + o0;
            m.flag_empty = 1;
            m.shard_flags = Bf.main + 2 * np * k1;
            m.shard_flags_stride = (int64_t)words;
            HIPCHK(vs::launch_merge_layout(m, (int64_t)words, k1, st));
            return VS_OK;
        };
        auto pass = [&](bool force_f32) -> int {
            int r2 = VS_OK;
            if (full) r2 = part(0, full, batch, force_f32);
            if (!r2 && rem) r2 = part((size_t)full * batch, 1, rem, force_f32);
            if (r2) return r2;
            HIPCHK(hipMemcpyAsync(hd.data(), o->d_out_d, (size_t)n * k1 * sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(hi.data(), o->d_out_i, (size_t)n * k1 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(hf.data(), o->d_flags, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            return VS_OK;
        };
        if ((rc = pass(false))) return rc;
        bool rerun = false;
        for (int64_t b = 0; b < n; ++b) rerun = rerun || hf[(size_t)b] == 2;
        // a shard's int8 scan skipped a batch (a query that is not an integer in [0, 255]): the merged flag says so on every
        // rank alike, and every rank reruns the chunk on its fp32 rows
        if (rerun && (rc = pass(true))) return rc;
        for (int64_t b = 0; b < n; ++b) {
            for (int t = 0; t < k; ++t) {
                const int32_t id = hi[(size_t)b * k1 + t];
                ids[(q0 + b) * k + t] = id;
                float d = id >= 0 ? hd[(size_t)b * k1 + t] : inf;
                if (o->metric == VS_METRIC_IP && id >= 0) d = -d;
                dists[(q0 + b) * k + t] = d;
            }
            if (hf[(size_t)b] == 1) flagged.push_back(q0 + b);
        }
    }
    tm.fine_search_ms = now_ms() - t_start;
    if (!flagged.empty() && o->metric == VS_METRIC_L2) {
        const double t0 = now_ms();
        if ((rc = resolve_ties_shards(S, Bf, meta, queries_host, flagged, k, ids, dists))) return rc;
        tm.tie_resolve_ms = now_ms() - t0;
        tm.tie_queries = (int64_t)flagged.size();
    }
    tm.total_ms = now_ms() - t_start;
    if (timing) *timing = tm;
    return VS_OK;
}

}  // namespace

extern "C" {

// Host-buffer forms of the sharded searches (what the CLIs call with --gpus N): every rank passes the same queries and
// receives the same result -- for brute force the reference's own (select_topk's tie order, fp32 rerun of batches the
// int8 scan cannot take), like vs_bf_search on one GPU.
int vs_bf_search_sharded(vs_index* h, vs_comm* c, const float* queries_host, int64_t nq, int k, int32_t* ids, float* dists,
                         vs_timing* timing) {
    if (!h || !c || h->kind != 0 || !queries_host || !ids || !dists || nq < 0 || k < 1) {
        set_error("vs_bf_search_sharded: bad arguments");
        return VS_ERR_INVALID;
    }
    if (h->device != c->device) {
        set_error("index and communicator live on different devices");
        return VS_ERR_INVALID;
    }
    return guarded([&]() -> int {
        int rc = set_device(h);
        if (rc) return rc;
        if ((rc = order_begin(h, h->stream))) return rc;
        BfShards S;
        S.hs = {h};
        S.G = c->world;
        S.first = c->rank;
        S.c = c;
        return bf_search_shards(S, queries_host, nq, k, ids, dists, timing);
    });
}

// Virtual ranks: the same call for G row shards that live on ONE device (shards[g] = vs_bf_create(rows of shard g,
// id_offset = its first row)), driven by the calling thread; the exchanges are no-ops.  For tests of the sharded tie
// order / fp32 rerun without a multi-GPU node.
int vs_bf_search_vshards(vs_index* const* shards, int G, const float* queries_host, int64_t nq, int k, int32_t* ids, float* dists,
                         vs_timing* timing) {
    if (!shards || G < 1 || G > 64 || !queries_host || !ids || !dists || nq < 0 || k < 1) {
        set_error("vs_bf_search_vshards: bad arguments");
        return VS_ERR_INVALID;
    }
    for (int g = 0; g < G; ++g)
        if (!shards[g] || shards[g]->kind != 0 || shards[g]->device != shards[0]->device || shards[g]->metric != shards[0]->metric ||
            shards[g]->batch != shards[0]->batch) {
            set_error("vs_bf_search_vshards: shards must be brute-force indexes on one device with one metric and batch size");
            return VS_ERR_INVALID;
        }
    return guarded([&]() -> int {
        int rc = set_device(shards[0]);
        if (rc) return rc;
        BfShards S;
        S.hs.assign(shards, shards + G);
        S.G = G;
        for (int g = 0; g < G; ++g)
            if ((rc = order_begin(shards[g], shards[0]->stream))) return rc;
        return bf_search_shards(S, queries_host, nq, k, ids, dists, timing);
    });
}

int vs_ivf_search_sharded(vs_index* h, vs_comm* c, const float* queries_host, int64_t nq, int k, int nprobe, int32_t* ids,
                          float* dists, int64_t* total_candidates, vs_timing* timing) {
    if (!h || !c || h->kind != 1 || !queries_host || !ids || !dists || nq < 0 || k < 1 || k > 16 || nprobe < 1) {
        set_error("vs_ivf_search_sharded: bad arguments");
        return VS_ERR_INVALID;
    }
    return guarded([&]() -> int {
        int rc = set_device(h);
        if (rc) return rc;
        const double t_start = now_ms();
        vs_timing tm{};
        HIPCHK(hipMemsetAsync(h->d_cand, 0, sizeof(unsigned long long), h->stream));
        const int64_t chunk = (int64_t)kMaxMulti * h->batch;
        std::vector<float> hd((size_t)chunk * k);
        std::vector<int32_t> hi((size_t)chunk * k);
        const float inf = std::numeric_limits<float>::infinity();
        for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
            const int64_t n = std::min<int64_t>(chunk, nq - q0);
            const int full = (int)(n / h->batch), rem = (int)(n % h->batch);
            HIPCHK(hipMemcpyAsync(h->d_q, queries_host + q0 * vs::kDim, (size_t)n * vs::kDim * sizeof(float), hipMemcpyHostToDevice, h->stream));
            if (full && (rc = vs_ivf_search_dev_sharded(h, c, h->d_q, full, h->batch, k, nprobe, h->d_out_i, h->d_out_d, h->stream))) return rc;
            if (rem) {
                const size_t o = (size_t)full * h->batch;
                if ((rc = vs_ivf_search_dev_sharded(h, c, h->d_q + o * vs::kDim, 1, rem, k, nprobe, h->d_out_i + o * k, h->d_out_d + o * k,
                                                    h->stream)))
                    return rc;
            }
            HIPCHK(hipMemcpyAsync(hd.data(), h->d_out_d, (size_t)n * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(hi.data(), h->d_out_i, (size_t)n * k * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            for (int64_t b = 0; b < n; ++b)
                for (int t = 0; t < k; ++t) {
                    const int32_t id = hi[(size_t)b * k + t];
                    ids[(q0 + b) * k + t] = id;
                    dists[(q0 + b) * k + t] = id >= 0 ? hd[(size_t)b * k + t] : inf;
                }
        }
        unsigned long long cand = 0;  // rows THIS rank scanned (IVFIndex::searchBatch's return value, per shard)
        HIPCHK(hipMemcpy(&cand, h->d_cand, sizeof(cand), hipMemcpyDeviceToHost));
        if (total_candidates) *total_candidates = (int64_t)cand;
        tm.total_ms = tm.fine_search_ms = now_ms() - t_start;
        if (timing) *timing = tm;
        return VS_OK;
    });
}

}  // extern "C"
