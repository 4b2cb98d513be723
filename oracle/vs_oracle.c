/*
 * vs_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's exact-search hot path
 * (/root/reference/cpu/cpu_baseline.cpp) and of the IVF search operator
 * (/root/reference/qidk_ivf/android/app/main/jni/IVFIndex.cpp, reordered mode,
 * with the north-star's L2 metric instead of inner product).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file, and only as the checker / reported baseline.  The
 * product (libvsearch_hip.so) never links or loads it.
 *
 * PARITY PIN STATUS
 *   exact path : "parity unpinned" by the rules: the reference holds no golden
 *                vectors, and it is NOT rebuilt by this repo (it needs <cblas.h>,
 *                which this image lacks; stand-in headers are not allowed, see
 *                DESIGN.md "Oracle").  What agrees with this file: the outputs of
 *                the survey's stand-in-header build (tests/golden/ref_*.txt,
 *                PROVENANCE.md), the two tie probes quoted in SURVEY.md
 *                Appendix A, and an independent int64 recomputation
 *                (oracle.exact_int_dists).
 *   IVF path   : "parity unpinned" -- the reference IVF code cannot be compiled
 *                (arm_neon.h, QNN SDK, broken definition at IVFIndex.cpp:498) and
 *                holds no golden vectors.  Pinned only by nprobe==nlist == exact.
 *   uint8 path : "parity unpinned" (vo_q8_*, see there).
 *
 * Third-party arithmetic: the reference's dot products come from OpenBLAS
 * cblas_sgemm (un-vendored, version unpinned, cpu/cpu_baseline.cpp:229-237).
 * Its summation order is unspecified; it is restated here as an 8-lane fp32
 * FMA accumulation.  On integer-valued SIFT-range data every partial sum is an
 * integer < 2^24, so any order gives the same bits.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ fvecs */
/* cpu_baseline.cpp:31-58 read_fvecs: repeated [int32 d][d x f32]; constant d;
 * returns 0 on success, -1 cannot open, -2 inconsistent dim, -3 truncated.
 * Two-call protocol: data==NULL -> only rows/dim are filled. */
VO_API int vo_read_fvecs(const char* path, float* data, int64_t cap_floats,
                         int64_t* rows_out, int* dim_out) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    int64_t rows = 0;
    int dim = 0, d = 0;
    int rc = 0;
    for (;;) {
        size_t got = fread(&d, 1, sizeof(int), f);
        if (got == 0) break;
        if (got != sizeof(int)) { rc = -3; break; }  /* :53-56 gcount()>0 */
        if (rows == 0) dim = d; else if (d != dim) { rc = -2; break; }
        if (data) {
            if ((rows + 1) * (int64_t)dim > cap_floats) { rc = -4; break; }
            if (fread(data + rows * dim, sizeof(float), (size_t)dim, f) != (size_t)dim) { rc = -3; break; }
        } else {
            if (fseek(f, (long)dim * (long)sizeof(float), SEEK_CUR) != 0) { rc = -3; break; }
        }
        rows++;
    }
    fclose(f);
    if (rows_out) *rows_out = rows;
    if (dim_out) *dim_out = dim;
    return rc;
}

/* ------------------------------------------------------------------ norms */
/* cpu_baseline.cpp:95-114 compute_norm_avx2: 8 fp32 lanes, lane j accumulates
 * v[8i+j]^2 with FMA (_mm256_fmadd_ps == per-lane fmaf), then the lanes are
 * added left to right r0+r1+...+r7 (:106-107), then a scalar tail (:109-111). */
static float vo_norm_one(const float* v, int dim) {
    float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 7 < dim; i += 8)
        for (int j = 0; j < 8; ++j) r[j] = fmaf(v[i + j], v[i + j], r[j]);
    float sum = r[0] + r[1];
    sum = sum + r[2]; sum = sum + r[3]; sum = sum + r[4];
    sum = sum + r[5]; sum = sum + r[6]; sum = sum + r[7];
    for (; i < dim; ++i) sum += v[i] * v[i];
    return sum;
}

/* cpu_baseline.cpp:116-125 compute_norms (omp parallel for over rows) */
VO_API void vo_compute_norms(const float* data, int64_t rows, int dim, float* norms) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i) norms[i] = vo_norm_one(data + i * dim, dim);
}

/* ------------------------------------------------------------- distances */
/* Stand-in for cblas_sgemm(RowMajor, NoTrans, Trans, 1, N, d) at
 * cpu_baseline.cpp:229-237: dot[j] = sum_t q[t]*B[j][t]. 8 FMA lanes + ordered
 * horizontal add (order unpinned in the reference; exact on integer data). */
static inline float vo_dot(const float* a, const float* b, int dim) {
    float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 7 < dim; i += 8)
        for (int j = 0; j < 8; ++j) r[j] = fmaf(a[i + j], b[i + j], r[j]);
    float sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < dim; ++i) sum = fmaf(a[i], b[i], sum);
    return sum;
}

/* One query against the whole base: sgemm stand-in + the epilogue
 * dist[j] = qn + bn[j] - 2*dot[j]  (cpu_baseline.cpp:239-242).  gcc -O3 -mfma
 * contracts the reference expression to fnmadd(2, dot, qn+bn); written
 * explicitly so the oracle does not depend on -ffp-contract. */
VO_API void vo_l2_row(const float* q, float qn, const float* base, const float* bn,
                      int64_t N, int dim, float* dist) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < N; ++j) {
        float dot = vo_dot(q, base + j * dim, dim);
        dist[j] = fmaf(-2.0f, dot, qn + bn[j]);
    }
}

/* ------------------------------------------------------------- select_topk */
/* cpu_baseline.cpp:127-153.  k-slot buffer seeded with rows 0..k-1; max_idx =
 * FIRST slot holding the max (strict > scans, :134-138 and :144-148); row j
 * replaces slot max_idx iff dist[j] < buf[max_idx] (strict, :141); finally
 * std::sort by dist only (:152, operator< :16-18).  libstdc++ std::sort on
 * <= 16 elements is a pure insertion sort, i.e. stable: ties leave in slot
 * order.  For k > 16 libstdc++ uses introsort whose tie order is
 * implementation-defined: this restatement stays stable there and the tie
 * order is then unpinned.  The reference reads out of bounds when N < k; here
 * k is clamped to N and the remaining outputs are (idx=-1, dist=+inf). */
typedef struct { float dist; int idx; } vo_result;

static void vo_slot_insert_sorted(vo_result* buf, int k) {
    for (int i = 1; i < k; ++i) {           /* stable insertion sort on dist */
        vo_result v = buf[i];
        int j = i - 1;
        while (j >= 0 && v.dist < buf[j].dist) { buf[j + 1] = buf[j]; --j; }
        buf[j + 1] = v;
    }
}

VO_API void vo_select_topk(const float* dist, int64_t N, int k, int* out_idx, float* out_dist) {
    int kk = (int)((int64_t)k < N ? k : N);
    vo_result* buf = (vo_result*)malloc(sizeof(vo_result) * (size_t)(k > 0 ? k : 1));
    for (int i = 0; i < kk; ++i) { buf[i].dist = dist[i]; buf[i].idx = i; }
    if (kk > 0) {
        int max_idx = 0;
        for (int i = 1; i < kk; ++i) if (buf[i].dist > buf[max_idx].dist) max_idx = i;
        for (int64_t j = kk; j < N; ++j) {
            if (dist[j] < buf[max_idx].dist) {
                buf[max_idx].dist = dist[j]; buf[max_idx].idx = (int)j;
                max_idx = 0;
                for (int i = 1; i < kk; ++i) if (buf[i].dist > buf[max_idx].dist) max_idx = i;
            }
        }
        vo_slot_insert_sorted(buf, kk);
    }
    for (int i = 0; i < k; ++i) {
        out_idx[i] = i < kk ? buf[i].idx : -1;
        out_dist[i] = i < kk ? buf[i].dist : INFINITY;
    }
    free(buf);
}

/* Same slot algorithm over a sparse, row-ordered candidate list (rows strictly
 * increasing, must contain rows 0..k-1 and every row that the dense scan
 * would have inserted).  Used by tests to check the product's tie resolver
 * against the dense restatement. */
VO_API void vo_select_topk_sparse(const int* rows, const float* dist, int64_t M, int k,
                                  int* out_idx, float* out_dist) {
    int kk = (int)((int64_t)k < M ? k : M);
    vo_result* buf = (vo_result*)malloc(sizeof(vo_result) * (size_t)(k > 0 ? k : 1));
    for (int i = 0; i < kk; ++i) { buf[i].dist = dist[i]; buf[i].idx = rows[i]; }
    if (kk > 0) {
        int max_idx = 0;
        for (int i = 1; i < kk; ++i) if (buf[i].dist > buf[max_idx].dist) max_idx = i;
        for (int64_t j = kk; j < M; ++j) {
            if (dist[j] < buf[max_idx].dist) {
                buf[max_idx].dist = dist[j]; buf[max_idx].idx = rows[j];
                max_idx = 0;
                for (int i = 1; i < kk; ++i) if (buf[i].dist > buf[max_idx].dist) max_idx = i;
            }
        }
        vo_slot_insert_sorted(buf, kk);
    }
    for (int i = 0; i < k; ++i) {
        out_idx[i] = i < kk ? buf[i].idx : -1;
        out_dist[i] = i < kk ? buf[i].dist : INFINITY;
    }
    free(buf);
}

/* ---------------------------------------------------------- run_benchmark */
/* cpu_baseline.cpp:209-254: norms once, then a SERIAL loop over queries:
 * sgemm(1xNxd) -> epilogue -> select_topk.  t_dist_s / t_topk_s accumulate
 * the two phases like distance_times / topk_times (:245,:250). */
static double vo_now(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

VO_API int vo_search_bf(const float* base, int64_t N, int dim, const float* queries, int64_t nq,
                        int k, int* out_idx, float* out_dist, double* t_dist_s, double* t_topk_s) {
    float* bn = (float*)malloc(sizeof(float) * (size_t)N);
    float* qn = (float*)malloc(sizeof(float) * (size_t)(nq > 0 ? nq : 1));
    float* dist = (float*)malloc(sizeof(float) * (size_t)N);
    if (!bn || !qn || !dist) { free(bn); free(qn); free(dist); return -1; }
    vo_compute_norms(queries, nq, dim, qn);
    vo_compute_norms(base, N, dim, bn);
    double td = 0, tk = 0;
    for (int64_t i = 0; i < nq; ++i) {
        double t0 = vo_now();
        vo_l2_row(queries + i * dim, qn[i], base, bn, N, dim, dist);
        double t1 = vo_now();
        vo_select_topk(dist, N, k, out_idx + i * k, out_dist + i * k);
        double t2 = vo_now();
        td += t1 - t0; tk += t2 - t1;
    }
    if (t_dist_s) *t_dist_s = td;
    if (t_topk_s) *t_topk_s = tk;
    free(bn); free(qn); free(dist);
    return 0;
}

/* ------------------------------------------------------------ write_results */
/* cpu_baseline.cpp:155-175: "Query <i>: (<idx>, <dist>) ...\n"; dist printed
 * with default ostream formatting == printf("%g") (6 significant digits). */
VO_API int vo_write_results(const char* path, const int* idx, const float* dist, int64_t nq, int k) {
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    for (int64_t i = 0; i < nq; ++i) {
        fprintf(f, "Query %lld:", (long long)i);
        for (int t = 0; t < k; ++t)
            if (idx[i * k + t] >= 0) fprintf(f, " (%d, %g)", idx[i * k + t], (double)dist[i * k + t]);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}

/* -------------------------------------------------------------------- IVF */
/* IVFIndex::searchBatch, reordered branch (IVFIndex.cpp:675-784), restated
 * with the north-star's L2 metric (SURVEY.md 8a "north-star deltas"):
 *   coarse  : score_c = ||c||^2 - 2 q.c (+||q||^2), nprobe SMALLEST (reference:
 *             nth_element on largest inner product, :711; order inside the
 *             probe set is unspecified there -> here ascending (dist, id)).
 *   ranges  : (offsets[c], offsets[c+1]-offsets[c])               (:715-723)
 *   scan    : q against each contiguous range                      (:738-747,
 *             computeDotProductsContiguous :270-358), L2 via stored norms.
 *   top-k   : reference keeps a size-k heap, replaces iff strictly better
 *             (:754-766) and sorts (:771); heap tie order is implementation
 *             defined, so the restatement returns the k best by (dist, reordered
 *             position) ascending -- identical wherever distances are distinct.
 *   remap   : ids = reorder_to_original[pos]                       (:774-779)
 * nprobe is clamped to nlist (:647); k is clamped to the candidate count
 * (:735) and missing outputs are (idx=-1, dist=+inf).
 * Returns total candidates scanned (the function's return value, :858). */
typedef struct { float d; int id; } vo_pair;
static int vo_pair_cmp(const void* a, const void* b) {
    const vo_pair* x = (const vo_pair*)a; const vo_pair* y = (const vo_pair*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* metric 0 = the north-star's L2 (above); metric 1 = the reference's own inner product: coarse = the nprobe LARGEST q.c
 * (IVFIndex.cpp:697-723), candidates scored by q.v (:738-747), the k largest kept (heap :449-496, :754-766) and sorted
 * descending (:771).  Restated on s = -dot, smallest first, ties by (s, reordered position); out_dist holds s. */
VO_API int64_t vo_ivf_search_metric(const float* vectors_reordered, const float* vec_norms, int64_t N, int dim,
                                    const float* centroids, int nlist, const int32_t* offsets,
                                    const int32_t* reorder_to_original,
                                    const float* queries, int64_t nq, int k, int nprobe, int metric,
                                    int* out_idx, float* out_dist, int32_t* out_probes /* nq*nprobe or NULL */);

VO_API int64_t vo_ivf_search(const float* vectors_reordered, const float* vec_norms, int64_t N, int dim,
                             const float* centroids, int nlist, const int32_t* offsets,
                             const int32_t* reorder_to_original,
                             const float* queries, int64_t nq, int k, int nprobe,
                             int* out_idx, float* out_dist, int32_t* out_probes /* nq*nprobe or NULL */) {
    return vo_ivf_search_metric(vectors_reordered, vec_norms, N, dim, centroids, nlist, offsets, reorder_to_original, queries, nq, k,
                                nprobe, 0, out_idx, out_dist, out_probes);
}

VO_API int64_t vo_ivf_search_metric(const float* vectors_reordered, const float* vec_norms, int64_t N, int dim,
                                    const float* centroids, int nlist, const int32_t* offsets,
                                    const int32_t* reorder_to_original,
                                    const float* queries, int64_t nq, int k, int nprobe, int metric,
                                    int* out_idx, float* out_dist, int32_t* out_probes /* nq*nprobe or NULL */) {
    if (nprobe > nlist) nprobe = nlist;
    float* cn = (float*)malloc(sizeof(float) * (size_t)nlist);
    vo_compute_norms(centroids, nlist, dim, cn);
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
    for (int64_t b = 0; b < nq; ++b) {
        const float* q = queries + b * dim;
        float qn = vo_norm_one(q, dim);
        vo_pair* cs = (vo_pair*)malloc(sizeof(vo_pair) * (size_t)nlist);
        for (int c = 0; c < nlist; ++c) {
            const float cdot = vo_dot(q, centroids + (int64_t)c * dim, dim);
            cs[c].d = metric ? -cdot : fmaf(-2.0f, cdot, qn + cn[c]);
            cs[c].id = c;
        }
        qsort(cs, (size_t)nlist, sizeof(vo_pair), vo_pair_cmp);
        int64_t cand = 0;
        for (int p = 0; p < nprobe; ++p) cand += offsets[cs[p].id + 1] - offsets[cs[p].id];
        total += cand;
        vo_pair* all = (vo_pair*)malloc(sizeof(vo_pair) * (size_t)(cand > 0 ? cand : 1));
        int64_t m = 0;
        for (int p = 0; p < nprobe; ++p) {
            int c = cs[p].id;
            if (out_probes) out_probes[b * nprobe + p] = c;
            for (int32_t r = offsets[c]; r < offsets[c + 1]; ++r) {
                float dot = vo_dot(q, vectors_reordered + (int64_t)r * dim, dim);
                all[m].d = metric ? -dot : fmaf(-2.0f, dot, qn + vec_norms[r]);
                all[m].id = r;
                ++m;
            }
        }
        qsort(all, (size_t)m, sizeof(vo_pair), vo_pair_cmp);
        for (int t = 0; t < k; ++t) {
            if (t < m) {
                out_idx[b * k + t] = reorder_to_original ? reorder_to_original[all[t].id] : all[t].id;
                out_dist[b * k + t] = all[t].d;
            } else { out_idx[b * k + t] = -1; out_dist[b * k + t] = INFINITY; }
        }
        free(all); free(cs);
    }
    free(cn);
    (void)N;
    return total;
}

/* main_ivf.cpp:52-59 compute_recall: |pred[:k] n gt[:k]| / k (set overlap). */
VO_API double vo_recall(const int* pred, int npred, const int* gt, int ngt, int k) {
    int hits = 0;
    int kg = k < ngt ? k : ngt, kp = k < npred ? k : npred;
    for (int i = 0; i < kp; ++i)
        for (int j = 0; j < kg; ++j)
            if (pred[i] == gt[j]) { hits++; break; }
    return (double)hits / (double)k;
}

/* ------------------------------------------ UFIXED_POINT_8 score path (qidk) */
/* PARITY UNPINNED: the reference's quantised scores come out of a QNN graph on
 * the phone's HTP (closed converter + runtime; weight encoding and accumulator
 * requantisation are not in the reference, and it holds no recorded scores).
 * Restated from what IS in the reference: the input quantiser and the I/O
 * encodings (QnnRunner.cpp:13-55, 490-521), the top-k over raw uint8 scores
 * (main.cpp:30-57), with the quantiser's rounding rule applied to the output. */

/* QnnRunner.cpp:13-55 quantize_buffer_neon: vmulq_n_f32 (one rounding), vaddq_f32
 * 0.5 (a second one), vcvtq_s32_f32 (towards zero), saturating narrow to uint8. */
static inline uint8_t vo_q8_one(float x, float inv_scale) {
    volatile float p = x * inv_scale; /* no fused multiply-add: the NEON body has none */
    float v = p + 0.5f;
    if (!(v > 0.0f)) return 0; /* negatives and NaN -> 0 (:52-53) */
    if (v >= 255.0f) return 255;
    return (uint8_t)(int32_t)v;
}
VO_API void vo_q8_quantize(const float* src, uint8_t* dst, int64_t count, float inv_scale) {
    for (int64_t i = 0; i < count; ++i) dst[i] = vo_q8_one(src[i], inv_scale);
}
/* Database tensor, QNN scale-offset encoding real = scale * (q + offset), offset <= 0. */
VO_API void vo_q8_quantize_weights(const float* src, uint8_t* dst, int64_t count, float inv_scale, int offset) {
    for (int64_t i = 0; i < count; ++i) {
        int q = (int)vo_q8_one(src[i], inv_scale) - offset;
        dst[i] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
}
/* The graph: out[b][j] = sat_u8(trunc(ip * mult + 0.5)), ip = sum_t q8[b][t] * (w8[j][t] + w_off)
 * in int32, mult = (input_scale * weight_scale) / output_scale (fp32, computed by the caller). */
VO_API void vo_q8_scores(const uint8_t* q8, const uint8_t* w8, int B, int64_t n, int d, int w_off, float mult,
                         uint8_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j)
        for (int b = 0; b < B; ++b) {
            int32_t ip = 0;
            for (int t = 0; t < d; ++t) ip += (int32_t)q8[(size_t)b * d + t] * ((int32_t)w8[(size_t)j * d + t] + w_off);
            out[(size_t)b * n + j] = vo_q8_one((float)ip, mult);
        }
}
/* main.cpp:36-57 find_top_k_int8: a row replaces the smallest kept score only if its score is
 * strictly larger (:45); ordered by score, largest first (:53-56).  The reference leaves the
 * order (and, at the cut, the choice) among equal scores to its C++ library's heap; here rows
 * are visited in ascending order and equal scores keep ascending row order. */
VO_API int vo_q8_topk(const uint8_t* scores, int64_t n, int k, int32_t* ids, uint8_t* top) {
    int cnt = 0;
    for (int64_t j = 0; j < n; ++j) {
        const uint8_t s = scores[j];
        if (cnt == k && !(s > top[k - 1])) continue;
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && top[pos - 1] < s) { top[pos] = top[pos - 1]; ids[pos] = ids[pos - 1]; --pos; }
        top[pos] = s;
        ids[pos] = (int32_t)j;
        if (cnt < k) ++cnt;
    }
    for (int i = cnt; i < k; ++i) { ids[i] = -1; top[i] = 0; }
    return cnt;
}

VO_API int vo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
