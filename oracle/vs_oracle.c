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
 *   exact path : pinned by the reference outputs recorded during the survey
 *                (tests/golden/ref_*.txt, provenance in tests/golden/PROVENANCE.md)
 *                and by an independent int64 recomputation (oracle/exact_int.py).
 *                The reference itself is NOT rebuilt by this repo: it needs
 *                <cblas.h>, which this image lacks, and stand-in headers are
 *                not allowed (see DESIGN.md "Oracle").
 *   IVF path   : "parity unpinned" -- the reference IVF code cannot be compiled
 *                (arm_neon.h, QNN SDK, broken definition at IVFIndex.cpp:498) and
 *                holds no golden vectors.  Pinned only by nprobe==nlist == exact.
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

VO_API int64_t vo_ivf_search(const float* vectors_reordered, const float* vec_norms, int64_t N, int dim,
                             const float* centroids, int nlist, const int32_t* offsets,
                             const int32_t* reorder_to_original,
                             const float* queries, int64_t nq, int k, int nprobe,
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
            cs[c].d = fmaf(-2.0f, vo_dot(q, centroids + (int64_t)c * dim, dim), qn + cn[c]);
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
                all[m].d = fmaf(-2.0f, dot, qn + vec_norms[r]);
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

VO_API int vo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
