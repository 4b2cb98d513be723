/*
 * vsearch.h -- C ABI of libvsearch_hip.so, the MI355X (gfx950) vector-search backend.
 *
 * This is the drop-in boundary for the distance + top-k hot path of
 * zyx7k/HAI-25-RAG-on-Edge.  The reference has no FFI; its device seam is the
 * C++ class pair QnnRunner ("queries[B x d] in -> scores[B x N] out") and
 * IVFIndex ("queries in -> (ids, scores) out").  Every entry point below cites
 * the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, no exceptions: every call returns VS_OK (0) or a negative
 *     vs_status; vs_last_error() gives the message for the calling thread.
 *     (Reference: bool + std::cerr in cpu/cpu_baseline.cpp:194-207; throw
 *     std::runtime_error caught in main in main_ivf.cpp:287-290.)
 *   - "host" pointers are ordinary memory owned by the caller; "_dev" entry
 *     points take device (HBM) pointers and a hipStream_t passed as void*.
 *   - one vs_index = one GPU = one caller thread at a time (QnnRunner is not
 *     re-entrant either: shared I/O buffers, QnnRunner.cpp:322-323).  An index has ONE
 *     set of scratch buffers: calls may come on different streams, but each call's work
 *     is ordered behind the previous call's (an event the library records), so calls on
 *     one index never overlap on the device.  Use one index per concurrent stream.
 *   - ids are 0-based row numbers of the base file (cpu_baseline.cpp:130,142),
 *     or reorder_to_original[] values for IVF (IVFIndex.cpp:774-779).
 *   - the product path never falls back to a CPU implementation: without a
 *     usable HIP device every create/search call fails with VS_ERR_DEVICE.
 */
#ifndef VSEARCH_H
#define VSEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_API __attribute__((visibility("default")))

typedef enum vs_status {
    VS_OK = 0,
    VS_ERR_INVALID = -1,     /* bad argument / shape mismatch (main.cpp:121-126, main_ivf.cpp:106-109) */
    VS_ERR_IO = -2,          /* cannot open / truncated / inconsistent file (cpu_baseline.cpp:33-56)     */
    VS_ERR_DEVICE = -3,      /* HIP error or no gfx950 device                                              */
    VS_ERR_NOMEM = -4,
    VS_ERR_UNSUPPORTED = -5  /* e.g. dim != 128, k too large for the compiled kernels                      */
} vs_status;

typedef enum vs_metric {
    VS_METRIC_L2 = 0,        /* squared L2, smallest wins (cpu_baseline.cpp:239-242) -- every graded config */
    VS_METRIC_IP = 1         /* raw inner product, largest wins (qidk main.cpp:30-57, IVFIndex.cpp:449-496) */
} vs_metric;

typedef struct vs_index vs_index; /* opaque: owns device memory, scratch, streams */

/* Mirrors IVFIndex::SearchTiming (IVFIndex.h:31-36) plus the distance / top-k
 * split cpu_baseline prints (cpu_baseline.cpp:281-299).  All in milliseconds,
 * accumulated over the call. */
typedef struct vs_timing {
    double centroid_search_ms; /* IVF coarse stage (IVFIndex.cpp:654-666)                  */
    double gather_ms;          /* probe selection (IVFIndex.cpp:694-726)                   */
    double fine_search_ms;     /* scan + top-k; for brute force: the whole device pipeline */
    double total_ms;           /* wall time of the call                                    */
    double h2d_ms;             /* query upload                                             */
    double d2h_ms;             /* result download                                          */
    double tie_resolve_ms;     /* exact select_topk slot emulation for flagged queries     */
    int64_t tie_queries;       /* how many queries needed it                               */
} vs_timing;

/* ------------------------------------------------------------------ library */
VS_API const char* vs_version(void);
VS_API const char* vs_last_error(void);           /* thread-local, never NULL */
VS_API int vs_device_count(void);                 /* 0 when no HIP device is visible */

/* -------------------------------------------------------------- file formats */
/* .fvecs / .ivecs: repeated [int32 d][d x 4 bytes], little endian
 * (cpu_baseline.cpp:31-58 read_fvecs; main_ivf.cpp:18-50 load_fvecs/load_ivecs).
 * *_shape fills rows/dim only; *_read needs cap_elems >= rows*dim. */
VS_API int vs_fvecs_shape(const char* path, int64_t* rows, int* dim);
VS_API int vs_fvecs_read(const char* path, float* dst, int64_t cap_elems, int64_t* rows, int* dim);
VS_API int vs_ivecs_read(const char* path, int32_t* dst, int64_t cap_elems, int64_t* rows, int* dim);
VS_API int vs_fvecs_write(const char* path, const float* src, int64_t rows, int dim);
VS_API int vs_ivecs_write(const char* path, const int32_t* src, int64_t rows, int dim);

/* results.txt: "Query <i>: (<id>, <dist>) ...\n".
 * style 0 = cpu_baseline.cpp:155-175 (default ostream float formatting),
 * style 1 = main_ivf.cpp:179-183 (std::fixed, 4 decimals).  ids < 0 are skipped. */
VS_API int vs_results_write(const char* path, const int32_t* ids, const float* dists,
                            int64_t nq, int k, int style);

/* Deterministic SIFT-shaped synthetic data: integer-valued
 * f32 in [0, 218], clustered.  Row i depends only on (seed, i), so any slice
 * can be generated independently.  row_begin lets ranks generate shards. */
VS_API int vs_synth_sift(float* dst, int64_t row_begin, int64_t rows, int dim, uint64_t seed);
/* The same generator with the mixture as parameters: x = clip(rint(c_u + N(0, row_sigma^2)), 0, 218), u uniform over
 * n_centers centres c = |N(0, center_sigma^2)| (vs_synth_sift = 4096 centres, 40, 18: strongly clustered, an IVF index
 * reaches recall ~ 1 with a handful of probes).  Many centres and a wide row_sigma give weakly clustered data on which
 * recall falls well below 1: the second distribution of the IVF tests and bench. */
VS_API int vs_synth_mixture(float* dst, int64_t row_begin, int64_t rows, int dim, uint64_t seed, int n_centers,
                            double center_sigma, double row_sigma);

/* The reference's select_topk (cpu_baseline.cpp:127-153) applied to a sparse,
 * row-ordered candidate list; exposed so the tie resolver can be unit-tested. */
VS_API int vs_select_topk_slots(const int32_t* rows, const float* dists, int64_t m, int k,
                                int32_t* out_ids, float* out_dists);

/* -------------------------------------------------------- exact brute force */
/* Replaces the body of run_benchmark's query loop (cpu_baseline.cpp:209-254):
 * norms + cblas_sgemm(1 x N x d) + L2 epilogue + select_topk, and the QNN
 * "database baked into the model" runner (QnnRunner ctor, QnnRunner.h:20).
 * The base is copied to HBM once; id_offset is added to every returned id
 * (row-sharded multi-GPU: each rank passes its shard and its first row). */
VS_API int vs_bf_create(const float* base_host, int64_t n_rows, int dim, int metric,
                        int device, int64_t id_offset, vs_index** out);

/* Fixed model batch, like QnnRunner::getBatchSize (QnnRunner.h:37); 1..32, default 32.
 * Larger query sets are processed in batches of this size, the last one
 * zero-padded (main.cpp:206-211). */
VS_API int vs_set_batch(vs_index* h, int batch);

/* Data path of the scan.  0 = auto (default): when every base value is an integer in [0, 255] -- true for
 * SIFT -- the index also keeps the rows as bytes and scans them with int8 MFMA; distances are then
 * computed in int32 and are the same integers the fp32 path produces exactly, at a quarter of the memory
 * traffic.  A batch containing a non-integer query is detected on the device, skipped and rerun in fp32
 * (vs_bf_search does that itself; the *_dev calls report it as flags == 2).  1 = force fp32 (the reference
 * arithmetic, cblas_sgemm + epilogue); 2 = require int8 (VS_ERR_UNSUPPORTED when the base does not allow it).
 * IVF indexes alike: 1 = the list scan reads the fp32 rows (IVFIndex.cpp:270-358's arithmetic). */
VS_API int vs_set_precision(vs_index* h, int precision);
/* IVF indexes are created with the north-star's squared L2; VS_METRIC_IP selects the reference's own ranking (IVFIndex.cpp:
 * 449-496, :697-723): nprobe lists of LARGEST q.c, k candidates of largest q.v (fp32 rows; the int8 copies are an L2 device).
 * The host call returns the scores q.v (descending); the *_dev calls return -q.v (ascending), like brute force. */
VS_API int vs_ivf_set_metric(vs_index* h, int metric);

/* Host-buffer search with the reference's exact semantics: ids/dists are
 * [nq x k], ascending distance, ties ordered exactly as select_topk leaves
 * them (flagged queries are re-resolved, see DESIGN.md "Ties").  k clamps to
 * n_rows; unused slots are id -1 / dist +inf. */
VS_API int vs_bf_search(vs_index* h, const float* queries_host, int64_t nq, int k,
                        int32_t* ids, float* dists, vs_timing* timing);

/* Device-level, asynchronous on `stream`: one batch (B <= batch) of queries
 * already in HBM -> the k+1 best (dist, id) per query by (dist, id) ascending,
 * [B x (k+1)], plus flags[B]: 1 where two of those distances are equal (the
 * caller must then use vs_bf_search for reference tie order), 2 where the int8
 * path had to skip the batch (rerun with vs_set_precision(h, 1)).  This is the
 * analogue of QnnRunner::executeBatchRaw + getRawOutputBuffer (QnnRunner.h:28-30)
 * with the top-k fused in, so the B x N score matrix never exists. */
VS_API int vs_bf_search_dev(vs_index* h, const float* queries_dev, int B, int k,
                            int32_t* ids_dev, float* dists_dev, int32_t* flags_dev, void* stream);

/* The same for n_batches consecutive batches of exactly B queries each
 * (queries_dev [n_batches*B x d], outputs [n_batches*B x (k+1)], flags [n_batches*B]):
 * the harness loop of main.cpp:201-251 in one call.  Every batch still streams the
 * base once; groups of up to 32 batches share ONE persistent scan launch (preceded by
 * three small bound-seeding launches, followed by one merge launch), all on `stream`,
 * so no launch gap, grid fill or drain separates consecutive batches. */
VS_API int vs_bf_search_dev_multi(vs_index* h, const float* queries_dev, int n_batches, int B, int k,
                                  int32_t* ids_dev, float* dists_dev, int32_t* flags_dev, void* stream);

/* QnnRunner::executeBatchRaw proper (QnnRunner.cpp:683-724): the raw
 * [B x ld] score matrix, scores_dev[b*ld + j] = dist(query b, row j), ld >= n_rows. */
VS_API int vs_bf_scores_dev(vs_index* h, const float* queries_dev, int B,
                            float* scores_dev, int64_t ld, void* stream);

/* ------------------------------------------- quantised score path (UFIXED_POINT_8) */
/* The reference's device runner with its uint8 I/O (qidk_bruteforce/android/app/main/jni):
 * QnnRunner ctor (QnnRunner.h:20) bakes the database into the graph as uint8 weights;
 * executeBatchRaw (QnnRunner.cpp:608-645) quantises a [B x d] batch with
 * quantize_buffer_neon (QnnRunner.cpp:13-55: q8 = sat_u8(trunc(x * (1 / input_scale) + 0.5)): a multiplication by the
 * reciprocal and an addition, rounded separately, as QnnRunner.cpp:544 computes it --
 * offset 0), runs the graph and leaves the raw uint8 [B x N] inner-product scores in
 * its output buffer (getRawOutputBuffer, QnnRunner.h:37); the harness takes the k
 * largest per query (find_top_k_int8, main.cpp:30-57) and prints score * output_scale
 * (main.cpp:244-246).  Here: real = scale * (q + offset) for all three tensors (QNN's
 * scale-offset encoding, QnnRunner.cpp:490-508; input and output offsets are 0 there),
 *   ip      = sum_t q8[t] * (w8[t] + weight_offset)                      (int32, exact)
 *   score8  = sat_u8(trunc(ip * ((input_scale * weight_scale) / output_scale) + 0.5))
 * with every product and sum rounded separately in fp32 (no fused multiply-add), the
 * rounding rule of the reference's own quantiser.  What the closed QNN converter / HTP
 * runtime do inside the graph (weight encoding, accumulator requantisation) is not in the
 * reference: results are checked against oracle/ (vo_q8_*), parity unpinned.
 * Equal scores are returned in ascending id order (the reference: whatever its C++
 * library's heap leaves). */
typedef struct vs_q8 vs_q8; /* opaque: quantised database, I/O buffers, stream */
typedef struct vs_q8_encodings {
    float input_scale;     /* QnnRunner.cpp:490  0.6627451181411743  */
    float weight_scale;    /* database tensor (inside the reference's model blob) */
    int32_t weight_offset; /* <= 0, real = scale * (q + offset) */
    float output_scale;    /* QnnRunner.cpp:507  1013.43121337890625 */
} vs_q8_encodings;

/* enc == NULL: the runner's hard-coded input / output scales and weight_scale =
 * max(database) / 255, offset 0.  id_offset as in vs_bf_create. */
VS_API int vs_q8_create(const float* base_host, int64_t n_rows, int dim, const vs_q8_encodings* enc,
                        int device, int64_t id_offset, vs_q8** out);
VS_API void vs_q8_destroy(vs_q8* h);
VS_API int64_t vs_q8_num_docs(const vs_q8* h);     /* QnnRunner::getNumDocs   (QnnRunner.h:52) */
VS_API int vs_q8_dim(const vs_q8* h);              /* QnnRunner::getDim       (QnnRunner.h:50) */
VS_API int vs_q8_batch(const vs_q8* h);            /* QnnRunner::getBatchSize (QnnRunner.h:48): 32 = the largest B */
VS_API float vs_q8_output_scale(const vs_q8* h);   /* QnnRunner::getOutputScale (QnnRunner.h:41) */
VS_API int vs_q8_get_encodings(const vs_q8* h, vs_q8_encodings* out);

/* executeBatchRaw + getRawOutputBuffer: scores_dev[b*ld + j] = score8(query b, row j),
 * 1 <= B <= 32, ld >= rows (16-byte stores when ld and scores_dev are multiples of 16). */
VS_API int vs_q8_execute_dev(vs_q8* h, const float* queries_dev, int B, uint8_t* scores_dev,
                             int64_t ld, void* stream);
/* The same through host buffers: scores_host is [B x rows], dense. Blocking. */
VS_API int vs_q8_execute(vs_q8* h, const float* queries_host, int B, uint8_t* scores_host);

/* executeBatchRaw + find_top_k_batch_parallel (main.cpp:59-71) for n_batches batches of B
 * queries: ids_dev [n_batches*B x k] (-1 where the database has fewer than k rows),
 * scores_dev [n_batches*B x k] uint8, largest first.  k <= 16. */
VS_API int vs_q8_search_dev(vs_q8* h, const float* queries_dev, int n_batches, int B, int k,
                            int32_t* ids_dev, uint8_t* scores_dev, void* stream);
/* Host-buffer form: the harness loop of main.cpp:201-251 (batches of 32, the last one
 * short). Blocking. */
VS_API int vs_q8_search(vs_q8* h, const float* queries_host, int64_t nq, int k, int32_t* ids,
                        uint8_t* scores);

/* ---------------------------------------------------------------------- IVF */
/* IVFIndex ctor (IVFIndex.cpp:154-177): reads ivf_config.json, cluster_offsets.npy,
 * vectors_reordered.npy, reorder_to_original.npy (reordered mode) and
 * centroids.npy (the reference bakes centroids into centroids.bin for the NPU,
 * IVFIndex.cpp:167-169).  rank/world select the lists this GPU owns
 * (cluster-sharded search); 0/1 = everything. */
VS_API int vs_ivf_load(const char* index_dir, int device, int rank, int world, vs_index** out);

/* Same, from arrays in host memory (layout of create_ivf_model_reordered.py:141-169). */
VS_API int vs_ivf_create(const float* vectors_reordered, int64_t n_rows, int dim,
                         const float* centroids, int nlist, const int32_t* cluster_offsets,
                         const int32_t* reorder_to_original, int device, int rank, int world,
                         vs_index** out);

/* GPU index builder, the device side of build_ivf_index_reordered (create_ivf_model_reordered.py:82-177):
 * Lloyd k-means under L2 with sklearn's stopping rule (sum of squared centre shifts <= tol * mean feature
 * variance, default tol 1e-4, max_iter 100 in the reference, :97-103).  Assignment runs on the MFMA scan
 * kernel, the update uses fixed-point integer atomics (deterministic).  Initial centres: k-means++ (D^2 sampling
 * on the GPU, sklearn's default init; its RNG stream and greedy multi-trial variant are not reproduced, so the
 * centres are statistically, not bitwise, sklearn's; VSEARCH_KMEANS_INIT=random = nlist distinct random rows).
 * Outputs: centroids_out [nlist x dim], assign_out [n_rows] (cluster of every row); vs_ivf_layout turns the
 * assignment into the reordered layout of :108-128, vs_ivf_build_index does all of it. */
VS_API int vs_ivf_build(const float* base_host, int64_t n_rows, int dim, int nlist, int max_iter, double tol,
                        uint64_t seed, int device, float* centroids_out, int32_t* assign_out, int* iters_done);

/* The host side of the same builder (no GPU needed).  vs_ivf_clamp_nlist: the nlist rule of :92-94
 * (nlist > N/10 -> max(16, N/100)).  vs_ivf_layout: :108-128 -- rows sorted by cluster id (stable), cluster_offsets
 * [nlist+1] = running sum of the cluster sizes, reorder_to_original [n_rows] = the sort permutation
 * (vectors_reordered[i] = base[reorder_to_original[i]]). */
VS_API int vs_ivf_clamp_nlist(int64_t n_vectors, int nlist);
VS_API int vs_ivf_layout(const int32_t* assign, int64_t n_rows, int nlist, int32_t* cluster_offsets,
                         int32_t* reorder_to_original);

/* build_ivf_index_reordered end to end (:82-177): nlist clamp, k-means (vs_ivf_build), reordered layout, and the
 * resulting index resident on `device`; vs_ivf_save then writes the reference's directory.  No Python involved. */
VS_API int vs_ivf_build_index(const float* base_host, int64_t n_rows, int dim, int nlist, int max_iter, double tol,
                              uint64_t seed, int device, vs_index** out, int* iters_done);

/* Writes the index held by h in the reference's directory format. */
VS_API int vs_ivf_save(vs_index* h, const char* index_dir);

/* IVFIndex::searchBatch (IVFIndex.h:45-48, IVFIndex.cpp:640-859): ids/dists
 * [nq x k] (L2: ascending distance; ids are original row numbers), returns the
 * total number of candidates scanned through *total_candidates (the
 * function's return value in the reference) and the SearchTiming fields. */
VS_API int vs_ivf_search(vs_index* h, const float* queries_host, int64_t nq, int k, int nprobe,
                         int32_t* ids, float* dists, int64_t* total_candidates, vs_timing* timing);

/* Device-level, asynchronous: one batch, outputs [B x k] on the device.
 * For a sharded index the outputs are this shard's local top-k (global ids). */
VS_API int vs_ivf_search_dev(vs_index* h, const float* queries_dev, int B, int k, int nprobe,
                             int32_t* ids_dev, float* dists_dev, void* stream);

/* The batch loop of main_ivf.cpp:150-214 with the queries already on the device: n_batches independent
 * batches [n_batches][B][dim] -> [n_batches][B][k].  Every kernel of the pipeline is launched once for a
 * launch group of up to 32 batches (1024 queries share ONE list-major pass over the probed lists), and the
 * groups of a call are dealt to two internal streams that fork from and join `stream`: pass as many batches
 * per call as there are (128 batches per call: 13.6 M QPS at nlist 1024 / nprobe 32; 32 per call: 10.9 M). */
VS_API int vs_ivf_search_dev_multi(vs_index* h, const float* queries_dev, int n_batches, int B, int k, int nprobe,
                                   int32_t* ids_dev, float* dists_dev, void* stream);

/* ---------------------------------------------------------------- multi-GPU */
/* Which rank owns which inverted list in a cluster-sharded index (host only, no GPU needed):
 * lists sorted by length, longest first, dealt round-robin -> owner_out[nlist] in [0, world).
 * vs_ivf_create / vs_ivf_load use exactly this assignment. */
VS_API int vs_ivf_list_owners(const int32_t* cluster_offsets, int nlist, int world, int32_t* owner_out);

/* Merge G per-shard sorted lists (e.g. the receive buffer of an RCCL
 * all-gather) into [B x kout] by (dist, id) ascending; flags as above when
 * flags_dev != NULL.  Entry (g, b, j) lives at g*stride_g + b*kin + j in both
 * arrays; stride_g = 0 means the dense layout B*kin.  Runs on the current
 * device, asynchronously on `stream`. */
VS_API int vs_topk_merge_dev(const float* dists_dev, const int32_t* ids_dev, int G, int B, int kin,
                             int64_t stride_g, int kout, float* out_dists_dev, int32_t* out_ids_dev,
                             int32_t* flags_dev, void* stream);

/* One process per GPU.  A vs_comm owns an RCCL communicator over the ranks of the job, a stream for
 * the collective and the exchange buffers.  Bootstrap like any NCCL program: rank 0 calls vs_comm_unique_id and
 * hands the VS_COMM_ID_BYTES bytes to the other ranks by whatever channel the host program has (MPI, a pipe, a file,
 * torch.distributed's store), then EVERY rank calls vs_comm_create (collective).  RCCL is loaded on first use
 * (dlopen of librccl.so.1): single-GPU users never need it.  The reference is single device; this replaces nothing
 * there -- it is the north-star's "cluster-sharded IVF / row-sharded brute force over the 8 GPUs of a node". */
#define VS_COMM_ID_BYTES 128
typedef struct vs_comm vs_comm;
VS_API int vs_comm_unique_id(void* id_out /* VS_COMM_ID_BYTES */);
VS_API int vs_comm_create(const void* unique_id, int rank, int world, int device, vs_comm** out);
VS_API int vs_comm_rank(const vs_comm* c);
VS_API int vs_comm_world(const vs_comm* c);
VS_API void vs_comm_destroy(vs_comm* c);

/* Sharded search, collective over the communicator: every rank passes the SAME queries (device memory) and its own
 * shard -- vs_bf_create(rows of this rank, id_offset = first row) or vs_ivf_create/load(..., rank, world) -- and every
 * rank receives the merged global result.  Per launch group of up to 32 batches: local scan -> ONE ncclAllGather of
 * the per-shard top-(k+1) [brute force] / top-k [IVF] lists (dists and ids as 32-bit words) -> device merge by
 * (dist, id); all-gather + merge of group g run on the communicator's stream beside the local scan of group g + 1.
 * Outputs and flags as vs_bf_search_dev_multi / vs_ivf_search_dev_multi (flags = 2: no shard produced a result for
 * the query because the int8 path skipped its batch).  Asynchronous on `stream`. */
VS_API int vs_bf_search_dev_sharded(vs_index* h, vs_comm* c, const float* queries_dev, int n_batches, int B, int k,
                                    int32_t* ids_dev, float* dists_dev, int32_t* flags_dev, void* stream);
VS_API int vs_ivf_search_dev_sharded(vs_index* h, vs_comm* c, const float* queries_dev, int n_batches, int B, int k,
                                     int nprobe, int32_t* ids_dev, float* dists_dev, void* stream);
/* The cluster-sharded IVF call (BASELINE configs[4]; IVFIndex::searchBatch, IVFIndex.cpp:640-859, over `world` ranks).
 * A launch group is cut into `world` slices of up to 32 batches.  Rank r alone runs the per-query stages of slice r
 * (centroid scores, top-nprobe, the bound), ONE all-gather exchanges the slices' results (nprobe + 2 words per query),
 * every rank then scans its resident lists for ALL slices and ranks its candidates, and the all-gather of top-k lists
 * + merge finishes the group.  Per launch group a rank so does what an unsharded index does for ONE slice (plus the
 * ranking of every query over an eighth of the candidates).  The exchanges of group g run beside the front half of
 * group g + 1.  Indexes with nlist > 4096 or fewer lists than ranks run the whole pipeline per rank (one all-gather).
 *
 * Virtual ranks: the same pipeline for G shards (vs_ivf_create / vs_ivf_load with rank r, world G) on ONE device, driven
 * by the calling thread, the collectives replaced by writing into the gathered layout.  For tests and for measuring
 * a rank's cost per launch group on one GPU: rank_ms[r] (optional, G doubles) = device time of rank r's two halves. */
/* Host-side arithmetic of the sliced pipeline (no device needed).  vs_ivf_shard_group: batches per launch group of an index
 * created with `world`.  vs_ivf_shard_slice: a group of n_batches batches is cut into `world` slices of *slice_batches
 * batches (the last ones may be short or empty); rank's own slice is [*first_batch, *first_batch + *own_batches).
 * vs_ivf_shard_block_words: 32-bit words of the block a rank contributes to the exchange between the two halves:
 * probes [slice_batches * 32][nprobe] (int32, batch padded) | bounds [slice_batches * 32] (f32) | slow marks [.. * 32]. */
VS_API int vs_ivf_shard_group(int world);
VS_API int vs_ivf_shard_slice(int n_batches, int world, int rank, int32_t* slice_batches, int32_t* first_batch, int32_t* own_batches);
VS_API int64_t vs_ivf_shard_block_words(int slice_batches, int nprobe);
VS_API int vs_ivf_search_dev_vshards(vs_index* const* shards, int G, const float* queries_dev, int n_batches, int B, int k,
                                     int nprobe, int32_t* ids_dev, float* dists_dev, double* rank_ms, void* stream);

/* Host-buffer forms (what the CLIs run with --gpus N): same contract, queries and results in host memory on every
 * rank.  Brute force returns what vs_bf_search returns on one GPU -- the reference's answer (cpu_baseline.cpp:127-153):
 * a chunk whose int8 scan was skipped on some shard (merged flag 2) is rerun on the fp32 rows by every rank, and
 * queries with a tie inside the k+1 best are re-resolved exactly: shard 0's first rows are taken densely, their k-th
 * smallest distance bounds select_topk's buffer maximum for every later row, every shard filters its rows under that
 * bound, the candidates are exchanged (all-gathers of fixed-size buffers) and every rank replays the slots over "dense
 * rows, then candidates in row order".  Shards must be contiguous row ranges in rank order.
 * IVF: *total_candidates = rows scanned by THIS rank's shard.
 * vs_bf_search_vshards: the brute-force call for G shards on ONE device, driven by the calling thread (no collective):
 * tests of the sharded tie order without a multi-GPU node. */
VS_API int vs_bf_search_sharded(vs_index* h, vs_comm* c, const float* queries_host, int64_t nq, int k, int32_t* ids,
                                float* dists, vs_timing* timing);
VS_API int vs_ivf_search_sharded(vs_index* h, vs_comm* c, const float* queries_host, int64_t nq, int k, int nprobe,
                                 int32_t* ids, float* dists, int64_t* total_candidates, vs_timing* timing);
VS_API int vs_bf_search_vshards(vs_index* const* shards, int G, const float* queries_host, int64_t nq, int k, int32_t* ids,
                                float* dists, vs_timing* timing);

/* ------------------------------------------------------------------ profiling */
/* HIP-event timing of the dominant scan kernel on the stream it is launched on
 * (bench.py's roofline leg).  which: 0 = brute-force scan, 1 = IVF list scan. */
VS_API int vs_prof_enable(vs_index* h, int on);
VS_API int vs_prof_read(vs_index* h, int which, double* total_ms, int64_t* launches);
/* per-launch durations (ms) of the same window, oldest first; *launches = how many there were (may exceed cap).
 * One brute-force launch serves up to 32 batches: the CLIs turn these into the per-batch statistics of
 * main.cpp:262-330 (avg / stddev / min / max / P50 / P95 / P99 "graph execute time"). */
VS_API int vs_prof_read_launches(vs_index* h, int which, double* ms_out, int64_t cap, int64_t* launches);

VS_API int64_t vs_index_rows(const vs_index* h);
VS_API int vs_index_dim(const vs_index* h);
VS_API int vs_index_nlist(const vs_index* h);     /* 0 for brute force */
VS_API void vs_destroy(vs_index* h);

#ifdef __cplusplus
}
#endif
#endif /* VSEARCH_H */
