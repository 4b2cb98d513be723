// vsearch.hpp -- header-only C++ host layer over the C ABI (vsearch.h).
//
// Mirrors the reference's C++ operator interfaces so that a caller of the reference can switch
// with the same names, argument meaning and error behaviour:
//   vsearch::read_fvecs / load_ivecs   <- cpu/cpu_baseline.cpp:31-58, main_ivf.cpp:35-50
//   vsearch::ExactSearch               <- the query loop of run_benchmark (cpu_baseline.cpp:209-254)
//                                         with QnnRunner-style getters (QnnRunner.h:37-39)
//   vsearch::QuantizedRunner           <- class QnnRunner (qidk_bruteforce QnnRunner.h:18-55), UFIXED_POINT_8 I/O
//   vsearch::IVFIndex                  <- class IVFIndex (IVFIndex.h:14-97): search / searchBatch /
//                                         SearchTiming / getNumVectors / getNumClusters / getDim
// Errors: the IVF side throws std::runtime_error like the reference (IVFIndex.cpp:184-198,
// main_ivf.cpp:287-290); the exact-search side returns bool like cpu_baseline.cpp:194-207.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <chrono>
#include <string>
#include <vector>

#include "vsearch.h"

namespace vsearch {

inline void check(int rc) {
    if (rc != VS_OK) throw std::runtime_error(vs_last_error());
}

// cpu_baseline.cpp:31-58 -- same signature, same bool/std::cerr-free contract (message via vs_last_error)
inline bool read_fvecs(const std::string& filename, std::vector<float>& data, int& rows, int& dim) {
    int64_t r = 0;
    int d = 0;
    if (vs_fvecs_shape(filename.c_str(), &r, &d) != VS_OK) return false;
    data.resize((size_t)r * (size_t)d);
    if (r > 0 && vs_fvecs_read(filename.c_str(), data.data(), (int64_t)data.size(), &r, &d) != VS_OK) return false;
    rows = (int)r;
    dim = d;
    return true;
}

// main_ivf.cpp:35-50 (throws like the reference)
inline void load_ivecs(const std::string& filename, std::vector<std::vector<int>>& vectors, int& dim) {
    int64_t r = 0;
    int d = 0;
    check(vs_fvecs_shape(filename.c_str(), &r, &d));
    std::vector<int32_t> flat((size_t)r * (size_t)d);
    if (r > 0) check(vs_ivecs_read(filename.c_str(), flat.data(), (int64_t)flat.size(), &r, &d));
    vectors.assign((size_t)r, std::vector<int>((size_t)d));
    for (int64_t i = 0; i < r; ++i)
        for (int j = 0; j < d; ++j) vectors[(size_t)i][(size_t)j] = flat[(size_t)i * d + j];
    dim = d;
}

// main_ivf.cpp:52-59
inline double compute_recall(const std::vector<int>& predicted, const std::vector<int>& ground_truth, int k) {
    int hits = 0;
    const int kg = std::min<int>(k, (int)ground_truth.size());
    for (int i = 0; i < std::min<int>(k, (int)predicted.size()); ++i)
        for (int j = 0; j < kg; ++j)
            if (predicted[(size_t)i] == ground_truth[(size_t)j]) {
                ++hits;
                break;
            }
    return static_cast<double>(hits) / k;
}

struct Result {  // cpu_baseline.cpp:13-19
    float dist;
    int idx;
    bool operator<(const Result& other) const { return dist < other.dist; }
};

class ExactSearch {
public:
    ExactSearch(const std::vector<float>& base, int rows, int dim, int device = 0, int metric = VS_METRIC_L2) {
        check(vs_bf_create(base.data(), rows, dim, metric, device, 0, &h_));
    }
    // one rank's row shard of a larger base: ids are shard rows + id_offset (multi-GPU, vs_bf_search_sharded)
    ExactSearch(const float* shard, int64_t rows, int dim, int64_t id_offset, int device, int metric = VS_METRIC_L2) {
        check(vs_bf_create(shard, rows, dim, metric, device, id_offset, &h_));
    }
    ~ExactSearch() { vs_destroy(h_); }
    ExactSearch(const ExactSearch&) = delete;
    ExactSearch& operator=(const ExactSearch&) = delete;

    size_t getNumDocs() const { return (size_t)vs_index_rows(h_); }
    size_t getDim() const { return (size_t)vs_index_dim(h_); }
    void setBatchSize(int b) { check(vs_set_batch(h_, b)); }

    // all queries -> results[i] = k nearest, ascending, reference tie order
    void search(const std::vector<float>& queries, int nq, int k, std::vector<std::vector<Result>>& results,
                vs_timing* timing = nullptr) {
        std::vector<int32_t> ids((size_t)nq * k);
        std::vector<float> dists((size_t)nq * k);
        check(vs_bf_search(h_, queries.data(), nq, k, ids.data(), dists.data(), timing));
        results.assign((size_t)nq, {});
        for (int i = 0; i < nq; ++i)
            for (int t = 0; t < k; ++t)
                if (ids[(size_t)i * k + t] >= 0) results[(size_t)i].push_back({dists[(size_t)i * k + t], ids[(size_t)i * k + t]});
    }
    // collective over `comm` (every rank: same queries, own shard); ties come out in (dist, id) order
    void searchSharded(vs_comm* comm, const std::vector<float>& queries, int nq, int k, std::vector<std::vector<Result>>& results,
                       vs_timing* timing = nullptr) {
        std::vector<int32_t> ids((size_t)nq * k);
        std::vector<float> dists((size_t)nq * k);
        check(vs_bf_search_sharded(h_, comm, queries.data(), nq, k, ids.data(), dists.data(), timing));
        results.assign((size_t)nq, {});
        for (int i = 0; i < nq; ++i)
            for (int t = 0; t < k; ++t)
                if (ids[(size_t)i * k + t] >= 0) results[(size_t)i].push_back({dists[(size_t)i * k + t], ids[(size_t)i * k + t]});
    }
    vs_index* handle() { return h_; }

private:
    vs_index* h_ = nullptr;
};

// QnnRunner with its UFIXED_POINT_8 I/O (qidk_bruteforce QnnRunner.h:18-55): same method names, the database is handed
// over as floats and quantised at construction (the reference bakes it into the model blob offline).
struct ExecutionTiming {  // QnnRunner.h:12-17
    double quantize_ms = 0.0;       // here: inside graph_execute_ms (the quantiser is a device launch)
    double graph_execute_ms = 0.0;
    double dequantize_ms = 0.0;
    double total_ms = 0.0;
};

class QuantizedRunner {
public:
    // enc == nullptr: the runner's hard-coded input / output scales (QnnRunner.cpp:490-521), weights min-max
    QuantizedRunner(const std::vector<float>& docs, int64_t rows, int dim, const vs_q8_encodings* enc = nullptr, int device = 0) {
        check(vs_q8_create(docs.data(), rows, dim, enc, device, 0, &h_));
        out_.resize((size_t)vs_q8_batch(h_) * (size_t)rows);
    }
    ~QuantizedRunner() { vs_q8_destroy(h_); }
    QuantizedRunner(const QuantizedRunner&) = delete;
    QuantizedRunner& operator=(const QuantizedRunner&) = delete;

    size_t getBatchSize() const { return (size_t)vs_q8_batch(h_); }
    size_t getDim() const { return (size_t)vs_q8_dim(h_); }
    size_t getNumDocs() const { return (size_t)vs_q8_num_docs(h_); }
    float getOutputScale() const { return vs_q8_output_scale(h_); }
    bool isFloatModel() const { return false; }

    // batch_queries: [B * dim], B <= getBatchSize() (the reference pads to the model batch with zeros, main.cpp:206-211)
    void executeBatchRaw(const std::vector<float>& batch_queries, ExecutionTiming& timing) {
        const auto t0 = std::chrono::high_resolution_clock::now();
        const int B = (int)(batch_queries.size() / getDim());
        check(vs_q8_execute(h_, batch_queries.data(), B, out_.data()));
        timing.graph_execute_ms = timing.total_ms =
            std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        rows_valid_ = B;
    }
    // [B x getNumDocs()] uint8, valid until the next execute (QnnRunner.h:37)
    const uint8_t* getRawOutputBuffer() const { return out_.data(); }
    size_t getOutputSize() const { return (size_t)rows_valid_ * getNumDocs(); }

    // executeBatchRaw + find_top_k_batch_parallel (main.cpp:59-71) on the device: ids / raw scores, [nq x k]
    void search(const std::vector<float>& queries, int64_t nq, int k, std::vector<int32_t>& ids, std::vector<uint8_t>& scores) {
        ids.assign((size_t)nq * k, -1);
        scores.assign((size_t)nq * k, 0);
        check(vs_q8_search(h_, queries.data(), nq, k, ids.data(), scores.data()));
    }
    vs_q8* handle() { return h_; }

private:
    vs_q8* h_ = nullptr;
    std::vector<uint8_t> out_;
    int rows_valid_ = 0;
};

class IVFIndex {
public:
    struct SearchTiming {  // IVFIndex.h:31-36
        double centroid_search_ms = 0.0;
        double gather_ms = 0.0;
        double fine_search_ms = 0.0;
        double total_ms = 0.0;
    };

    // backendPath is kept for signature compatibility (the reference passes libQnnHtp.so); the
    // backend here is always the HIP library this header links against.
    explicit IVFIndex(const std::string& indexDir, const std::string& backendPath = "libvsearch_hip.so",
                      int device = 0, int rank = 0, int world = 1) {
        (void)backendPath;
        check(vs_ivf_load(indexDir.c_str(), device, rank, world, &h_));
    }
    ~IVFIndex() { vs_destroy(h_); }
    IVFIndex(const IVFIndex&) = delete;
    IVFIndex& operator=(const IVFIndex&) = delete;

    size_t search(const std::vector<float>& query, int k, int nprobe, std::vector<int>& indices,
                  std::vector<float>& scores) {
        SearchTiming t;
        return search(query, k, nprobe, indices, scores, t);
    }
    size_t search(const std::vector<float>& query, int k, int nprobe, std::vector<int>& indices,
                  std::vector<float>& scores, SearchTiming& timing) {
        std::vector<std::vector<int>> ai;
        std::vector<std::vector<float>> as;
        size_t n = searchBatch(query, 1, k, nprobe, ai, as, timing);
        indices = ai[0];
        scores = as[0];
        return n;
    }
    // IVFIndex.h:45-48 -- queries holds batchSize x dim floats (zero-padded by the caller or not)
    size_t searchBatch(const std::vector<float>& queries, int batchSize, int k, int nprobe,
                       std::vector<std::vector<int>>& allIndices, std::vector<std::vector<float>>& allScores,
                       SearchTiming& timing) {
        std::vector<int32_t> ids((size_t)batchSize * k);
        std::vector<float> dists((size_t)batchSize * k);
        int64_t total = 0;
        vs_timing tm{};
        check(vs_ivf_search(h_, queries.data(), batchSize, k, nprobe, ids.data(), dists.data(), &total, &tm));
        allIndices.assign((size_t)batchSize, {});
        allScores.assign((size_t)batchSize, {});
        for (int b = 0; b < batchSize; ++b)
            for (int t = 0; t < k; ++t)
                if (ids[(size_t)b * k + t] >= 0) {
                    allIndices[(size_t)b].push_back(ids[(size_t)b * k + t]);
                    allScores[(size_t)b].push_back(dists[(size_t)b * k + t]);
                }
        timing.centroid_search_ms = tm.centroid_search_ms;
        timing.gather_ms = tm.gather_ms;
        timing.fine_search_ms = tm.fine_search_ms;
        timing.total_ms = tm.total_ms;
        return (size_t)total;
    }

    // collective over `comm`: this index holds rank's lists (constructed with the same rank / world); returns the rows
    // scanned by THIS rank
    size_t searchBatchSharded(vs_comm* comm, const std::vector<float>& queries, int batchSize, int k, int nprobe,
                              std::vector<std::vector<int>>& allIndices, std::vector<std::vector<float>>& allScores,
                              SearchTiming& timing) {
        std::vector<int32_t> ids((size_t)batchSize * k);
        std::vector<float> dists((size_t)batchSize * k);
        int64_t total = 0;
        vs_timing tm{};
        check(vs_ivf_search_sharded(h_, comm, queries.data(), batchSize, k, nprobe, ids.data(), dists.data(), &total, &tm));
        allIndices.assign((size_t)batchSize, {});
        allScores.assign((size_t)batchSize, {});
        for (int b = 0; b < batchSize; ++b)
            for (int t = 0; t < k; ++t)
                if (ids[(size_t)b * k + t] >= 0) {
                    allIndices[(size_t)b].push_back(ids[(size_t)b * k + t]);
                    allScores[(size_t)b].push_back(dists[(size_t)b * k + t]);
                }
        timing.fine_search_ms = tm.fine_search_ms;
        timing.total_ms = tm.total_ms;
        return (size_t)total;
    }

    size_t getNumVectors() const { return (size_t)vs_index_rows(h_); }
    size_t getNumClusters() const { return (size_t)vs_index_nlist(h_); }
    size_t getDim() const { return (size_t)vs_index_dim(h_); }
    void setBatchSize(int b) { check(vs_set_batch(h_, b)); }
    vs_index* handle() { return h_; }

private:
    vs_index* h_ = nullptr;
};

}  // namespace vsearch
