// vsearch_bf -- drop-in for the reference's brute-force CLIs on the MI355X backend.
//
//   vsearch_bf                                 : the reference's hard-coded run (cpu_baseline.cpp:323-345):
//                                                k = 5, siftsmall/ and sift/ relative to the CWD
//   vsearch_bf <base> <query> <k> <out> [batch]: the documented form (cpu/README.md:84)
//   vsearch_bf <context_binary> <queries.fvecs> <results_dir> <backend.so> <documents.fvecs> <top_k> [batch]
//                                              : the qidk_bruteforce form (main.cpp:73-85); the model and backend
//                                                slots are accepted and ignored, results_dir gets results.txt + metrics.txt
//   ... --q8[=in_scale,w_scale,w_offset,out_scale]: the qidk form through the runner's UFIXED_POINT_8 path (QnnRunner.cpp:
//                                                13-55, 608-645; main.cpp:30-57): uint8 scores, results.txt holds
//                                                (id, score8 * output_scale) with 4 decimals (main.cpp:244-246)
//   ... --gpus N                               : any form on N GPUs, one process per GPU (forked before HIP starts): the base
//                                                is row-sharded, per-shard top-(k+1) lists meet in one RCCL all-gather per
//                                                32 batches (vs_bf_search_sharded); rank 0 writes the files
//
// results: grammar of cpu_baseline.cpp:155-175.  metrics: sections of qidk_bruteforce main.cpp:321-390 with the
// device's own stages in place of the NPU's (upload instead of quantisation, scan + top-k instead of graphExecute,
// tie resolution instead of the CPU heap); per-batch statistics come from HIP-event times of the scan launches
// (one launch serves up to 32 batches).
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/vsearch.hpp"
#include "cli_ranks.hpp"

namespace {

struct RunStats {
    int Q_rows = 0, B_rows = 0, dim = 0, k = 0, batch = 32;
    double total_s = 0;
    vs_timing tm{};
    std::vector<double> batch_ms;  // device time of the scan + top-k per batch (launch time / batches in the launch)
};

double percentile(std::vector<double> v, double p) {  // main.cpp:308-314
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    size_t idx = static_cast<size_t>(p * v.size());
    if (idx >= v.size()) idx = v.size() - 1;
    return v[idx];
}

void write_metrics(const std::string& path, const RunStats& r) {
    std::ofstream m(path);
    if (!m) throw std::runtime_error("Cannot open metrics file: " + path);
    const double nq = std::max(r.Q_rows, 1);
    const size_t n_batches = (size_t)((r.Q_rows + r.batch - 1) / r.batch);
    double avg = 0, var = 0, mn = 0, mx = 0;
    if (!r.batch_ms.empty()) {
        for (double t : r.batch_ms) avg += t;
        avg /= r.batch_ms.size();
        for (double t : r.batch_ms) var += (t - avg) * (t - avg);
        var /= r.batch_ms.size();
        mn = *std::min_element(r.batch_ms.begin(), r.batch_ms.end());
        mx = *std::max_element(r.batch_ms.begin(), r.batch_ms.end());
    }
    const double flops_per_batch = 2.0 * r.batch * r.dim * (double)r.B_rows;   // main.cpp:284
    const double flops_per_query = 2.0 * r.dim * (double)r.B_rows;
    auto gflops = [&](double ms) { return ms > 0 ? flops_per_batch / (ms / 1000.0) / 1e9 : 0.0; };
    // bytes per batch: fp32 everywhere (the reference counts 1 byte per element because its NPU path is int8,
    // main.cpp:298-305); the score matrix is never written: the output is the top-k lists
    const double bytes_query = 4.0 * r.batch * r.dim;
    const double bytes_docs = 4.0 * r.dim * (double)r.B_rows + 4.0 * r.B_rows;
    const double bytes_output = 8.0 * r.batch * r.k;
    const double total_bytes = bytes_query + bytes_docs + bytes_output;
    const double oi = flops_per_batch / total_bytes;
    const double total_ms = r.total_s * 1000.0;
    const double scan_total = avg * n_batches;
    m << std::fixed << std::setprecision(6);
    m << "=== MI355X RAG Demo Performance Metrics (Batched) ===\n\n";
    m << "Dataset Information:\n  Number of queries: " << r.Q_rows << "\n  Number of documents: " << r.B_rows
      << "\n  Dimension: " << r.dim << "\n  Batch size: " << r.batch << "\n  Number of batches: " << n_batches
      << "\n  Top-K: " << r.k << "\n\n";
    m << "Operational Intensity Analysis:\n  FLOPs per batch: " << std::scientific << flops_per_batch << "\n  Bytes moved per batch: "
      << std::fixed << total_bytes << "\n    - Query input: " << bytes_query << " bytes\n    - Doc matrix (reused!): " << bytes_docs
      << " bytes\n    - Output top-k: " << bytes_output << " bytes\n  Operational Intensity: " << oi << " FLOPs/byte\n";
    if (r.batch == 1) {
        m << "  (Tip: Increase batch size to improve OI via data reuse)\n";
    } else {
        const double oi_single = flops_per_query / (4.0 * r.dim + bytes_docs + 8.0 * r.k);
        m << "  OI improvement vs batch=1: " << (oi / oi_single) << "x\n";
    }
    m << "\nOverall Performance:\n  Total execution time: " << r.total_s << " s\n  Throughput: " << (r.Q_rows / std::max(r.total_s, 1e-12))
      << " queries/sec\n\n";
    m << "GPU Execution (per batch):\n  Avg upload time: " << (r.tm.h2d_ms / std::max<size_t>(n_batches, 1))
      << " ms\n  Avg graph execute time: " << avg << " ms\n  Avg total GPU time: " << (avg + r.tm.h2d_ms / std::max<size_t>(n_batches, 1))
      << " ms\n  Std deviation (graph exec): " << std::sqrt(var) << " ms\n  Min graph exec time: " << mn
      << " ms\n  Max graph exec time: " << mx << " ms\n  P50 graph exec time: " << percentile(r.batch_ms, 0.50)
      << " ms\n  P95 graph exec time: " << percentile(r.batch_ms, 0.95) << " ms\n  P99 graph exec time: "
      << percentile(r.batch_ms, 0.99) << " ms\n\n";
    m << "GPU Performance (per batch):\n  Avg GFLOPS: " << gflops(avg) << "\n  Max GFLOPS: " << gflops(mn) << "\n  Min GFLOPS: "
      << gflops(mx) << "\n\n";
    m << "Per-Query Amortized Performance:\n  Avg GPU time per query: " << (scan_total + r.tm.h2d_ms) / nq
      << " ms\n  Avg graph exec per query: " << scan_total / nq << " ms\n  Avg CPU (tie order) per query: "
      << r.tm.tie_resolve_ms / nq << " ms\n  Avg total per query: " << total_ms / nq << " ms\n  Effective GFLOPS per query: "
      << (scan_total > 0 ? flops_per_query / (scan_total / nq / 1000.0) / 1e9 : 0.0) << "\n\n";
    m << "Tie resolution (reference select_topk order, cpu_baseline.cpp:127-153):\n  Queries re-resolved: " << r.tm.tie_queries
      << "\n  Total time: " << (r.tm.tie_resolve_ms / 1000.0) << " s\n\n";
    m << "Time Breakdown (% of end-to-end):\n  Scan + top-k (GPU):       " << (scan_total / total_ms * 100.0)
      << "%\n  Upload (host staging):    " << (r.tm.h2d_ms / total_ms * 100.0) << "%\n  Tie resolution:           "
      << (r.tm.tie_resolve_ms / total_ms * 100.0) << "%\n  Other overhead:           "
      << (100.0 - (scan_total + r.tm.h2d_ms + r.tm.tie_resolve_ms) / total_ms * 100.0) << "%\n";
}

vsearch::RankSet g_ranks;

bool run_benchmark(const std::string& dataset_name, const std::string& base_file, const std::string& query_file, int k,
                   const std::string& output_file, const std::string& metrics_file, int batch) {
    using namespace std::chrono;
    std::cout << "\n========================================" << std::endl;
    std::cout << "Processing dataset: " << dataset_name << std::endl;
    std::cout << "========================================\n" << std::endl;

    std::vector<float> Q_data, B_data;
    int Q_rows = 0, Q_dim = 0, B_rows = 0, B_dim = 0;
    std::cout << "Loading base file: " << base_file << std::endl;
    if (!vsearch::read_fvecs(base_file, B_data, B_rows, B_dim)) {
        std::cerr << "Error: " << vs_last_error() << std::endl;
        std::cerr << "Failed to load base file!" << std::endl;
        return false;
    }
    std::cout << "Loading query file: " << query_file << std::endl;
    if (!vsearch::read_fvecs(query_file, Q_data, Q_rows, Q_dim)) {
        std::cerr << "Error: " << vs_last_error() << std::endl;
        std::cerr << "Failed to load query file!" << std::endl;
        return false;
    }
    if (Q_dim != B_dim) {
        std::cerr << "Error: Query and Base dimensions must be equal." << std::endl;
        return false;
    }
    try {
        std::cout << "Uploading base to HBM and pre-computing norms..." << std::endl;
        // --gpus N: this rank keeps rows [r0, r1) of the base on device `rank`
        const int64_t r0 = vsearch::shard_bound(B_rows, g_ranks.world, g_ranks.rank);
        const int64_t r1 = vsearch::shard_bound(B_rows, g_ranks.world, g_ranks.rank + 1);
        if (r1 <= r0) throw std::runtime_error("more GPUs than 16-row tiles in the base");
        vsearch::ExactSearch index(B_data.data() + (size_t)r0 * B_dim, r1 - r0, B_dim, r0, g_ranks.rank);
        // the device scans at most 32 queries per pass: a model batch of 64 (run_all.sh's grid) is two passes
        const int dev_batch = std::min(std::max(batch, 1), 32);
        index.setBatchSize(dev_batch);
        std::vector<std::vector<vsearch::Result>> results;
        RunStats r;
        r.Q_rows = Q_rows;
        r.B_rows = B_rows;
        r.dim = Q_dim;
        r.k = k;
        r.batch = batch;
        {   // one untimed call first (the reference has none): the first launch of every kernel loads its code, which
            // costs more than the 100 queries of SIFT-small take.  Results are discarded.
            const int wn = std::min(Q_rows, 32 * dev_batch);
            std::vector<float> wq(Q_data.begin(), Q_data.begin() + (long)wn * Q_dim);
            std::vector<std::vector<vsearch::Result>> wr;
            vs_timing wt{};
            if (wn > 0) {
                if (g_ranks.world > 1) index.searchSharded(g_ranks.comm, wq, wn, k, wr, &wt);
                else index.search(wq, wn, k, wr, &wt);
            }
        }
        vsearch::check(vs_prof_enable(index.handle(), 1));
        auto t0 = high_resolution_clock::now();
        if (g_ranks.world > 1) index.searchSharded(g_ranks.comm, Q_data, Q_rows, k, results, &r.tm);
        else index.search(Q_data, Q_rows, k, results, &r.tm);
        auto t1 = high_resolution_clock::now();
        r.total_s = duration_cast<duration<double>>(t1 - t0).count();
        {   // per-batch device times: launch l served min(32, remaining) batches
            int64_t n = 0;
            vsearch::check(vs_prof_read_launches(index.handle(), 0, nullptr, 0, &n));
            std::vector<double> ms((size_t)n);
            if (n > 0) vsearch::check(vs_prof_read_launches(index.handle(), 0, ms.data(), n, &n));
            std::vector<int> sizes;  // batches per scan launch, in launch order (vs_bf_search: chunks of 32 batches + ragged tail)
            for (int left = Q_rows / dev_batch; left > 0; left -= std::min(left, 32)) sizes.push_back(std::min(left, 32));
            if (Q_rows % dev_batch) sizes.push_back(1);
            std::vector<double> pass_ms;
            for (size_t l = 0; l < sizes.size() && l < ms.size(); ++l)
                for (int b = 0; b < sizes[l]; ++b) pass_ms.push_back(ms[l] / sizes[l]);
            const size_t per = (size_t)((batch + dev_batch - 1) / dev_batch);  // passes per model batch
            for (size_t i = 0; i < pass_ms.size(); i += per) {
                double t = 0;
                for (size_t j = i; j < std::min(pass_ms.size(), i + per); ++j) t += pass_ms[j];
                r.batch_ms.push_back(t);
            }
            vsearch::check(vs_prof_enable(index.handle(), 0));
        }

        if (g_ranks.rank != 0) return true;  // rank 0 reports and writes
        std::cout << "\n=== MI355X RAG Performance Metrics ===" << std::endl;
        if (g_ranks.world > 1) std::cout << "  GPUs (row shards): " << g_ranks.world << std::endl;
        std::cout << "\nDataset Information:" << std::endl;
        std::cout << "  Number of queries: " << Q_rows << std::endl;
        std::cout << "  Number of documents: " << B_rows << std::endl;
        std::cout << "  Dimension: " << Q_dim << std::endl;
        std::cout << "  Top-K: " << k << std::endl;
        std::cout << "  Batch: " << batch << std::endl;
        std::cout << "\nOverall Performance:" << std::endl;
        std::cout << "  Total execution time: " << r.total_s << " s" << std::endl;
        std::cout << "  Throughput: " << (Q_rows / r.total_s) << " queries/sec" << std::endl;
        std::cout << "\nDistance Computation + Top-K Selection (fused on device):" << std::endl;
        std::cout << "  Total time: " << r.tm.fine_search_ms / 1000.0 << " s" << std::endl;
        std::cout << "  Average latency: " << (r.tm.fine_search_ms / std::max(Q_rows, 1)) << " ms/query" << std::endl;
        std::cout << "\nTie resolution (reference select_topk order):" << std::endl;
        std::cout << "  Queries re-resolved: " << r.tm.tie_queries << std::endl;
        std::cout << "  Total time: " << r.tm.tie_resolve_ms / 1000.0 << " s" << std::endl;
        std::cout << "\nWriting results to " << output_file << "..." << std::endl;
        std::vector<int32_t> ids((size_t)Q_rows * k, -1);
        std::vector<float> dists((size_t)Q_rows * k, 0.f);
        for (int i = 0; i < Q_rows; ++i)
            for (size_t t = 0; t < results[(size_t)i].size(); ++t) {
                ids[(size_t)i * k + t] = results[(size_t)i][t].idx;
                dists[(size_t)i * k + t] = results[(size_t)i][t].dist;
            }
        if (vs_results_write(output_file.c_str(), ids.data(), dists.data(), Q_rows, k, 0) != VS_OK) {
            std::cerr << "Failed to write results!" << std::endl;
            return false;
        }
        write_metrics(metrics_file, r);
        std::cout << "Metrics saved to: " << metrics_file << std::endl;
        std::cout << "\nDone processing " << dataset_name << "!\n" << std::endl;
        return true;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return false;
    }
}

// the qidk_bruteforce harness with the quantised runner: main.cpp:196-251 (batch loop, top-k over raw uint8 scores,
// dequantised scores in results.txt) and the core sections of its metrics.txt (main.cpp:321-352)
bool run_q8(const std::string& docs_file, const std::string& query_file, int k, const std::string& results_dir, int batch,
            const vs_q8_encodings* enc) {
    using namespace std::chrono;
    std::vector<float> Q_data, B_data;
    int Q_rows = 0, Q_dim = 0, B_rows = 0, B_dim = 0;
    if (!vsearch::read_fvecs(docs_file, B_data, B_rows, B_dim) || !vsearch::read_fvecs(query_file, Q_data, Q_rows, Q_dim)) {
        std::cerr << "Error: " << vs_last_error() << std::endl;
        return false;
    }
    if (Q_dim != B_dim) {
        std::cerr << "Error: Query and Base dimensions must be equal." << std::endl;
        return false;
    }
    try {
        vsearch::QuantizedRunner runner(B_data, B_rows, B_dim, enc, 0);
        vs_q8_encodings e{};
        vsearch::check(vs_q8_get_encodings(runner.handle(), &e));
        std::cout << "UFIXED_POINT_8 runner: input scale " << e.input_scale << ", weight scale " << e.weight_scale << " offset "
                  << e.weight_offset << ", output scale " << e.output_scale << std::endl;
        std::vector<int32_t> ids;
        std::vector<uint8_t> top;
        runner.search(Q_data, std::min(Q_rows, 32), k, ids, top);  // untimed warm-up (kernel code loads)
        const auto t0 = high_resolution_clock::now();
        runner.search(Q_data, Q_rows, k, ids, top);
        const double total_s = duration_cast<duration<double>>(high_resolution_clock::now() - t0).count();
        std::vector<float> scores((size_t)Q_rows * k);
        for (size_t i = 0; i < scores.size(); ++i) scores[i] = static_cast<float>(top[i]) * e.output_scale;  // main.cpp:244
        if (vs_results_write((results_dir + "/results.txt").c_str(), ids.data(), scores.data(), Q_rows, k, 1) != VS_OK) {
            std::cerr << "Failed to write results!" << std::endl;
            return false;
        }
        const size_t n_batches = (size_t)((Q_rows + batch - 1) / batch);
        const double flops_per_batch = 2.0 * batch * Q_dim * (double)B_rows;                                  // main.cpp:284
        const double bytes = 1.0 * batch * Q_dim + 1.0 * Q_dim * (double)B_rows + 1.0 * batch * (double)B_rows;  // main.cpp:298-305: 1 byte per element
        std::ofstream m(results_dir + "/metrics.txt");
        m << std::fixed << std::setprecision(6);
        m << "=== MI355X RAG Demo Performance Metrics (Batched, UFIXED_POINT_8) ===\n\n";
        m << "Dataset Information:\n  Number of queries: " << Q_rows << "\n  Number of documents: " << B_rows << "\n  Dimension: " << Q_dim
          << "\n  Batch size: " << batch << "\n  Number of batches: " << n_batches << "\n  Top-K: " << k << "\n\n";
        m << "Quantization:\n  Input scale: " << e.input_scale << "\n  Weight scale: " << e.weight_scale << "\n  Weight offset: "
          << e.weight_offset << "\n  Output scale: " << e.output_scale << "\n\n";
        m << "Operational Intensity Analysis:\n  FLOPs per batch: " << std::scientific << flops_per_batch << "\n  Bytes moved per batch: "
          << std::fixed << bytes << "\n  Operational Intensity: " << flops_per_batch / bytes << " FLOPs/byte\n\n";
        m << "Overall Performance:\n  Total execution time: " << total_s << " s\n  Throughput: " << (Q_rows / std::max(total_s, 1e-12))
          << " queries/sec\n  Avg total per query: " << total_s * 1000.0 / std::max(Q_rows, 1) << " ms\n";
        std::cout << "Throughput: " << (Q_rows / std::max(total_s, 1e-12)) << " queries/sec\nMetrics saved to: " << results_dir
                  << "/metrics.txt" << std::endl;
        return true;
    } catch (const std::exception& ex) {
        std::cerr << "Error: " << ex.what() << std::endl;
        return false;
    }
}

std::string metrics_name(const std::string& out) {
    const std::string tag = "_results.txt";
    if (out.size() >= tag.size() && out.compare(out.size() - tag.size(), tag.size(), tag) == 0)
        return out.substr(0, out.size() - tag.size()) + "_metrics.txt";
    return out + ".metrics.txt";
}

}  // namespace

int main(int argc, char* argv[]) {
    int status = 0;
    bool q8 = false, q8_enc_given = false;
    vs_q8_encodings q8_enc{};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a.rfind("--q8", 0) != 0) continue;
        q8 = true;
        if (a.size() > 5 && a[4] == '=') {
            q8_enc_given = std::sscanf(a.c_str() + 5, "%f,%f,%d,%f", &q8_enc.input_scale, &q8_enc.weight_scale, &q8_enc.weight_offset,
                                       &q8_enc.output_scale) == 4;
            if (!q8_enc_given) {
                std::cerr << "FATAL ERROR: --q8=<input_scale>,<weight_scale>,<weight_offset>,<output_scale>" << std::endl;
                return 1;
            }
        }
        for (int j = i; j + 1 < argc; ++j) argv[j] = argv[j + 1];
        --argc;
        break;
    }
    try {
        g_ranks = vsearch::fork_ranks(vsearch::take_gpus_flag(argc, argv));  // before anything touches HIP
        vsearch::connect_ranks(g_ranks);
    } catch (const std::exception& e) {
        std::cerr << "FATAL ERROR: " << e.what() << std::endl;
        return vsearch::join_ranks(g_ranks, 1);
    }
    std::cout << "=== MI355X (gfx950) backend for k-NN Search ===" << std::endl;
    std::cout << vs_version() << ", " << vs_device_count() << " HIP device(s), " << g_ranks.world << " rank(s)" << std::endl;
    std::cout << "====================================\n" << std::endl;
    if (argc == 7 || argc == 8) {
        // qidk_bruteforce form (main.cpp:73-85); unlike cpu_baseline this one fails loudly (main.cpp:400-403)
        const std::string results_dir = argv[3];
        int k = 0, batch = 32;
        if (!vsearch::arg_int(argv[6], k) || (argc > 7 && !vsearch::arg_int(argv[7], batch))) {
            if (g_ranks.rank == 0) std::cerr << "FATAL ERROR: <top_k> and [batch] must be integers" << std::endl;
            return vsearch::join_ranks(g_ranks, 1);
        }
        mkdir(results_dir.c_str(), 0755);
        if (q8) {
            if (g_ranks.world > 1 || !run_q8(argv[5], argv[2], k, results_dir, batch, q8_enc_given ? &q8_enc : nullptr)) {
                std::cerr << "FATAL ERROR" << (g_ranks.world > 1 ? ": --q8 runs on one GPU" : "") << std::endl;
                status = 1;
            }
        } else if (!run_benchmark(argv[5], argv[5], argv[2], k, results_dir + "/results.txt", results_dir + "/metrics.txt", batch)) {
            std::cerr << "FATAL ERROR" << std::endl;
            status = 1;
        }
    } else if (argc >= 5) {
        int k = 0, batch = 32;
        if (!vsearch::arg_int(argv[3], k) || (argc > 5 && !vsearch::arg_int(argv[5], batch))) {
            if (g_ranks.rank == 0) std::cerr << "FATAL ERROR: <k> and [batch] must be integers" << std::endl;
            return vsearch::join_ranks(g_ranks, 1);
        }
        run_benchmark(argv[1], argv[1], argv[2], k, argv[4], metrics_name(argv[4]), batch);
    } else {
        const int k = 5;  // cpu_baseline.cpp:329
        run_benchmark("SIFT-small", "siftsmall/siftsmall_base.fvecs", "siftsmall/siftsmall_query.fvecs", k,
                      "siftsmall_results.txt", "siftsmall_metrics.txt", 32);
        run_benchmark("SIFT", "sift/sift_base.fvecs", "sift/sift_query.fvecs", k, "sift_results.txt", "sift_metrics.txt", 32);
    }
    std::cout << "\n========================================" << std::endl;
    std::cout << "All benchmarks completed!" << std::endl;
    std::cout << "========================================" << std::endl;
    return vsearch::join_ranks(g_ranks, status);  // single GPU: 0 like the reference (cpu_baseline.cpp:351), except the qidk form
}
