// vsearch_bf -- drop-in for the reference's cpu_baseline CLI on the MI355X backend.
//
//   vsearch_bf                              : the reference's hard-coded run (cpu_baseline.cpp:323-345):
//                                             k = 5, siftsmall/ and sift/ relative to the CWD
//   vsearch_bf <base> <query> <k> <out> [batch]: the documented form (cpu/README.md:84)
//
// Output grammar of results is cpu_baseline.cpp:155-175; the metrics banner keeps the
// reference's section names (cpu_baseline.cpp:270-312) with per-batch instead of per-query
// latencies, because the device processes `batch` queries per pass.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/vsearch.hpp"

static void run_benchmark(const std::string& dataset_name, const std::string& base_file, const std::string& query_file,
                          int k, const std::string& output_file, int batch) {
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
        return;
    }
    std::cout << "Loading query file: " << query_file << std::endl;
    if (!vsearch::read_fvecs(query_file, Q_data, Q_rows, Q_dim)) {
        std::cerr << "Error: " << vs_last_error() << std::endl;
        std::cerr << "Failed to load query file!" << std::endl;
        return;
    }
    if (Q_dim != B_dim) {
        std::cerr << "Error: Query and Base dimensions must be equal." << std::endl;
        return;
    }
    try {
        std::cout << "Uploading base to HBM and pre-computing norms..." << std::endl;
        vsearch::ExactSearch index(B_data, B_rows, B_dim);
        index.setBatchSize(batch);
        std::vector<std::vector<vsearch::Result>> results;
        vs_timing tm{};
        auto t0 = high_resolution_clock::now();
        index.search(Q_data, Q_rows, k, results, &tm);
        auto t1 = high_resolution_clock::now();
        const double total_time = duration_cast<duration<double>>(t1 - t0).count();

        std::cout << "\n=== MI355X RAG Performance Metrics ===" << std::endl;
        std::cout << "\nDataset Information:" << std::endl;
        std::cout << "  Number of queries: " << Q_rows << std::endl;
        std::cout << "  Number of documents: " << B_rows << std::endl;
        std::cout << "  Dimension: " << Q_dim << std::endl;
        std::cout << "  Top-K: " << k << std::endl;
        std::cout << "  Batch: " << batch << std::endl;
        std::cout << "\nOverall Performance:" << std::endl;
        std::cout << "  Total execution time: " << total_time << " s" << std::endl;
        std::cout << "  Throughput: " << (Q_rows / total_time) << " queries/sec" << std::endl;
        std::cout << "\nDistance Computation + Top-K Selection (fused on device):" << std::endl;
        std::cout << "  Total time: " << tm.fine_search_ms / 1000.0 << " s" << std::endl;
        std::cout << "  Average latency: " << (tm.fine_search_ms / std::max(Q_rows, 1)) << " ms/query" << std::endl;
        std::cout << "\nTie resolution (reference select_topk order):" << std::endl;
        std::cout << "  Queries re-resolved: " << tm.tie_queries << std::endl;
        std::cout << "  Total time: " << tm.tie_resolve_ms / 1000.0 << " s" << std::endl;
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
            return;
        }
        std::cout << "\nDone processing " << dataset_name << "!\n" << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
    }
}

int main(int argc, char* argv[]) {
    std::cout << "=== MI355X (gfx950) backend for k-NN Search ===" << std::endl;
    std::cout << vs_version() << ", " << vs_device_count() << " HIP device(s)" << std::endl;
    std::cout << "====================================\n" << std::endl;
    if (argc >= 5) {
        const int k = std::stoi(argv[3]);
        const int batch = argc > 5 ? std::stoi(argv[5]) : 32;
        run_benchmark(argv[1], argv[1], argv[2], k, argv[4], batch);
    } else {
        const int k = 5;  // cpu_baseline.cpp:329
        run_benchmark("SIFT-small", "siftsmall/siftsmall_base.fvecs", "siftsmall/siftsmall_query.fvecs", k,
                      "siftsmall_results.txt", 32);
        run_benchmark("SIFT", "sift/sift_base.fvecs", "sift/sift_query.fvecs", k, "sift_results.txt", 32);
    }
    std::cout << "\n========================================" << std::endl;
    std::cout << "All benchmarks completed!" << std::endl;
    std::cout << "========================================" << std::endl;
    return 0;  // the reference always returns 0 (cpu_baseline.cpp:351)
}
