// vs_host.h -- host-side pieces of libvsearch_hip.so: file formats, synthetic data, tie resolver.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace vs {

void set_error(const std::string& msg);
const char* get_error();

// .npy (v1/v2/v3 headers) -- what IVFIndex.cpp:52-152 parses, but validating descr / C order.
bool npy_read_f32(const std::string& path, std::vector<float>& data, std::vector<int64_t>& shape);
bool npy_read_i32(const std::string& path, std::vector<int32_t>& data, std::vector<int64_t>& shape);
bool npy_write(const std::string& path, const void* data, const char* descr, const std::vector<int64_t>& shape,
               size_t elem_size);

// ivf_config.json (create_ivf_model_reordered.py:148-160; read like IVFIndex.cpp:181-204)
struct IvfConfig {
    int64_t n_vectors = 0, n_clusters = 0, dim = 0, batch_size = 32;
    double avg_cluster_size = 0;
    int64_t min_cluster_size = 0, max_cluster_size = 0;
    bool reordered = false;
};
bool ivf_config_read(const std::string& path, IvfConfig& cfg);
bool ivf_config_write(const std::string& path, const IvfConfig& cfg);

// cpu_baseline.cpp:127-153 over a dense distance row / a sparse row-ordered candidate list.
void select_topk_slots_dense(const float* dist, int64_t n, int k, int32_t id_offset, int32_t* out_ids,
                             float* out_dists);
void select_topk_slots_sparse(const int32_t* rows, const float* dist, int64_t m, int k, int32_t* out_ids,
                              float* out_dists);

}  // namespace vs
