// vs_host.cpp -- file formats, synthetic data and the select_topk slot emulation (host only).
#include "vs_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>
#include <thread>

#include "../../include/vsearch.h"

namespace vs {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* get_error() { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------ xvecs
// cpu_baseline.cpp:31-58: records of [int32 d][d x 4 bytes]; d constant; a partial trailing
// record is "truncated".  Returns VS_OK / VS_ERR_IO.
static int xvecs_scan(const char* path, void* dst, int64_t cap_elems, int64_t* rows_out, int* dim_out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) {
        set_error(std::string("Cannot open file ") + path);
        return VS_ERR_IO;
    }
    std::fseek(f, 0, SEEK_END);
    const int64_t fsize = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    int64_t rows = 0;
    int dim = 0;
    int rc = VS_OK;
    if (fsize >= 4) {
        int32_t d0 = 0;
        if (std::fread(&d0, 4, 1, f) != 1 || d0 <= 0) {
            set_error(std::string("bad leading dimension in ") + path);
            rc = VS_ERR_IO;
        } else {
            dim = d0;
            const int64_t rec = 4 + 4 * (int64_t)dim;
            rows = fsize / rec;
            if (fsize % rec != 0) {
                set_error(std::string("File seems truncated: ") + path);  // cpu_baseline.cpp:53-56
                rc = VS_ERR_IO;
            }
            if (rc == VS_OK && dst) {
                if (rows * dim > cap_elems) {
                    set_error("destination too small for " + std::string(path));
                    rc = VS_ERR_INVALID;
                } else {
                    std::fseek(f, 0, SEEK_SET);
                    std::vector<char> buf((size_t)rec * 1024);
                    char* out = static_cast<char*>(dst);
                    int64_t done = 0;
                    while (done < rows && rc == VS_OK) {
                        const int64_t n = std::min<int64_t>(1024, rows - done);
                        if ((int64_t)std::fread(buf.data(), (size_t)rec, (size_t)n, f) != n) {
                            set_error(std::string("short read in ") + path);
                            rc = VS_ERR_IO;
                            break;
                        }
                        for (int64_t i = 0; i < n; ++i) {
                            int32_t d;
                            std::memcpy(&d, buf.data() + i * rec, 4);
                            if (d != dim) {
                                set_error("Inconsistent dimension.");  // cpu_baseline.cpp:43-46
                                rc = VS_ERR_IO;
                                break;
                            }
                            std::memcpy(out + (done + i) * 4 * (int64_t)dim, buf.data() + i * rec + 4, 4 * (size_t)dim);
                        }
                        done += n;
                    }
                }
            }
        }
    } else if (fsize > 0) {
        set_error(std::string("File seems truncated: ") + path);
        rc = VS_ERR_IO;
    }
    std::fclose(f);
    if (rows_out) *rows_out = rows;
    if (dim_out) *dim_out = dim;
    return rc;
}

static int xvecs_write(const char* path, const void* src, int64_t rows, int dim) {
    FILE* f = std::fopen(path, "wb");
    if (!f) {
        set_error(std::string("Cannot open output file ") + path);
        return VS_ERR_IO;
    }
    const char* in = static_cast<const char*>(src);
    const int32_t d = dim;
    std::vector<char> rec(4 + 4 * (size_t)dim);
    for (int64_t i = 0; i < rows; ++i) {
        std::memcpy(rec.data(), &d, 4);
        std::memcpy(rec.data() + 4, in + i * 4 * (int64_t)dim, 4 * (size_t)dim);
        if (std::fwrite(rec.data(), rec.size(), 1, f) != 1) {
            std::fclose(f);
            set_error(std::string("write failed: ") + path);
            return VS_ERR_IO;
        }
    }
    std::fclose(f);
    return VS_OK;
}

// -------------------------------------------------------------------------------------------- npy
static bool npy_parse_header(std::ifstream& file, std::string& descr, bool& fortran, std::vector<int64_t>& shape) {
    char magic[6];
    file.read(magic, 6);
    if (!file || std::memcmp(magic, "\x93NUMPY", 6) != 0) return false;
    uint8_t ver[2];
    file.read(reinterpret_cast<char*>(ver), 2);
    uint32_t hlen = 0;
    if (ver[0] == 1) {
        uint16_t h16;
        file.read(reinterpret_cast<char*>(&h16), 2);
        hlen = h16;
    } else {
        file.read(reinterpret_cast<char*>(&hlen), 4);
    }
    if (!file || hlen > (1u << 20)) return false;
    std::string header(hlen, '\0');
    file.read(&header[0], hlen);
    if (!file) return false;
    auto find_val = [&](const char* key) -> size_t {
        size_t p = header.find(std::string("'") + key + "'");
        if (p == std::string::npos) return p;
        p = header.find(':', p);
        if (p == std::string::npos) return p;
        ++p;
        while (p < header.size() && header[p] == ' ') ++p;
        return p;
    };
    size_t p = find_val("descr");
    if (p == std::string::npos || header[p] != '\'') return false;
    size_t e = header.find('\'', p + 1);
    descr = header.substr(p + 1, e - p - 1);
    p = find_val("fortran_order");
    if (p == std::string::npos) return false;
    fortran = header.compare(p, 4, "True") == 0;
    p = find_val("shape");
    if (p == std::string::npos || header[p] != '(') return false;
    e = header.find(')', p);
    shape.clear();
    size_t i = p + 1;
    while (i < e) {
        while (i < e && !isdigit((unsigned char)header[i])) ++i;
        if (i >= e) break;
        int64_t v = 0;
        while (i < e && isdigit((unsigned char)header[i])) v = v * 10 + (header[i++] - '0');
        shape.push_back(v);
    }
    return true;
}

template <typename T>
static bool npy_read_t(const std::string& path, const char* want_a, const char* want_b, std::vector<T>& data,
                       std::vector<int64_t>& shape) {
    std::ifstream file(path, std::ios::binary);
    if (!file) {
        set_error("Cannot open " + path);
        return false;
    }
    std::string descr;
    bool fortran = false;
    if (!npy_parse_header(file, descr, fortran, shape)) {
        set_error("Bad .npy header: " + path);
        return false;
    }
    if (descr != want_a && descr != want_b) {
        set_error("Unexpected dtype '" + descr + "' in " + path);
        return false;
    }
    if (fortran && shape.size() > 1) {
        set_error("Fortran-ordered array not supported: " + path);
        return false;
    }
    // the payload must be in the file: a corrupt or hostile shape is an I/O error, not an allocation
    const std::streampos payload = file.tellg();
    file.seekg(0, std::ios::end);
    const int64_t remaining = (int64_t)(file.tellg() - payload);
    file.seekg(payload);
    int64_t total = 1;
    for (int64_t s : shape) {
        if (s < 0 || (s > 0 && total > remaining / s)) {  // total * s would exceed what the file can hold (or overflow)
            set_error("Truncated .npy payload (shape larger than the file): " + path);
            return false;
        }
        total *= s;
    }
    if (total > remaining / (int64_t)sizeof(T)) {
        set_error("Truncated .npy payload: " + path);
        return false;
    }
    try {
        data.resize((size_t)total);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory reading " + path);
        return false;
    }
    file.read(reinterpret_cast<char*>(data.data()), total * (int64_t)sizeof(T));
    if (file.gcount() != total * (int64_t)sizeof(T)) {
        set_error("Truncated .npy payload: " + path);
        return false;
    }
    return true;
}

bool npy_read_f32(const std::string& path, std::vector<float>& data, std::vector<int64_t>& shape) {
    return npy_read_t<float>(path, "<f4", "|f4", data, shape);
}
bool npy_read_i32(const std::string& path, std::vector<int32_t>& data, std::vector<int64_t>& shape) {
    return npy_read_t<int32_t>(path, "<i4", "|i4", data, shape);
}

bool npy_write(const std::string& path, const void* data, const char* descr, const std::vector<int64_t>& shape,
               size_t elem_size) {
    std::ostringstream hs;
    hs << "{'descr': '" << descr << "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); ++i) hs << shape[i] << (shape.size() == 1 || i + 1 < shape.size() ? "," : "") << (i + 1 < shape.size() ? " " : "");
    hs << "), }";
    std::string h = hs.str();
    const size_t pre = 10;  // magic + version + u16 len
    size_t total = pre + h.size() + 1;
    const size_t pad = (64 - total % 64) % 64;
    h.append(pad, ' ');
    h.push_back('\n');
    std::ofstream f(path, std::ios::binary);
    if (!f) {
        set_error("Cannot open " + path);
        return false;
    }
    f.write("\x93NUMPY\x01\x00", 8);
    const uint16_t hl = (uint16_t)h.size();
    f.write(reinterpret_cast<const char*>(&hl), 2);
    f.write(h.data(), (std::streamsize)h.size());
    int64_t n = 1;
    for (int64_t s : shape) n *= s;
    f.write(static_cast<const char*>(data), (std::streamsize)(n * (int64_t)elem_size));
    return f.good();
}

// ------------------------------------------------------------------------------------------- json
static bool json_find(const std::string& json, const std::string& key, size_t& pos) {
    pos = json.find("\"" + key + "\"");  // IVFIndex.cpp:14
    if (pos == std::string::npos) return false;
    pos = json.find(':', pos);
    if (pos == std::string::npos) return false;
    ++pos;
    while (pos < json.size() && isspace((unsigned char)json[pos])) ++pos;
    return pos < json.size();
}

bool ivf_config_read(const std::string& path, IvfConfig& cfg) {
    std::ifstream file(path);
    if (!file) {
        set_error("Cannot open config file: " + path);  // IVFIndex.cpp:184
        return false;
    }
    std::string json((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    size_t p;
    auto need_int = [&](const char* key, int64_t& v) {
        if (!json_find(json, key, p)) {
            set_error(std::string("Missing ") + key + " in config");  // IVFIndex.cpp:190-198
            return false;
        }
        v = std::strtoll(json.c_str() + p, nullptr, 10);
        return true;
    };
    if (!need_int("n_vectors", cfg.n_vectors) || !need_int("n_clusters", cfg.n_clusters) || !need_int("dim", cfg.dim))
        return false;
    if (json_find(json, "avg_cluster_size", p)) cfg.avg_cluster_size = std::strtod(json.c_str() + p, nullptr);
    if (json_find(json, "batch_size", p)) cfg.batch_size = std::strtoll(json.c_str() + p, nullptr, 10);
    if (json_find(json, "min_cluster_size", p)) cfg.min_cluster_size = std::strtoll(json.c_str() + p, nullptr, 10);
    if (json_find(json, "max_cluster_size", p)) cfg.max_cluster_size = std::strtoll(json.c_str() + p, nullptr, 10);
    cfg.reordered = false;  // default when absent (IVFIndex.cpp:202-203)
    if (json_find(json, "reordered", p)) cfg.reordered = json.compare(p, 4, "true") == 0;
    return true;
}

bool ivf_config_write(const std::string& path, const IvfConfig& c) {
    std::ofstream f(path);
    if (!f) {
        set_error("Cannot open " + path);
        return false;
    }
    f << "{\n  \"n_vectors\": " << c.n_vectors << ",\n  \"n_clusters\": " << c.n_clusters << ",\n  \"dim\": " << c.dim
      << ",\n  \"batch_size\": " << c.batch_size << ",\n  \"avg_cluster_size\": " << std::setprecision(17)
      << c.avg_cluster_size << ",\n  \"min_cluster_size\": " << c.min_cluster_size
      << ",\n  \"max_cluster_size\": " << c.max_cluster_size << ",\n  \"reordered\": " << (c.reordered ? "true" : "false")
      << "\n}\n";
    return f.good();
}

// ------------------------------------------------------------------------------------ select_topk
// cpu_baseline.cpp:127-153: k slots seeded with the first k rows; max_idx = first slot holding the
// max; a later row replaces that slot iff strictly smaller; final std::sort by dist, which for
// k <= 16 is libstdc++'s insertion sort, i.e. stable in slot order.
namespace {
struct Slot {
    float dist;
    int32_t idx;
};
template <typename RowOf>
void slots_run(RowOf row_of, const float* dist, int64_t n, int k, int32_t* out_ids, float* out_dists) {
    const int kk = (int)std::min<int64_t>(k, n);
    std::vector<Slot> buf((size_t)std::max(kk, 1));
    for (int i = 0; i < kk; ++i) buf[i] = {dist[i], row_of(i)};
    if (kk > 0) {
        int max_idx = 0;
        for (int i = 1; i < kk; ++i)
            if (buf[i].dist > buf[max_idx].dist) max_idx = i;
        for (int64_t j = kk; j < n; ++j) {
            if (dist[j] < buf[max_idx].dist) {
                buf[max_idx] = {dist[j], row_of(j)};
                max_idx = 0;
                for (int i = 1; i < kk; ++i)
                    if (buf[i].dist > buf[max_idx].dist) max_idx = i;
            }
        }
        std::stable_sort(buf.begin(), buf.begin() + kk, [](const Slot& a, const Slot& b) { return a.dist < b.dist; });
    }
    for (int i = 0; i < k; ++i) {
        out_ids[i] = i < kk ? buf[i].idx : -1;
        out_dists[i] = i < kk ? buf[i].dist : std::numeric_limits<float>::infinity();
    }
}
}  // namespace

void select_topk_slots_dense(const float* dist, int64_t n, int k, int32_t id_offset, int32_t* out_ids,
                             float* out_dists) {
    slots_run([&](int64_t j) { return (int32_t)(j + id_offset); }, dist, n, k, out_ids, out_dists);
}
void select_topk_slots_sparse(const int32_t* rows, const float* dist, int64_t m, int k, int32_t* out_ids,
                              float* out_dists) {
    slots_run([&](int64_t j) { return rows[j]; }, dist, m, k, out_ids, out_dists);
}

// --------------------------------------------------------------------------------- synthetic SIFT
// centers[4096][128] = |N(0, 40^2)|; x = clip(rint(center[u] + N(0, 18^2)), 0, 218).
// splitmix64-seeded xoshiro256**, Box-Muller.  Row i depends only on (seed, i).
namespace {
inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Xoshiro {
    uint64_t s[4];
    explicit Xoshiro(uint64_t seed) {
        for (auto& v : s) v = splitmix64(seed);
    }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    inline uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    inline double uniform() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }  // (0,1)
    inline void normal2(double& a, double& b) {
        const double u1 = uniform(), u2 = uniform();
        const double rad = std::sqrt(-2.0 * std::log(u1));
        a = rad * std::cos(6.283185307179586476925 * u2);
        b = rad * std::sin(6.283185307179586476925 * u2);
    }
};
constexpr int kCenters = 4096;
constexpr uint64_t kCenterSeed = 0x53494654ull;  // "SIFT": base and query files share the mixture
}  // namespace

static void synth_centers(int dim, std::vector<float>& centers, int n_centers = kCenters, double sigma = 40.0) {
    centers.resize((size_t)n_centers * dim);
    Xoshiro rng(kCenterSeed + (n_centers == kCenters ? 0 : (uint64_t)n_centers));
    for (size_t i = 0; i < centers.size(); i += 2) {
        double a, b;
        rng.normal2(a, b);
        centers[i] = (float)std::fabs(sigma * a);
        if (i + 1 < centers.size()) centers[i + 1] = (float)std::fabs(sigma * b);
    }
}

static void synth_rows(float* dst, int64_t row_begin, int64_t rows, int dim, uint64_t seed,
                       const std::vector<float>& centers, int n_centers = kCenters, double sigma = 18.0) {
    for (int64_t i = 0; i < rows; ++i) {
        const uint64_t gi = (uint64_t)(row_begin + i);
        Xoshiro rng(seed * 0xD1342543DE82EF95ull + gi * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull);
        const int u = (int)(rng.next() % (uint64_t)n_centers);
        const float* c = centers.data() + (size_t)u * dim;
        float* out = dst + i * dim;
        for (int t = 0; t < dim; t += 2) {
            double a, b;
            rng.normal2(a, b);
            double x0 = std::nearbyint(c[t] + sigma * a);
            out[t] = (float)std::min(218.0, std::max(0.0, x0));
            if (t + 1 < dim) {
                double x1 = std::nearbyint(c[t + 1] + sigma * b);
                out[t + 1] = (float)std::min(218.0, std::max(0.0, x1));
            }
        }
    }
}

}  // namespace vs

// ============================================================================ C ABI (host-only part)
extern "C" {

const char* vs_last_error(void) { return vs::get_error(); }

// create_ivf_model_reordered.py:92-94
int vs_ivf_clamp_nlist(int64_t n_vectors, int nlist) {
    if (nlist > n_vectors / 10) nlist = (int)std::max<int64_t>(16, n_vectors / 100);
    return nlist;
}

// create_ivf_model_reordered.py:108-128: rows sorted by cluster (stable: a counting sort, so rows of a cluster keep their
// original order -- numpy's default argsort there leaves that order unspecified), offsets = running sum of the sizes.
int vs_ivf_layout(const int32_t* assign, int64_t n_rows, int nlist, int32_t* cluster_offsets, int32_t* reorder_to_original) {
    if (!assign || !cluster_offsets || !reorder_to_original || n_rows < 0 || nlist <= 0 || n_rows > 0x7fffffffll) {
        vs::set_error("vs_ivf_layout: bad arguments");
        return VS_ERR_INVALID;
    }
    for (int c = 0; c <= nlist; ++c) cluster_offsets[c] = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        if (assign[i] < 0 || assign[i] >= nlist) {
            vs::set_error("vs_ivf_layout: cluster id out of range");
            return VS_ERR_INVALID;
        }
        ++cluster_offsets[assign[i] + 1];
    }
    for (int c = 0; c < nlist; ++c) cluster_offsets[c + 1] += cluster_offsets[c];
    try {
        std::vector<int32_t> cursor(cluster_offsets, cluster_offsets + nlist);
        for (int64_t i = 0; i < n_rows; ++i) reorder_to_original[cursor[(size_t)assign[i]]++] = (int32_t)i;
    } catch (const std::bad_alloc&) {
        vs::set_error("out of host memory");
        return VS_ERR_NOMEM;
    }
    return VS_OK;
}

int vs_fvecs_shape(const char* path, int64_t* rows, int* dim) { return vs::xvecs_scan(path, nullptr, 0, rows, dim); }
int vs_fvecs_read(const char* path, float* dst, int64_t cap, int64_t* rows, int* dim) {
    if (!dst) { vs::set_error("dst is NULL"); return VS_ERR_INVALID; }
    return vs::xvecs_scan(path, dst, cap, rows, dim);
}
int vs_ivecs_read(const char* path, int32_t* dst, int64_t cap, int64_t* rows, int* dim) {
    if (!dst) { vs::set_error("dst is NULL"); return VS_ERR_INVALID; }
    return vs::xvecs_scan(path, dst, cap, rows, dim);
}
int vs_fvecs_write(const char* path, const float* src, int64_t rows, int dim) { return vs::xvecs_write(path, src, rows, dim); }
int vs_ivecs_write(const char* path, const int32_t* src, int64_t rows, int dim) { return vs::xvecs_write(path, src, rows, dim); }

int vs_results_write(const char* path, const int32_t* ids, const float* dists, int64_t nq, int k, int style) {
    std::ofstream out(path);
    if (!out) {
        vs::set_error(std::string("Cannot open output file ") + path);  // cpu_baseline.cpp:158-160
        return VS_ERR_IO;
    }
    for (int64_t i = 0; i < nq; ++i) {
        out << "Query " << i << ":";
        for (int t = 0; t < k; ++t) {
            if (ids[i * k + t] < 0) continue;
            if (style == 1)
                out << " (" << ids[i * k + t] << ", " << std::fixed << std::setprecision(4) << dists[i * k + t] << ")";
            else
                out << " (" << ids[i * k + t] << ", " << dists[i * k + t] << ")";
        }
        out << "\n";
    }
    return out.good() ? VS_OK : VS_ERR_IO;
}

static int synth_impl(float* dst, int64_t row_begin, int64_t rows, int dim, uint64_t seed, int n_centers, double center_sigma, double row_sigma) {
    if (!dst || rows < 0 || dim <= 0 || n_centers < 1 || !(center_sigma >= 0) || !(row_sigma >= 0)) { vs::set_error("bad arguments"); return VS_ERR_INVALID; }
    std::vector<float> centers;
    vs::synth_centers(dim, centers, n_centers, center_sigma);
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt, 32u));
    if (rows < 4096) nt = 1;
    std::vector<std::thread> th;
    const int64_t per = (rows + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t b = (int64_t)t * per, e = std::min(rows, b + per);
        if (b >= e) break;
        th.emplace_back([=, &centers] { vs::synth_rows(dst + b * dim, row_begin + b, e - b, dim, seed, centers, n_centers, row_sigma); });
    }
    for (auto& x : th) x.join();
    return VS_OK;
}

int vs_synth_sift(float* dst, int64_t row_begin, int64_t rows, int dim, uint64_t seed) {
    return synth_impl(dst, row_begin, rows, dim, seed, vs::kCenters, 40.0, 18.0);
}

int vs_synth_mixture(float* dst, int64_t row_begin, int64_t rows, int dim, uint64_t seed, int n_centers, double center_sigma, double row_sigma) {
    return synth_impl(dst, row_begin, rows, dim, seed, n_centers, center_sigma, row_sigma);
}

int vs_select_topk_slots(const int32_t* rows, const float* dists, int64_t m, int k, int32_t* out_ids, float* out_dists) {
    if (!rows || !dists || !out_ids || !out_dists || k <= 0) { vs::set_error("bad arguments"); return VS_ERR_INVALID; }
    vs::select_topk_slots_sparse(rows, dists, m, k, out_ids, out_dists);
    return VS_OK;
}

}  // extern "C"
