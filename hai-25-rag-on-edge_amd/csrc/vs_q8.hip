// vs_q8.hip -- the UFIXED_POINT_8 score path of the reference's device runner (second half).
//
// What the reference's QnnRunner does around its NPU graph (qidk_bruteforce/android/app/main/jni):
//   QnnRunner.cpp:13-55    quantize_buffer_neon: q8 = sat_u8(trunc(x * (1 / input_scale) + 0.5)), offset 0
//   QnnRunner.cpp:490-521  encodings: input scale 0.6627451, output scale 1013.4312, both offsets 0
//   QnnRunner.cpp:608-645  executeBatchRaw: quantise the [B x d] batch, run the graph, leave the raw uint8 [B x N]
//                          score matrix in the runner's output buffer (getRawOutputBuffer, QnnRunner.h:37)
//   main.cpp:30-57         find_top_k_int8: k largest uint8 scores per query
//   main.cpp:244-246       printed score = uint8 score * output_scale
// The graph itself (database as uint8 weights with a per-tensor scale / offset, integer accumulation, requantisation
// of the accumulator to the output encoding) lives in QNN's closed converter and HTP runtime; it is restated here with
// the reference's own rounding rule: score8 = sat_u8(trunc(ip * (in_scale * w_scale / out_scale) + 0.5)) with
// ip = sum_t q8[t] * (w8[t] + w_offset), exact in int32.  Parity of this half is therefore unpinned.
//
// Kernels (gfx950):
//   q8_quantize_queries_kernel  one wave per query: the quantiser above, bytes stored as (q8 - 128) for the signed MFMA
//   q8_scores_kernel            a wave takes 64 database rows at a time, no LDS: A operands (rows) and B operands
//                               (queries) are 16-byte global loads in v_mfma_i32_16x16x64_i8 fragment order,
//                               the row -> MFMA-row assignment is chosen so that a lane ends up with 16 CONSECUTIVE rows
//                               of one query: one 16-byte store per lane and column block.  HBM bound: N*128 bytes in,
//                               B*N bytes out per batch.
//   q8_topk_chunk_kernel        per (16384-row chunk, query): a lower bound of the cut from the 256 per-thread maxima,
//                               the few entries above it ranked in LDS; long runs of equal scores: bisection over the
//                               byte value with SWAR compares, lowest-numbered entries at the cut first
//   q8_topk_final_kernel        per query: k rounds of "largest key below the previous one" over the chunks' candidates
// Order of equal scores: ascending row number (the reference's order among equal uint8 scores is whatever its C++
// library's heap leaves, main.cpp:36-57).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/vsearch.h"
#include "vs_host.h"

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

constexpr int kDim = 128;
constexpr int kBatch = 32;
constexpr int kGroup = 32;          // batches whose queries one quantiser launch prepares
constexpr int kGroupRows = 64;      // rows a wave scores per step
constexpr int kChunkRows = 16384;   // rows per top-k chunk (256 threads x 64 bytes)
constexpr int kCand = 16;           // candidate slots per (chunk, query); k <= 16
constexpr int kIpBias = 128 * 128 * kDim;  // sum over t of 128 * 128

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// sat_u8(trunc(v)) for a value that already carries its +0.5: NaN and negatives -> 0 (QnnRunner.cpp:52-53)
__device__ __forceinline__ unsigned q8_sat(float v) {
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    return (unsigned)(int)v;
}

__global__ __launch_bounds__(64) void q8_quantize_queries_kernel(const float* __restrict__ q, int B, float inv_scale, int w_off,
                                                                 int8_t* __restrict__ q8, int32_t* __restrict__ cq) {
    const int nb = blockIdx.x >> 5, b = blockIdx.x & 31, l = threadIdx.x;  // batch, row of the batch
    q8 += (size_t)nb * kBatch * kDim;
    cq += nb * kBatch;
    unsigned u0 = 0, u1 = 0;
    if (b < B) {  // rows past B are the zero padding of main.cpp:206-211
        const float2 x = *reinterpret_cast<const float2*>(q + ((size_t)nb * B + b) * kDim + 2 * l);
        u0 = q8_sat(__fadd_rn(__fmul_rn(x.x, inv_scale), 0.5f));  // vmulq_n_f32 then vaddq_f32: two roundings
        u1 = q8_sat(__fadd_rn(__fmul_rn(x.y, inv_scale), 0.5f));
    }
    char2 s;
    s.x = (signed char)((int)u0 - 128);
    s.y = (signed char)((int)u1 - 128);
    *reinterpret_cast<char2*>(q8 + (size_t)b * kDim + 2 * l) = s;
    int sum = (int)(u0 + u1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    // sum q8 * (w8 + off) = dot(q8 - 128, w8 - 128) + 128 sum(w8) + [128 sum(q8) - 128 * 128 * d + off * sum(q8)]
    if (l == 0) cq[b] = 128 * sum - kIpBias + w_off * sum;
}

// wq: [n_pad][128] bytes (w8 - 128), n_pad a multiple of 64; wterm: [n_pad] 128 * sum(w8)
template <int NQH>
__global__ __launch_bounds__(256) void q8_scores_kernel(const int8_t* __restrict__ wq, const int32_t* __restrict__ wterm,
                                                        const int8_t* __restrict__ q8, const int32_t* __restrict__ cq,
                                                        int64_t n_rows, int64_t n_groups, int B, float mult,
                                                        uint8_t* __restrict__ out, int64_t ld, int aligned) {
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (int64_t)gridDim.x * 4;
    i32x4 qb[NQH][2];
    int cqv[NQH];
#pragma unroll
    for (int h = 0; h < NQH; ++h) {
        const int8_t* qp = q8 + (size_t)(16 * h + c) * kDim + 16 * g;
        qb[h][0] = *reinterpret_cast<const i32x4*>(qp);
        qb[h][1] = *reinterpret_cast<const i32x4*>(qp + 64);
        cqv[h] = cq[16 * h + c];
    }
    // MFMA row m of tile t <- database row 16*(m/4) + 4*t + (m%4): the C fragment of lane group g (MFMA rows 4g..4g+3)
    // then holds rows 16g + 4t + i, i.e. over the four tiles the 16 consecutive rows 16g .. 16g+15
    const int arow = 16 * (c >> 2) + (c & 3);
    for (int64_t grp = wave0; grp < n_groups; grp += waves) {
        const int64_t row0 = grp * kGroupRows;
        const int8_t* ap = wq + (size_t)(row0 + arow) * kDim + 16 * g;
        i32x4 a0[4], a1[4], rw[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0[t] = *reinterpret_cast<const i32x4*>(ap + (size_t)(4 * t) * kDim);
            a1[t] = *reinterpret_cast<const i32x4*>(ap + (size_t)(4 * t) * kDim + 64);
            rw[t] = *reinterpret_cast<const i32x4*>(wterm + row0 + 16 * g + 4 * t);
        }
#pragma unroll
        for (int h = 0; h < NQH; ++h) {
            i32x4 word;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                i32x4 acc = (i32x4){0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[t], qb[h][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[t], qb[h][1], acc, 0, 0, 0);
                unsigned w = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ip = acc[i] + cqv[h] + rw[t][i];
                    w |= q8_sat(__fadd_rn(__fmul_rn((float)ip, mult), 0.5f)) << (8 * i);
                }
                word[t] = (int)w;
            }
            const int qi = 16 * h + c;
            const int64_t r = row0 + 16 * g;
            if (qi < B && r < n_rows) {
                uint8_t* dst = out + (size_t)qi * ld + r;
                if (aligned && r + 16 <= n_rows) {
                    *reinterpret_cast<i32x4*>(dst) = word;
                } else {
                    const int n = n_rows - r < 16 ? (int)(n_rows - r) : 16;
                    for (int j = 0; j < n; ++j) dst[j] = (uint8_t)(((unsigned)word[j >> 2] >> (8 * (j & 3))) & 0xffu);
                }
            }
        }
    }
}

// 128 * (number of bytes of x that are >= the byte replicated in m4), added to acc.  m7 = m4 & 0x7f7f7f7f, nm = ~m4.
__device__ __forceinline__ unsigned ge_bytes_acc(unsigned x, unsigned m4, unsigned m7, unsigned nm, unsigned acc) {
    const unsigned H = 0x80808080u;
    const unsigned t = (x | H) - m7;                    // per byte (x | 0x80) - (m & 0x7f): no borrow crosses bytes
    const unsigned ge = ((x & nm) | (~(x ^ m4) & t)) & H;  // bit 7 of a byte: x >= m (unsigned)
    return __builtin_amdgcn_udot4(ge, 0x01010101u, acc, false);
}

// bytes of the thread's 16 words that are >= mid (mid in 1..255)
__device__ __forceinline__ int count_ge(const unsigned (&w)[16], int mid) {
    const unsigned m4 = 0x01010101u * (unsigned)mid, m7 = m4 & 0x7f7f7f7fu, nm = ~m4;
    unsigned a = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) a = ge_bytes_acc(w[j], m4, m7, nm, a);
    return (int)(a >> 7);
}

__device__ __forceinline__ int block_sum_256(int v, int* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();  // sh may still be read from the previous use
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// one workgroup: the k largest of n keys
__device__ void q8_topk_merge(const u64* src, int n, int k, int32_t id_offset, int32_t* ids, uint8_t* out_scores) {
    __shared__ u64 shm[4];
    const int tid = threadIdx.x;
    constexpr int kOwn = 8;  // keys a thread keeps in registers (2048 per query = 2 M rows); more: re-read per round
    u64 own[kOwn];
#pragma unroll
    for (int j = 0; j < kOwn; ++j)
        own[j] = tid + 256 * j < n ? src[tid + 256 * j] : 0;
    u64 prev = ~0ull;
    for (int r = 0; r < k; ++r) {
        u64 best = 0;
#pragma unroll
        for (int j = 0; j < kOwn; ++j)
            if (own[j] < prev && own[j] > best) best = own[j];
        for (int i = tid + 256 * kOwn; i < n; i += 256) {
            const u64 v = src[i];
            if (v < prev && v > best) best = v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 other = __shfl_xor(best, o);
            best = other > best ? other : best;
        }
        __syncthreads();
        if ((tid & 63) == 0) shm[tid >> 6] = best;
        __syncthreads();
        best = max(max(shm[0], shm[1]), max(shm[2], shm[3]));
        if (tid == 0) {
            ids[r] = best ? (int32_t)(0xffffffffu - (unsigned)(best & 0xffffffffu)) + id_offset : -1;
            out_scores[r] = (uint8_t)(best >> 32);
        }
        prev = best;  // 0 when nothing is left: the remaining rounds find nothing either
    }
}

// one workgroup per query.  (Merging in the chunk kernel's last workgroup per query instead was measured: the fences it
// needs -- L2 write-back and invalidate in 2000 workgroups -- made that kernel four times slower.)
__global__ __launch_bounds__(256) void q8_topk_final_kernel(const u64* __restrict__ cand, int n_chunks, int k, int32_t id_offset,
                                                            int32_t* __restrict__ ids, uint8_t* __restrict__ out_scores) {
    const int b = blockIdx.x;
    q8_topk_merge(cand + (size_t)b * n_chunks * kCand, n_chunks * kCand, k, id_offset, ids + (size_t)b * k, out_scores + (size_t)b * k);
}

// bit 7 of every byte of x that is >= the byte replicated in m4
__device__ __forceinline__ unsigned ge_mask(unsigned x, unsigned m4) {
    const unsigned H = 0x80808080u;
    const unsigned t = (x | H) - (m4 & 0x7f7f7f7fu);
    return ((x & ~m4) | (~(x ^ m4) & t)) & H;
}

// scores: [B][ld] bytes (ld a multiple of 64, buffer 16-byte aligned); cand: [B][n_chunks][kCand] keys
// key = score << 32 | (0xffffffff - row): larger key = better (score descending, then row ascending); 0 = empty
//
// A thread holds 64 consecutive scores (16 words).  Fast path: L = the n_take-th largest of the 256 per-thread maxima is
// a lower bound of the cut (n_take different entries are >= L); when at most kListCap entries are >= L they are
// gathered in LDS and ranked against each other.  Otherwise (long runs of equal scores: saturated or coarse encodings)
// the cut is found by bisection over the byte value between L and the chunk maximum, and the entries above it plus the
// lowest-numbered entries equal to it are taken.
constexpr int kListCap = 256;
__global__ __launch_bounds__(256) void q8_topk_chunk_kernel(const uint8_t* __restrict__ scores, int64_t ld, int64_t n_rows, int k,
                                                            int n_chunks, u64* __restrict__ cand) {
    __shared__ int sh[4];
    __shared__ int sh_scan[4];
    __shared__ unsigned smax[64];  // the 256 per-thread maxima, one byte each
    __shared__ unsigned lst[kListCap];
    __shared__ u64 list[kCand];
    __shared__ int s_cnt, s_cntL;
    const int chunk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t r0 = (int64_t)chunk * kChunkRows + (int64_t)tid * 64;
    const int64_t left = n_rows - r0;
    const int nvalid = left <= 0 ? 0 : (left < 64 ? (int)left : 64);
    unsigned w[16];
    {
        const uint8_t* src = scores + (size_t)b * ld + r0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (16 * j < nvalid) v = *reinterpret_cast<const uint4*>(src + 16 * j);  // ld is padded to 64: whole pieces exist
            w[4 * j] = v.x, w[4 * j + 1] = v.y, w[4 * j + 2] = v.z, w[4 * j + 3] = v.w;
        }
        // bytes past the last row count as score 0 with the highest row numbers
        if (nvalid < 64) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rem = nvalid - 4 * j;
                if (rem < 4) w[j] = rem <= 0 ? 0u : (w[j] & ((1u << (8 * rem)) - 1u));
            }
        }
    }
    if (tid < kCand) list[tid] = 0;
    if (tid == 0) s_cnt = 0, s_cntL = 0;
    const int64_t chunk_left = n_rows - (int64_t)chunk * kChunkRows;
    const int chunk_rows = chunk_left < kChunkRows ? (int)chunk_left : kChunkRows;  // >= 1: the grid covers n_rows
    const int n_take = k < chunk_rows ? k : chunk_rows;
    // per-thread maximum -> LDS
    unsigned my_max;
    {
        u16x2 e = (u16x2){0, 0}, o = (u16x2){0, 0};  // even / odd bytes as 16-bit fields (v_pk_max_u16)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            e = __builtin_elementwise_max(e, __builtin_bit_cast(u16x2, w[j] & 0x00ff00ffu));
            o = __builtin_elementwise_max(o, __builtin_bit_cast(u16x2, (w[j] >> 8) & 0x00ff00ffu));
        }
        e = __builtin_elementwise_max(e, o);
        my_max = max((unsigned)e[0], (unsigned)e[1]);
        reinterpret_cast<uint8_t*>(smax)[tid] = (uint8_t)my_max;
    }
    __syncthreads();
    // every wave finds L from the 256 maxima on its own (lane l looks at word l = the maxima of threads 4l .. 4l+3)
    int L;
    {
        const unsigned mw = smax[tid & 63];
        const int m0 = (int)(mw & 0xffu), m1 = (int)((mw >> 8) & 0xffu), m2 = (int)((mw >> 16) & 0xffu), m3 = (int)(mw >> 24);
        int lo = 0, hi = 256;  // the number of maxima >= lo is >= n_take (256 of them are >= 0), fewer are >= hi
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            const int c = __builtin_popcountll(__ballot(m0 >= mid)) + __builtin_popcountll(__ballot(m1 >= mid)) +
                          __builtin_popcountll(__ballot(m2 >= mid)) + __builtin_popcountll(__ballot(m3 >= mid));
            if (c >= n_take) lo = mid;
            else hi = mid;
        }
        L = lo;
    }
    int c_lo = L > 0 ? (my_max >= (unsigned)L ? count_ge(w, L) : 0) : nvalid;  // this thread's entries >= L
    if (c_lo) atomicAdd(&s_cntL, c_lo);  // few threads hold such entries
    __syncthreads();
    const int cnt_L = s_cntL;  // >= n_take
    if (L > 0 && cnt_L <= kListCap) {
        // gather (score, row) of every entry >= L: key = score << 16 | (16383 - row in chunk)
        if (my_max >= (unsigned)L) {
            const unsigned m4 = 0x01010101u * (unsigned)L;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                unsigned g = ge_mask(w[j], m4);
                while (g) {
                    const int i = (__builtin_ctz(g) >> 3);
                    g &= g - 1;
                    const unsigned v = (w[j] >> (8 * i)) & 0xffu;
                    const int pos = atomicAdd(&s_cnt, 1);
                    lst[pos] = (v << 16) | (unsigned)(16383 - (tid * 64 + 4 * j + i));
                }
            }
        }
        __syncthreads();
        if (tid < cnt_L) {
            const unsigned key = lst[tid];
            int rank = 0;
            for (int j = 0; j < cnt_L; ++j) rank += lst[j] > key;
            if (rank < n_take)
                list[rank] = ((u64)(key >> 16) << 32) | (u64)(0xffffffffu - (unsigned)((int64_t)chunk * kChunkRows + (16383 - (int)(key & 0xffffu))));
        }
    } else {
        // t = the largest byte value with at least n_take entries >= t.
        // Invariant: cnt(lo) >= n_take > cnt(hi); c_lo / c_hi are this thread's own counts at lo / hi.
        int lo = L, hi = 256, cnt_hi = 0, c_hi = 0;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            const int c = count_ge(w, mid);
            const int cnt = block_sum_256(c, sh);
            if (cnt >= n_take) lo = mid, c_lo = c;
            else hi = mid, cnt_hi = cnt, c_hi = c;
        }
        const int t = lo;
        const int need_eq = n_take - cnt_hi;  // entries equal to t still to take, lowest rows first
        // entries equal to t per thread (valid ones: c_lo at t = 0 is nvalid) -> exclusive scan over the block (row order)
        const int c_eq = c_lo - c_hi;
        int incl = c_eq;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += up;
        }
        if ((tid & 63) == 63) sh_scan[tid >> 6] = incl;
        __syncthreads();
        int base = incl - c_eq;
        for (int wv = 0; wv < (tid >> 6); ++wv) base += sh_scan[wv];
        // only threads that hold an entry above the cut, or one of the first need_eq entries at the cut
        if (c_hi > 0 || (c_eq > 0 && base < need_eq)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = (int)((w[j] >> (8 * i)) & 0xffu);
                    bool take = v > t;
                    if (v == t && 4 * j + i < nvalid && base < need_eq) take = true, ++base;
                    if (take) {
                        const int pos = atomicAdd(&s_cnt, 1);
                        if (pos < kCand) list[pos] = ((u64)v << 32) | (u64)(0xffffffffu - (unsigned)(r0 + 4 * j + i));
                    }
                }
            }
        }
    }
    __syncthreads();
    if (tid < kCand) cand[((size_t)b * n_chunks + chunk) * kCand + tid] = list[tid];
}

}  // namespace

struct vs_q8 {
    int device = 0;
    int64_t n_rows = 0, n_pad = 0;
    int32_t id_offset = 0;
    vs_q8_encodings enc{};
    float inv_in = 0, mult = 0;
    int num_cus = 256;
    int8_t* d_wq = nullptr;      // [n_pad][128]
    int32_t* d_wterm = nullptr;  // [n_pad]
    int8_t* d_q8 = nullptr;      // [kGroup][32][128] quantised queries of up to kGroup batches
    int32_t* d_cq = nullptr;     // [kGroup][32]
    float* d_q = nullptr;        // [32][128] staging of host queries
    uint8_t* d_scores = nullptr; // [32][n_pad]  the runner's output buffer (QnnRunner.cpp:322-323)
    u64* d_cand = nullptr;       // [32][n_chunks][kCand]
    int32_t* d_ids = nullptr;    // [32][16]
    uint8_t* d_top = nullptr;    // [32][16]
    int n_chunks = 0;
    hipStream_t stream = nullptr;
};

namespace {

void q8_free(vs_q8* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    void* p[] = {h->d_wq, h->d_wterm, h->d_q8, h->d_cq, h->d_q, h->d_scores, h->d_cand, h->d_ids, h->d_top};
    for (void* x : p)
        if (x) (void)hipFree(x);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// the reference's quantiser on the host (database side, once): QnnRunner.cpp:50-54, then the weight offset
inline uint8_t quant_host(float x, float inv_scale, int w_off) {
    volatile float p = x * inv_scale;  // two roundings, as in the NEON body (no fused multiply-add)
    float v = p + 0.5f;
    v = std::fmin(std::fmax(v, 0.0f), 255.0f);  // NaN -> 0
    int q = (int)v - w_off;
    return (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
}

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

int q8_create_impl(const float* base_host, int64_t n_rows, int dim, const vs_q8_encodings* enc_in, int device, int64_t id_offset,
                   vs_q8** out) {
    if (!out || !base_host || n_rows <= 0) {
        set_error("vs_q8_create: bad arguments");
        return VS_ERR_INVALID;
    }
    if (dim != kDim) {
        set_error("only dim == 128 is compiled in");
        return VS_ERR_UNSUPPORTED;
    }
    if (n_rows + id_offset > 0x7fffffffLL || id_offset < 0) {
        set_error("ids must fit int32");
        return VS_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return VS_ERR_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_error("device index out of range");
        return VS_ERR_INVALID;
    }
    vs_q8_encodings enc;
    if (enc_in) {
        enc = *enc_in;
    } else {  // the runner's hard-coded I/O encodings (QnnRunner.cpp:490-521); weights: min-max over the database
        enc.input_scale = 0.6627451181411743f;
        enc.output_scale = 1013.4312133789062500f;
        float mx = 0.0f;
        for (int64_t i = 0; i < n_rows * kDim; ++i) mx = std::fmax(mx, base_host[i]);
        enc.weight_scale = mx > 0.0f ? mx / 255.0f : 1.0f;
        enc.weight_offset = 0;
    }
    if (!(enc.input_scale > 0.0f) || !(enc.weight_scale > 0.0f) || !(enc.output_scale > 0.0f) || enc.weight_offset > 0 ||
        enc.weight_offset < -255) {
        set_error("vs_q8_create: scales must be positive, weight_offset in [-255, 0] (real = scale * (q + offset))");
        return VS_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(device));
    vs_q8* h = new (std::nothrow) vs_q8();
    if (!h) {
        set_error("out of host memory");
        return VS_ERR_NOMEM;
    }
    h->device = device;
    h->n_rows = n_rows;
    h->n_pad = (n_rows + kGroupRows - 1) / kGroupRows * kGroupRows;
    h->id_offset = (int32_t)id_offset;
    h->enc = enc;
    h->inv_in = 1.0f / enc.input_scale;                                 // QnnRunner.cpp:619
    h->mult = (enc.input_scale * enc.weight_scale) / enc.output_scale;  // accumulator unit -> output unit
    h->n_chunks = (int)((n_rows + kChunkRows - 1) / kChunkRows);
    auto fail = [&](int rc) {
        q8_free(h);
        return rc;
    };
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
    // database -> uint8 (stored minus 128) + 128 * row sums, in slabs of 1 M rows
    if (hipMalloc((void**)&h->d_wq, (size_t)h->n_pad * kDim) != hipSuccess || hipMalloc((void**)&h->d_wterm, (size_t)h->n_pad * 4) != hipSuccess ||
        hipMalloc((void**)&h->d_q8, (size_t)kGroup * kBatch * kDim) != hipSuccess || hipMalloc((void**)&h->d_cq, (size_t)kGroup * kBatch * 4) != hipSuccess ||
        hipMalloc((void**)&h->d_q, kBatch * kDim * 4) != hipSuccess || hipMalloc((void**)&h->d_scores, (size_t)kBatch * h->n_pad) != hipSuccess ||
        hipMalloc((void**)&h->d_cand, (size_t)kBatch * h->n_chunks * kCand * 8) != hipSuccess ||
        hipMalloc((void**)&h->d_ids, kBatch * kCand * 4) != hipSuccess || hipMalloc((void**)&h->d_top, kBatch * kCand) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("vs_q8_create: device allocation failed");
        return fail(VS_ERR_DEVICE);
    }
    const float inv_w = 1.0f / enc.weight_scale;
    const int64_t slab = 1 << 20;
    std::vector<int8_t> bytes((size_t)std::min(slab, h->n_pad) * kDim);
    std::vector<int32_t> term((size_t)std::min(slab, h->n_pad));
    for (int64_t r0 = 0; r0 < h->n_pad; r0 += slab) {
        const int64_t rows = std::min(slab, h->n_pad - r0);
        for (int64_t i = 0; i < rows; ++i) {
            int sum = 0;
            for (int t = 0; t < kDim; ++t) {
                const int q = r0 + i < n_rows ? (int)quant_host(base_host[(size_t)(r0 + i) * kDim + t], inv_w, enc.weight_offset) : 0;
                bytes[(size_t)i * kDim + t] = (int8_t)(q - 128);
                sum += q;
            }
            term[(size_t)i] = 128 * sum;
        }
        if (hipMemcpy(h->d_wq + (size_t)r0 * kDim, bytes.data(), (size_t)rows * kDim, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_wterm + r0, term.data(), (size_t)rows * 4, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("vs_q8_create: upload failed");
            return fail(VS_ERR_DEVICE);
        }
    }
    *out = h;
    return VS_OK;
}

int q8_quantize_enqueue(vs_q8* h, const float* q_dev, int n_batches, int B, hipStream_t s) {
    hipLaunchKernelGGL(q8_quantize_queries_kernel, dim3(n_batches * kBatch), dim3(64), 0, s, q_dev, B, h->inv_in, h->enc.weight_offset, h->d_q8,
                       h->d_cq);
    HIPCHK(hipGetLastError());
    return VS_OK;
}

// scores of the slot-th quantised batch
int q8_scores_enqueue(vs_q8* h, int slot, int B, uint8_t* scores_dev, int64_t ld, hipStream_t s) {
    const int64_t n_groups = h->n_pad / kGroupRows;
    // 4 workgroups = 16 waves per CU: what 120 registers allow
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_groups + 3) / 4, (int64_t)h->num_cus * 4));
    const int aligned = (ld % 16 == 0) && (reinterpret_cast<uintptr_t>(scores_dev) % 16 == 0);
    const int8_t* q8 = h->d_q8 + (size_t)slot * kBatch * kDim;
    const int32_t* cq = h->d_cq + slot * kBatch;
    if (B <= 16)
        hipLaunchKernelGGL(q8_scores_kernel<1>, dim3(grid), dim3(256), 0, s, h->d_wq, h->d_wterm, q8, cq, h->n_rows, n_groups, B, h->mult,
                           scores_dev, ld, aligned);
    else
        hipLaunchKernelGGL(q8_scores_kernel<2>, dim3(grid), dim3(256), 0, s, h->d_wq, h->d_wterm, q8, cq, h->n_rows, n_groups, B, h->mult,
                           scores_dev, ld, aligned);
    HIPCHK(hipGetLastError());
    return VS_OK;
}

int q8_execute_enqueue(vs_q8* h, const float* q_dev, int B, uint8_t* scores_dev, int64_t ld, hipStream_t s) {
    int rc = q8_quantize_enqueue(h, q_dev, 1, B, s);
    return rc ? rc : q8_scores_enqueue(h, 0, B, scores_dev, ld, s);
}

int q8_topk_enqueue(vs_q8* h, int B, int k, int32_t* ids_dev, uint8_t* top_dev, hipStream_t s) {
    hipLaunchKernelGGL(q8_topk_chunk_kernel, dim3(h->n_chunks, B), dim3(256), 0, s, h->d_scores, h->n_pad, h->n_rows, k, h->n_chunks, h->d_cand);
    hipLaunchKernelGGL(q8_topk_final_kernel, dim3(B), dim3(256), 0, s, h->d_cand, h->n_chunks, k, h->id_offset, ids_dev, top_dev);
    HIPCHK(hipGetLastError());
    return VS_OK;
}

}  // namespace

extern "C" {

int vs_q8_create(const float* base_host, int64_t n_rows, int dim, const vs_q8_encodings* enc, int device, int64_t id_offset,
                 vs_q8** out) {
    return guarded([&]() -> int { return q8_create_impl(base_host, n_rows, dim, enc, device, id_offset, out); });
}

void vs_q8_destroy(vs_q8* h) { q8_free(h); }

int64_t vs_q8_num_docs(const vs_q8* h) { return h ? h->n_rows : 0; }
int vs_q8_dim(const vs_q8* h) { return h ? kDim : 0; }
int vs_q8_batch(const vs_q8* h) { return h ? kBatch : 0; }
float vs_q8_output_scale(const vs_q8* h) { return h ? h->enc.output_scale : 0.0f; }
int vs_q8_get_encodings(const vs_q8* h, vs_q8_encodings* out) {
    if (!h || !out) {
        set_error("vs_q8_get_encodings: bad arguments");
        return VS_ERR_INVALID;
    }
    *out = h->enc;
    return VS_OK;
}

int vs_q8_execute_dev(vs_q8* h, const float* queries_dev, int B, uint8_t* scores_dev, int64_t ld, void* stream) {
    if (!h || !queries_dev || !scores_dev || B < 1 || B > kBatch || ld < h->n_rows) {
        set_error("vs_q8_execute_dev: bad arguments (1 <= B <= 32, ld >= rows)");
        return VS_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(h->device));
    return q8_execute_enqueue(h, queries_dev, B, scores_dev, ld, static_cast<hipStream_t>(stream));
}

int vs_q8_execute(vs_q8* h, const float* queries_host, int B, uint8_t* scores_host) {
    if (!h || !queries_host || !scores_host || B < 1 || B > kBatch) {
        set_error("vs_q8_execute: bad arguments (1 <= B <= 32)");
        return VS_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_q, queries_host, (size_t)B * kDim * 4, hipMemcpyHostToDevice, h->stream));
    int rc = q8_execute_enqueue(h, h->d_q, B, h->d_scores, h->n_pad, h->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpy2DAsync(scores_host, (size_t)h->n_rows, h->d_scores, (size_t)h->n_pad, (size_t)h->n_rows, (size_t)B, hipMemcpyDeviceToHost,
                            h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return VS_OK;
}

int vs_q8_search_dev(vs_q8* h, const float* queries_dev, int n_batches, int B, int k, int32_t* ids_dev, uint8_t* scores_dev, void* stream) {
    if (!h || !queries_dev || !ids_dev || !scores_dev || n_batches < 1 || B < 1 || B > kBatch || k < 1) {
        set_error("vs_q8_search_dev: bad arguments");
        return VS_ERR_INVALID;
    }
    if (k > kCand) {
        set_error("k too large for the compiled top-k kernels (k <= 16)");
        return VS_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int g0 = 0; g0 < n_batches; g0 += kGroup) {  // one quantiser launch per group of batches, three launches per batch
        const int gs = std::min(kGroup, n_batches - g0);
        int rc = q8_quantize_enqueue(h, queries_dev + (size_t)g0 * B * kDim, gs, B, s);
        if (rc) return rc;
        for (int j = 0; j < gs; ++j) {
            const size_t o = (size_t)(g0 + j) * B * k;
            if ((rc = q8_scores_enqueue(h, j, B, h->d_scores, h->n_pad, s)) || (rc = q8_topk_enqueue(h, B, k, ids_dev + o, scores_dev + o, s))) return rc;
        }
    }
    return VS_OK;
}

int vs_q8_search(vs_q8* h, const float* queries_host, int64_t nq, int k, int32_t* ids, uint8_t* scores) {
    if (!h || !queries_host || !ids || !scores || nq < 0 || k < 1) {
        set_error("vs_q8_search: bad arguments");
        return VS_ERR_INVALID;
    }
    if (k > kCand) {
        set_error("k too large for the compiled top-k kernels (k <= 16)");
        return VS_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(h->device));
    for (int64_t q0 = 0; q0 < nq; q0 += kBatch) {  // the harness loop of main.cpp:201-251
        const int B = (int)std::min<int64_t>(kBatch, nq - q0);
        HIPCHK(hipMemcpyAsync(h->d_q, queries_host + (size_t)q0 * kDim, (size_t)B * kDim * 4, hipMemcpyHostToDevice, h->stream));
        int rc = q8_execute_enqueue(h, h->d_q, B, h->d_scores, h->n_pad, h->stream);
        if (rc || (rc = q8_topk_enqueue(h, B, k, h->d_ids, h->d_top, h->stream))) return rc;
        HIPCHK(hipMemcpyAsync(ids + (size_t)q0 * k, h->d_ids, (size_t)B * k * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(scores + (size_t)q0 * k, h->d_top, (size_t)B * k, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return VS_OK;
}

}  // extern "C"
