// vs_build.hip -- index builder kernels: compute_norms (cpu_baseline.cpp:95-125), Lloyd update and k-means++ seeding
// (create_ivf_model_reordered.py:88-118); see vs_kernels.h.
#include "vs_kernels.h"
#include "vs_dev.h"
#include <type_traits>
#include <algorithm>

namespace vs {

// ------------------------------------------------------------------------------------------------
// Row norms in the reference's order: 8 FMA lanes over v[8 i + j], then r0+r1+...+r7 left to
// right, then the scalar tail (cpu_baseline.cpp:95-114).  8 threads per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ v, int64_t rows, int dim,
                                                         float* __restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = gid >> 3;
    const int j = (int)(gid & 7);
    const bool ok = row < rows;
    const float* src = v + (ok ? row : 0) * dim;
    float acc = 0.f;
    const int d8 = dim & ~7;
    for (int i = 0; i < d8; i += 8) {
        const float x = src[i + j];
        acc = fmaf(x, x, acc);
    }
    const int lane = threadIdx.x & 63;
    const int b = lane & ~7;
    float sum = __shfl(acc, b);
#pragma unroll
    for (int u = 1; u < 8; ++u) sum = sum + __shfl(acc, b + u);
    for (int i = d8; i < dim; ++i) sum = fmaf(src[i], src[i], sum);
    if (ok && j == 0) out[row] = sum;
}

hipError_t launch_row_sqnorm(const float* v, int64_t rows, int dim, float* out, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t threads = rows * 8;
    const int grid = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3(grid), dim3(256), 0, s, v, rows, dim, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k-means update (index builder, create_ivf_model_reordered.py:96-105): cluster sums are accumulated
// in 44.20 fixed point with 64-bit integer atomics, so the result does not depend on the order in
// which rows arrive (float atomics would make the index differ from run to run).
// ------------------------------------------------------------------------------------------------
constexpr double kFix = 1048576.0;  // 2^20

__global__ __launch_bounds__(256) void kmeans_accum_kernel(const float* __restrict__ x, const int32_t* __restrict__ assign,
                                                           int64_t rows, unsigned long long* __restrict__ acc,
                                                           int32_t* __restrict__ counts) {
    // 128 threads per row, 2 rows per workgroup pass
    const int t = threadIdx.x & 127;
    for (int64_t row = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 7); row < rows; row += (int64_t)gridDim.x * 2) {
        const int c = assign[row];
        if (c < 0) continue;
        const long long v = __double2ll_rn((double)x[row * kDim + t] * kFix);
        atomicAdd(acc + (int64_t)c * kDim + t, (unsigned long long)v);
        if (t == 0) atomicAdd(counts + c, 1);
    }
}

__global__ __launch_bounds__(128) void kmeans_finalize_kernel(float* __restrict__ cents, const unsigned long long* __restrict__ acc,
                                                              const int32_t* __restrict__ counts, double* __restrict__ shift) {
    const int c = blockIdx.x, t = threadIdx.x;
    const int n = counts[c];
    float delta2 = 0.f;
    if (n > 0) {
        const double mean = (double)(long long)acc[(int64_t)c * kDim + t] / kFix / (double)n;
        const float nv = (float)mean;
        const float ov = cents[c * kDim + t];
        cents[c * kDim + t] = nv;
        delta2 = (nv - ov) * (nv - ov);
    }  // an empty cluster keeps its centroid
    __shared__ float red[128];
    red[t] = delta2;
    __syncthreads();
    for (int sft = 64; sft > 0; sft >>= 1) {
        if (t < sft) red[t] += red[t + sft];
        __syncthreads();
    }
    if (t == 0) shift[c] = (double)red[0];
}

hipError_t launch_kmeans_update(const float* x, const int32_t* assign, int64_t rows, int nlist, float* cents,
                                unsigned long long* acc, int32_t* counts, double* shift, hipStream_t s) {
    hipError_t e = hipMemsetAsync(acc, 0, (size_t)nlist * kDim * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(counts, 0, (size_t)nlist * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kmeans_accum_kernel, dim3(4096), dim3(256), 0, s, x, assign, rows, acc, counts);
    hipLaunchKernelGGL(kmeans_finalize_kernel, dim3(nlist), dim3(128), 0, s, cents, acc, counts, shift);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k-means++ seeding.  kpp_update_kernel: one workgroup per kKppBlockRows rows, 8 lanes per row (as in the IVF
// scans); kpp_pick_kernel: one workgroup finds the block, then the row, where the running sum passes u * total.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kpp_update_kernel(const float* __restrict__ x, const float* __restrict__ xnorm, int64_t rows,
                                                         const float* __restrict__ centre, float* __restrict__ d2,
                                                         double* __restrict__ block_sums) {
    __shared__ double wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rr = lane >> 3, s8 = lane & 7;
    f32x4 cf[4];
    float cn = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        cf[m] = *reinterpret_cast<const f32x4*>(centre + 4 * (s8 + 8 * m));
#pragma unroll
        for (int i = 0; i < 4; ++i) cn = fmaf(cf[m][i], cf[m][i], cn);
    }
    cn = dpp_add_xor1(cn);
    cn = dpp_add_xor2(cn);
    cn = dpp_add_half_mirror(cn);
    const int64_t row_begin = (int64_t)blockIdx.x * kKppBlockRows;
    double acc = 0.0;
    for (int r0 = wave * 8; r0 < kKppBlockRows; r0 += 32) {
        const int64_t row = row_begin + r0 + rr;
        const bool ok = row < rows;
        const float* src = x + (ok ? row : 0) * kDim + 4 * s8;
        float dot = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 32 * m);
#pragma unroll
            for (int i = 0; i < 4; ++i) dot = fmaf(v[i], cf[m][i], dot);
        }
        dot = dpp_add_xor1(dot);
        dot = dpp_add_xor2(dot);
        dot = dpp_add_half_mirror(dot);
        if (ok && s8 == 0) {
            const float d = fmaxf(fmaf(-2.0f, dot, xnorm[row] + cn), 0.f);
            const float nd = fminf(d2[row], d);
            d2[row] = nd;
            acc += (double)nd;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (tid == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void kpp_pick_kernel(const float* __restrict__ x, int64_t rows, const float* __restrict__ d2,
                                                        const double* __restrict__ block_sums, int n_blocks, double u,
                                                        float* __restrict__ out_centre) {
    __shared__ double s_part[1024];
    __shared__ double s_target, s_before;
    __shared__ int s_block;
    __shared__ long long s_row;
    const int tid = threadIdx.x;
    // total and the block where the running sum passes the target: block sums through LDS, 1024 at a time
    if (tid == 0) {
        s_before = 0.0;
        s_block = -1;
    }
    __syncthreads();
    double total = 0.0;
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
        s_part[tid] = b0 + tid < n_blocks ? block_sums[b0 + tid] : 0.0;
        __syncthreads();
        for (int t = 0; t < 1024 && b0 + t < n_blocks; ++t) total += s_part[t];  // every thread: the same order, the same sum
        __syncthreads();
    }
    const double target = u * total;
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
        s_part[tid] = b0 + tid < n_blocks ? block_sums[b0 + tid] : 0.0;
        __syncthreads();
        if (tid == 0 && s_block < 0) {
            double run = s_before;
            for (int t = 0; t < 1024 && b0 + t < n_blocks; ++t) {
                if (run + s_part[t] > target) {
                    s_block = b0 + t;
                    break;
                }
                run += s_part[t];
            }
            s_before = run;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (s_block < 0) s_block = n_blocks - 1;
        s_target = target;
        s_row = -1;
    }
    __syncthreads();
    const int64_t row = (int64_t)s_block * kKppBlockRows + tid;  // kKppBlockRows == blockDim.x
    s_part[tid] = row < rows ? (double)d2[row] : 0.0;
    __syncthreads();
    if (tid == 0) {
        double run = s_before;
        long long pick = -1;
        const int64_t last = min<int64_t>(rows, ((int64_t)s_block + 1) * kKppBlockRows) - 1;
        for (int t = 0; t < 1024; ++t) {
            run += s_part[t];
            if (run > s_target && s_part[t] > 0.0) {
                pick = (long long)s_block * kKppBlockRows + t;
                break;
            }
        }
        if (pick < 0) {  // rounding at the very end of the range (or an all-zero block): last row with d2 > 0, else the last row
            pick = last;
            for (int t = 1023; t >= 0; --t)
                if (s_part[t] > 0.0) {
                    pick = (long long)s_block * kKppBlockRows + t;
                    break;
                }
        }
        s_row = pick;
    }
    __syncthreads();
    if (tid < kDim) out_centre[tid] = x[s_row * kDim + tid];
}

hipError_t launch_kpp_step(const float* x, const float* xnorm, int64_t rows, float* cents, int c, float* d2, double* block_sums,
                           int n_blocks, double u, hipStream_t s) {
    hipLaunchKernelGGL(kpp_update_kernel, dim3(n_blocks), dim3(256), 0, s, x, xnorm, rows, cents + (size_t)(c - 1) * kDim, d2, block_sums);
    hipLaunchKernelGGL(kpp_pick_kernel, dim3(1), dim3(1024), 0, s, x, rows, d2, block_sums, n_blocks, u, cents + (size_t)c * kDim);
    return hipGetLastError();
}

}  // namespace vs
