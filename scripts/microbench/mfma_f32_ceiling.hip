// Microbenchmark behind DESIGN.md 6 ("what bounds the fp32 scan"): the rate of v_mfma_f32_16x16x4_f32 in the issue
// pattern of scan_f32s_kernel<2> -- 256 workgroups x 8 waves (2 per SIMD), 128 MFMAs per step in 4 accumulator chains
// with the B operands (128 registers) resident -- without any memory traffic, then with the step's vector epilogue
// appended behind the MFMAs (the kernel's order) or interleaved with the next step's MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_ceiling mfma_f32_ceiling.hip && ./mfma_f32_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMAs only.  1: + 16 fma / 16 compares per step behind the MFMAs.  2: the same work on the PREVIOUS step's
// accumulators, spread through the MFMA stream.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, const float* in) {
    const int lane = threadIdx.x & 63;
    f32x4 qf[4][8];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int c = 0; c < 8; ++c) qf[h][c] = (f32x4){in[lane + c], in[lane + 8 + c], in[lane + h], 1.f};
    f32x4 a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = (f32x4){in[lane + 16 + c], in[lane + 24 + c], in[lane + 1], 1.f};
    float tau[4] = {in[lane], in[lane + 1], in[lane + 2], in[lane + 3]};
    float cnt = 0.f;
    f32x4 prev[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        bool any = false;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 4; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[h][c][i], acc[h], 0, 0, 0);
            if (MODE == 2) {  // two (h, j) items of the previous step's epilogue per 16 MFMAs
#pragma unroll
                for (int e = 2 * c; e < 2 * c + 2; ++e) {
                    const float l2 = fmaf(-2.0f, prev[e >> 2][e & 3], tau[e >> 2] + a[0][e & 3]);
                    any = any || l2 < tau[e >> 2];
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float l2 = fmaf(-2.0f, acc[e >> 2][e & 3], tau[e >> 2] + a[0][e & 3]);
                any = any || l2 < tau[e >> 2];
            }
        }
        if (MODE == 0) any = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] < -1.f;
        if (__ballot(any)) cnt += 1.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) prev[h] = acc[h];
        a[0][0] += 1.f;
    }
    out[blockIdx.x * 512 + threadIdx.x] = cnt + prev[0][0] + prev[1][1] + prev[2][2] + prev[3][3];
}


// The tile loop of scan_f32s_kernel<2> without its epilogue: per step 9 LDS reads of A fragments from the wave's ring slot,
// the refill of that slot by 9 LDS-DMA instructions (8 KB of rows + 64 B of norms), 128 MFMAs.
// SRC 0: no refill (the ring keeps its first content).  1: refill from a 4 GB buffer (HBM, no reuse).
// 2: refill from a 2 MB buffer (L2 hits).
template <int SRC, int STAG = 0, int INTER = 0>
__global__ __launch_bounds__(512, 2) void kr(float* out, int iters, const float* in, const float* base, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    constexpr int kSlot = 8448;
    char* ring = smem + wave * 2 * kSlot;
    f32x4 qf[4][8];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int c = 0; c < 8; ++c) qf[h][c] = (f32x4){in[lane + c], in[lane + 8 + c], in[lane + h], 1.f};
    unsigned voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row_in = 2 * j + (lane >> 5);
        voff[j] = (unsigned)(row_in * 512 + 16 * ((lane & 31) ^ row_in));
    }
    auto issue_tile = [&](int tile, int slot) __attribute__((always_inline)) {
        char* dst = ring + slot * kSlot;
        const char* tb = reinterpret_cast<const char*>(base) + (size_t)tile * 8192;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + voff[j]),
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, SRC == 1 ? 2 : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dst + 8192), 4, 0, 0);
    };
    int tile = (int)blockIdx.x + wave * 256;
    issue_tile(tile % n_tiles, 0);
    issue_tile((tile + 2048) % n_tiles, 1);
    tile += 4096;
    f32x4 tot = {0, 0, 0, 0};
    if (STAG && wave >= 4) {  // the second wave of every SIMD starts STAG MFMAs late
        f32x4 t = {0, 0, 0, 0};
        for (int i = 0; i < STAG; ++i) t = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[0][0][0], qf[1][0][1], t, 0, 0, 0);
        tot += t;
    }
    for (int it = 0; it < iters; ++it) {
        const int sl = it & 1;
        if (SRC) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* src = ring + sl * kSlot;
        f32x4 a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = *reinterpret_cast<const f32x4*>(src + r * 512 + (((4 * c + g) ^ r) << 4));
        const f32x4 bn = *reinterpret_cast<const f32x4*>(src + 8192 + 16 * g);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (SRC) {
            issue_tile(tile % n_tiles, sl);
            tile += 2048;
        }
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 4; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[h][c][i], acc[h], 0, 0, 0);
            if (INTER) {  // one refill instruction per 14 MFMAs instead of all nine in front
                __builtin_amdgcn_sched_group_barrier(0x008, 14, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (c == 7) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
        }
        tot += acc[0] + acc[1] + acc[2] + acc[3] + bn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
}

template <int SRC, int STAG = 0, int INTER = 0>
void runr(const char* name, float* out, const float* in, const float* base, int n_tiles, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kr<SRC, STAG, INTER>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 8448);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kr<SRC, STAG, INTER>), dim3(256), dim3(512), 8 * 2 * 8448, 0, out, iters, in, base, n_tiles);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * iters * 128 * 2048;
        if (rep == 2)
            printf("%-44s %8.1f us  %6.1f TFLOP/s  (%.3f of 157.3)  %.2f us per step and wave, %.2f TB/s\n", name, ms * 1e3, flop / (ms * 1e-3) / 1e12,
                   flop / (ms * 1e-3) / 157.3e12, ms * 1e3 / iters, SRC ? 256.0 * 8 * iters * 8256 / (ms * 1e-3) / 1e12 : 0.0);
    }
}

// The ring shared by the workgroup: wave w fetches tile 8 j + w of group j into one half of the ring while
// all eight waves compute on the other half (each on all eight tiles, for its own pass); one barrier per group.
__global__ __launch_bounds__(512, 2) void krs(float* out, int groups, const float* in, const float* base, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    constexpr int kSlot = 8448;
    f32x4 qf[4][8];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int c = 0; c < 8; ++c) qf[h][c] = (f32x4){in[lane + c], in[lane + 8 + c], in[lane + h], 1.f};
    unsigned voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row_in = 2 * j + (lane >> 5);
        voff[j] = (unsigned)(row_in * 512 + 16 * ((lane & 31) ^ row_in));
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    auto issue_tile = [&](int tile, int half) __attribute__((always_inline)) {
        const unsigned dst = lds0 + (unsigned)((8 * half + wave) * kSlot);
        const char* tb = reinterpret_cast<const char*>(base) + (size_t)tile * 8192;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(dst), "v"(voff[j]), "s"(tb), "n"(j * 1024) : "memory", "scc");
        asm volatile("s_add_u32 m0, %0, 8192\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(dst), "v"((unsigned)lane * 4u), "s"(tb) : "memory", "scc");
    };
    int tile = (int)blockIdx.x * 8 + wave;
    issue_tile(tile % n_tiles, 0);
    tile += 2048;
    f32x4 tot = {0, 0, 0, 0};
    unsigned fa[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) fa[c] = (unsigned)(r * 512 + (((4 * c + g) ^ r) << 4));
    for (int grp = 0; grp < groups; ++grp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_tile(tile % n_tiles, (grp + 1) & 1);
        tile += 2048;
        const char* hb = smem + (grp & 1) * 8 * kSlot;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const char* src = hb + t * kSlot;
            f32x4 a[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] = *reinterpret_cast<const f32x4*>(src + fa[c]);
            const f32x4 bn = *reinterpret_cast<const f32x4*>(src + 8192 + 16 * g);
            f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 4; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], qf[h][c][i], acc[h], 0, 0, 0);
            tot += acc[0] + acc[1] + acc[2] + acc[3] + bn;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
}

void runs(const char* name, float* out, const float* in, const float* base, int n_tiles, int groups) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(krs), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8448);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(krs, dim3(256), dim3(512), 16 * 8448, 0, out, groups, in, base, n_tiles);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * groups * 8 * 128 * 2048;
        if (rep == 2)
            printf("%-44s %8.1f us  %6.1f TFLOP/s  (%.3f of 157.3)  %.2f us per step and wave, %.2f TB/s from HBM\n", name, ms * 1e3,
                   flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 157.3e12, ms * 1e3 / (groups * 8), 256.0 * 8 * groups * 8256 / (ms * 1e-3) / 1e12);
    }
}

template <int MODE>
void run(const char* name, float* out, const float* in, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, in);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * iters * 128 * 2048;  // 16 x 16 x 4 x 2 per MFMA and wave
        if (rep == 2) printf("%-44s %8.1f us  %6.1f TFLOP/s  (%.3f of 157.3)\n", name, ms * 1e3, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 157.3e12);
    }
}

int main() {
    float *out, *in;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&in, 4096);
    (void)hipMemset(in, 0, 4096);
    const int iters = 600;  // ~ the 2 ms of a 32-batch launch
    run<0>("MFMA only, 2 waves per SIMD", out, in, iters);
    run<1>("+ epilogue behind the MFMAs", out, in, iters);
    run<2>("+ epilogue of the previous step interleaved", out, in, iters);
    run<0>("MFMA only (again, clocks settled)", out, in, 4 * iters);
    float* base;
    const int n_tiles = 500000;  // 4 GB
    (void)hipMalloc(&base, (size_t)n_tiles * 8192 + 4096);
    (void)hipMemset(base, 0, (size_t)n_tiles * 8192 + 4096);
    runr<0>("ring: LDS reads + 128 MFMAs, no refill", out, in, base, n_tiles, iters);
    runr<2>("ring: + LDS-DMA refill from L2 (2 MB)", out, in, base, 256, iters);
    runr<1>("ring: + LDS-DMA refill from HBM (4 GB)", out, in, base, n_tiles, iters);
    runr<1, 0, 1>("ring + HBM refill spread through the MFMAs", out, in, base, n_tiles, iters);
    runr<1, 32>("ring + HBM refill, 2nd wave 32 MFMAs late", out, in, base, n_tiles, iters);
    runr<1, 64>("ring + HBM refill, 2nd wave 64 MFMAs late", out, in, base, n_tiles, iters);
    runr<1, 128>("ring + HBM refill, 2nd wave 128 MFMAs late", out, in, base, n_tiles, iters);
    runs("ring shared by the workgroup, 1 barrier / 8 tiles", out, in, base, n_tiles, iters / 8);
    return 0;
}
