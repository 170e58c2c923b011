// Does a chain of back-to-back DEPENDENT v_mfma_f32_16x16x32_f16 (same accumulator: the split-f16 kernels issue al*bh, ah*bl, ah*bh
// into acc[i][j] consecutively) run at the full MFMA rate?  24 MFMAs per iteration on 8 accumulators:
//   DEP 3: g0 g0 g0 g1 g1 g1 ...   (three consecutive MFMAs per accumulator — the kernels' order)
//   DEP 1: g0 g1 ... g7 g0 g1 ...   (the same work, dependent MFMAs 8 apart)
//   DEP 2: pairs of accumulators interleaved: g0 g1 g0 g1 g0 g1 g2 g3 ...
// zero and random data, 1 / 2 / 4 waves per SIMD.  hipcc -O3 --offload-arch=gfx950 tools/ubench_dep.hip -o tools/ubench_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MF(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
template <int DEP>
__global__ __launch_bounds__(256) void k(float* out, const u32x4* __restrict__ g, int iters, unsigned long long* clk) {
    const int tid = threadIdx.x;
    u32x4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = g[(blockIdx.x * 256 + tid) * 8 + i]; fb[i] = g[(blockIdx.x * 256 + tid) * 8 + 4 + i]; }
    f32x4 a4[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) a4[i][e] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < iters; ++t) {
        if constexpr (DEP == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { MF(a4[i], fa[i & 3], fb[i >> 1]); MF(a4[i], fa[(i + 1) & 3], fb[i >> 1]); MF(a4[i], fa[(i + 2) & 3], fb[i >> 1]); }
        } else if constexpr (DEP == 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) MF(a4[i], fa[(i + r) & 3], fb[i >> 1]);
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 3; ++r) { MF(a4[2 * p], fa[(2 * p + r) & 3], fb[p]); MF(a4[2 * p + 1], fa[(2 * p + 1 + r) & 3], fb[p]); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += a4[i][e];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int DEP>
void run(const char* name, int bpc, float* out, const u32x4* g, unsigned long long* clk) {
    const int blocks = 256 * bpc; int iters = 8000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<DEP>), dim3(blocks), dim3(256), 0, 0, out, g, 100, clk); hipDeviceSynchronize();
    float ms;
    hipEventRecord(e0); hipLaunchKernelGGL((k<DEP>), dim3(blocks), dim3(256), 0, 0, out, g, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 200.0 / ms);
    hipEventRecord(e0); hipLaunchKernelGGL((k<DEP>), dim3(blocks), dim3(256), 0, 0, out, g, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 2); hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0; for (int b = 0; b < blocks; ++b) { cs += h[b * 2]; rs += h[b * 2 + 1]; }
    const double n_mfma = 24.0 * iters;
    printf("%-44s %d waves/SIMD  %7.1f TF/s  clock %4.0f MHz  %.1f cycles per MFMA per SIMD\n", name, bpc, 16384.0 * n_mfma * 4.0 * blocks / (ms * 1e-3) / 1e12, cs / rs * 100.0,
           (cs / blocks) / (n_mfma * bpc));
}
int main() {
    const size_t n = 256 * 8 * 256 * 8;
    std::vector<unsigned> hr(n * 4), hz(n * 4, 0u);
    srand(1);
    for (size_t i = 0; i < n * 4; ++i) { unsigned lo = (rand() & 0x83ff) | 0x3800, hi = (rand() & 0x83ff) | 0x3800; hr[i] = lo | (hi << 16); }
    u32x4 *gr, *gz; float* out; unsigned long long* clk;
    hipMalloc(&gr, n * 16); hipMalloc(&gz, n * 16); hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 256 * 8 * 16);
    hipMemcpy(gr, hr.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(gz, hz.data(), n * 16, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<3>("3 consecutive per accumulator (kernel order), zeros", w, out, gz, clk);
        run<2>("pairs interleaved, zeros", w, out, gz, clk);
        run<1>("dependent MFMAs 8 apart, zeros", w, out, gz, clk);
        run<3>("3 consecutive per accumulator, random", w, out, gr, clk);
        run<1>("dependent MFMAs 8 apart, random", w, out, gr, clk);
    }
    return 0;
}
