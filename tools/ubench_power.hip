// Is the split-f16 conv power-limited?  Bare MFMA loops (operands in registers, no LDS, no memory) at 1, 2 and 4 waves per
// SIMD on every CU, on random / zero data, f16 vs bf16, 16x16x32 vs 32x32x16; an optional LDS-read mix (0.5 ds_read_b128
// per MFMA = the band kernel's ratio).  Reports TFLOP/s (wall, sustained ~0.2 s per case) and the in-kernel clock
// (s_memtime / s_memrealtime x 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_power.hip -o tools/ubench_power && tools/ubench_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// KIND 0: f16 16x16x32, 1: bf16 16x16x32, 2: f16 32x32x16;  LDSR: ds_read_b128 per 2 MFMAs (fragments re-read from LDS)
template <int KIND, int LDSR>
__global__ __launch_bounds__(256) void k(float* out, const u32x4* __restrict__ g, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
    const int tid = threadIdx.x;
    u32x4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = g[(blockIdx.x * 256 + tid) * 8 + i]; fb[i] = g[(blockIdx.x * 256 + tid) * 8 + 4 + i]; }
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(smem + (tid * 8 + i) * 16) = g[(blockIdx.x * 256 + tid) * 8 + i];
    __syncthreads();
    f32x4 a4[8]; f32x16 a16[2];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) a4[i][e] = 0.f;
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) a16[i][e] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < iters; ++t) {
        if (LDSR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(smem + (((tid + 256 * i + 64 * (t & 3)) & 2047) * 16));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                                          // in-place accumulate, pinned by asm (the builtin form made hipcc rotate the accumulator tuples through copies)
            if constexpr (KIND == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a4[i]) : "v"(fa[i & 3]), "v"(fb[(i >> 1) & 3]));
            else if constexpr (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[i]) : "v"(fa[i & 3]), "v"(fb[(i >> 1) & 3]));
            else if ((i & 1) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a16[(i >> 1) & 1]) : "v"(fa[i & 3]), "v"(fb[(i >> 1) & 3]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += a4[i][e];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) s += a16[i][e];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int KIND, int LDSR>
void run(const char* name, int blocks_per_cu, float* out, const u32x4* g, unsigned long long* clk) {
    const int blocks = 256 * blocks_per_cu;
    // flops per iteration per wave: 8 MFMAs x 16384 (KIND 2: 4 x 32768)
    const double flop_iter = 8.0 * 16384.0;
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, LDSR>), dim3(blocks), dim3(256), 0, 0, out, g, 100, clk);
    hipDeviceSynchronize();
    // calibrate to ~0.25 s
    hipEventRecord(e0); hipLaunchKernelGGL((k<KIND, LDSR>), dim3(blocks), dim3(256), 0, 0, out, g, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 250.0 / ms);
    hipEventRecord(e0); hipLaunchKernelGGL((k<KIND, LDSR>), dim3(blocks), dim3(256), 0, 0, out, g, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 2); hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0; for (int b = 0; b < blocks; ++b) { cs += h[b * 2]; rs += h[b * 2 + 1]; }
    const double tf = flop_iter * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;
    printf("%-34s %d waves/SIMD  %7.1f TF/s  clock %4.0f MHz  (%.0f ms)\n", name, blocks_per_cu, tf, cs / rs * 100.0, ms);
}

int main() {
    const size_t n = 256 * 8 * 256 * 8;
    std::vector<unsigned> hr(n * 4), hz(n * 4, 0u);
    srand(1);
    // random finite f16 / bf16 pairs: sign + exponent near 1.0 + random mantissa
    for (size_t i = 0; i < n * 4; ++i) {
        unsigned lo = (rand() & 0x83ff) | 0x3800, hi = (rand() & 0x83ff) | 0x3800;     // f16: exponent 14..15; as bf16: also finite, moderate
        hr[i] = lo | (hi << 16);
    }
    u32x4 *gr, *gz; float* out; unsigned long long* clk;
    hipMalloc(&gr, n * 16); hipMalloc(&gz, n * 16); hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 256 * 8 * 16);
    hipMemcpy(gr, hr.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(gz, hz.data(), n * 16, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<0, 0>("f16 16x16x32 random", w, out, gr, clk);
        run<0, 0>("f16 16x16x32 zeros", w, out, gz, clk);
        run<1, 0>("bf16 16x16x32 random", w, out, gr, clk);
        run<2, 0>("f16 32x32x16 random", w, out, gr, clk);
        run<0, 1>("f16 16x16x32 random + LDS 0.5/MFMA", w, out, gr, clk);
        run<1, 1>("bf16 16x16x32 random + LDS 0.5/MFMA", w, out, gr, clk);
    }
    return 0;
}
