// Microbenchmark: what bounds the split-f16 conv main loop?  Each variant runs the kernel's per-chunk
// instruction mix (24 x v_mfma_f32_32x32x16_f16 per wave per chunk, 16 ds_read_b128, 8 ds_write_b128,
// 1 barrier) with pieces switched off.  4 waves per block, 2 blocks per CU, 512 blocks.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_lds.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int READS, int WRITES, int BARRIER, int GLOADS>
__global__ __launch_bounds__(256, 2) void k(float* out, const u32x4* __restrict__ g, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8 fa[8], fb[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) { fa[i][e] = (_Float16)(0.001f * (lane + i + e)); fb[i][e] = (_Float16)(0.002f * (lane - i + e)); }
    u32x4 st[8];
    for (int i = 0; i < 8; ++i) st[i] = g[(blockIdx.x * 256 + tid) * 8 + i];
    const int rd_off = ((tid >> 6) & 1) * 8192 + (lane & 31) * 64 + ((((lane >> 5)) ^ ((lane >> 2) & 3)) << 4);
    const int wr_off = tid * 16;
    for (int t = 0; t < iters; ++t) {
        const int buf = (t & 1) * 32768;
        if (READS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                fa[i] = *reinterpret_cast<const f16x8*>(smem + buf + rd_off + (i & 3) * 2048 + (i >> 2) * 32);
                fb[i] = *reinterpret_cast<const f16x8*>(smem + buf + 16384 + rd_off + (i & 3) * 2048 + (i >> 2) * 32);
            }
        }
        if (WRITES) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(smem + (32768 - buf) + wr_off + i * 4096) = st[i];
        }
        if (GLOADS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) st[i] = g[((blockIdx.x * 256 + tid) * 8 + i + t * 4099) & 0xFFFFF];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[(r + i) & 7], fb[(r * 3 + i) & 7], acc[i], 0, 0, 0);
        if (BARRIER) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += (float)st[i][0];
    out[blockIdx.x * 256 + tid] = s;
}

template <int R, int W, int B, int G>
void run(const char* name, float* out, const u32x4* g, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<R, W, B, G>), dim3(blocks), dim3(256), 0, 0, out, g, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, W, B, G>), dim3(blocks), dim3(256), 0, 0, out, g, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 24 * 32768.0;
    printf("%-44s blocks %4d  %8.3f ms  %7.1f TFLOP/s (f16 mfma)  %5.1f%% of 2500\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
}

int main() {
    float* out; u32x4* g;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&g, (1 << 20) * 16 + 4096 * 256 * 8 * 16);
    hipMemset(g, 0x3c, (1 << 20) * 16);
    for (int blocks : {512, 1024}) {
        run<0, 0, 0, 0>("mfma only", out, g, blocks);
        run<1, 0, 0, 0>("mfma + 16 ds_read_b128", out, g, blocks);
        run<1, 1, 0, 0>("mfma + reads + 8 ds_write_b128", out, g, blocks);
        run<1, 1, 1, 0>("mfma + reads + writes + barrier", out, g, blocks);
        run<1, 1, 1, 1>("mfma + reads + writes + barrier + 8 gloads", out, g, blocks);
        run<0, 0, 1, 0>("mfma + barrier", out, g, blocks);
        run<0, 1, 1, 0>("mfma + writes + barrier", out, g, blocks);
    }
    return 0;
}
