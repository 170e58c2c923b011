// Microbenchmark 2: which workgroup / wave-tile shape sustains the most f16 MFMA with the conv kernel's
// instruction mix?  Per step and wave: TM*TN*6 MFMAs (3 products x 2 k16), (TM+TN)*4 ds_read_b128, WR
// ds_write_b128, one barrier.  LDS bytes sized so that the requested blocks/CU is what the hardware admits.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int TM, int TN, int WR, int LDS_KB, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_KB * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    u32x4 st[WR > 0 ? WR : 1];
    for (int i = 0; i < (WR > 0 ? WR : 1); ++i) st[i] = u32x4{0x3c003c00u + tid, 0x3c003c00u, 0x38003800u, 0x34003400u + i};
    const int rd = (lane & 31) * 64 + (((lane >> 5) ^ ((lane >> 2) & 3)) << 4);
    constexpr int HALF = LDS_KB * 512;
    for (int t = 0; t < iters; ++t) {
        const int buf = (t & 1) * HALF;
        f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[ks][i] = *reinterpret_cast<const f16x8*>(smem + buf + ((wave & 1) * TM + i) * 2048 + rd + ks * 32);
                al[ks][i] = *reinterpret_cast<const f16x8*>(smem + buf + 8192 + ((wave & 1) * TM + i) * 2048 + rd + ks * 32);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[ks][j] = *reinterpret_cast<const f16x8*>(smem + buf + 16384 + (j % 4) * 2048 + rd + ks * 32);
                bl[ks][j] = *reinterpret_cast<const f16x8*>(smem + buf + 24576 + (j % 4) * 2048 + rd + ks * 32);
            }
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) *reinterpret_cast<u32x4*>(smem + (HALF - buf) + (tid * 16 + i * 4096) % HALF) = st[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
                }
        __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int WAVES, int TM, int TN, int WR, int LDS_KB, int MINW>
void run(const char* name, float* out, int blocks_per_cu) {
    const int iters = 1500, blocks = 256 * blocks_per_cu * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WAVES, TM, TN, WR, LDS_KB, MINW>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, TM, TN, WR, LDS_KB, MINW>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * WAVES * iters * TM * TN * 6 * 32768.0;
    printf("%-64s %7.3f ms %7.1f TF  %5.1f%%\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 1024 * 4);
    run<4, 2, 2, 8, 64, 2>("A  4w 64x64 wave, 2 blk/CU, 24 mfma 16 rd 8 wr (current igemm)", out, 2);
    run<4, 2, 2, 4, 76, 2>("A' 4w 64x64 wave, 2 blk/CU, 24 mfma 16 rd 4 wr (current band)", out, 2);
    run<4, 3, 2, 4, 78, 2>("D  4w 96x64 wave, 2 blk/CU, 36 mfma 20 rd 4 wr (band 192x128)", out, 2);
    run<4, 2, 4, 8, 110, 1>("B  4w 64x128 wave, 1 blk/CU, 48 mfma 24 rd 8 wr (band 128x256)", out, 1);
    run<8, 2, 2, 2, 110, 2>("C  8w 64x64 wave, 1 blk/CU, 24 mfma 16 rd 2 wr (band 256x128)", out, 1);
    run<8, 2, 4, 4, 150, 2>("E  8w 64x128 wave, 1 blk/CU, 48 mfma 24 rd 4 wr (band 256x256)", out, 1);
    run<8, 4, 2, 4, 150, 2>("E' 8w 128x64 wave, 1 blk/CU, 48 mfma 24 rd 4 wr", out, 1);
    run<8, 1, 2, 2, 64, 4>("F  8w 32x64 wave, 2 blk/CU, 12 mfma 12 rd 2 wr (128x128, 8 waves)", out, 2);
    run<4, 1, 1, 2, 32, 4>("G  4w 32x32 wave, 4 blk/CU, 6 mfma 8 rd 2 wr (64x64)", out, 4);
    return 0;
}
