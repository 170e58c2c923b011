// Stem convolution (first layer: 3x3, Cin = 3, pad 1) on the exact-fp32 MFMA, reading the network input
// directly in the reference's NCHW layout (src/darknet.py:199: x is [B,3,H,W]) and writing NHWC in either
// activation format.  Replaces the pack_input + generic implicit-GEMM pair: K = 27 is too small for 32-wide
// K-chunks (the generic kernel pads it to 64), and the layer is HBM-write bound (608x608x32 outputs per frame).
//
// One wave = 32 output pixels x 32 output channels = one 32x32 MFMA tile; K = 27 (+1 zero) = 14 steps of
// v_mfma_f32_32x32x2_f32.  A operand: lane l -> pixel l&31, k = 2s + (l>>5), gathered straight from the three
// input planes (32 consecutive pixels per half-wave: coalesced 128-byte rows); B operand: 14 registers of
// BN-folded weights held for the whole kernel.  The tile is transposed through LDS so each lane stores eight
// consecutive channels (16 bytes per plane).  Numerics: exact fp32 fmaf chains (both precisions keep the stem exact).
#include "rtod_internal.h"

namespace rtod {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));

struct StemArgs {
    const float* x;          // [B,3,H,W]
    const float* w;          // [28][Cout] (k-major, k = (ky*3+kx)*3 + c, row 27 = 0), BN folded
    const float* bias;       // [Cout]
    float* out; int64_t out_ldc; int out_coff; int out_split;
    int B, H, W, Ho, Wo, stride, Cout, leaky;
};

__global__ __launch_bounds__(256)
void conv_stem_kernel(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) float T[4][32 * 36];       // per wave: 32 pixels x (32+4) floats
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, lh = lane >> 5;
    const int M = a.B * a.Ho * a.Wo;
    const int n_ct = a.Cout / 32;
    const int64_t plane = (int64_t)a.H * a.W;
    float* Tw = T[wave];

    for (int ct = 0; ct < n_ct; ++ct) {
        float wreg[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) wreg[s] = a.w[(2 * s + lh) * a.Cout + ct * 32 + lr];
        const float bias = a.bias[ct * 32 + lr];
        const int hw = a.Ho * a.Wo;
        // gather of one tile's A operand (14 dwords per lane); issued one tile ahead of its MFMAs
        auto gather = [&](int tile, float (&av)[14]) {
            const int m = tile * 32 + lr;
            const bool mok = m < M;
            const int mm = mok ? m : 0;
            const int b = mm / hw, r = mm - b * hw;
            const int oy = r / a.Wo, ox = r - oy * a.Wo;
            const int iy0 = oy * a.stride - 1, ix0 = ox * a.stride - 1;
            const float* xb = a.x + (int64_t)b * 3 * plane;
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                const int k = 2 * s + lh;                       // 0..27
                const int tap = k / 3, c = k - tap * 3;
                const int ky = tap / 3, kx = tap - ky * 3;
                const int iy = iy0 + ky, ix = ix0 + kx;
                const bool ok = mok && k < 27 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                av[s] = ok ? xb[c * plane + (int64_t)iy * a.W + ix] : 0.f;
            }
        };
        const int tstep = gridDim.x * 4;
        int tile = blockIdx.x * 4 + wave;
        float av[14], an[14];
        if (tile * 32 < M) gather(tile, av);
        for (; tile * 32 < M; tile += tstep) {
            if ((tile + tstep) * 32 < M) gather(tile + tstep, an);          // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wreg[s], acc, 0, 0, 0);
            // D: col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[e] + bias;
                v = apply_act(v, a.leaky);
                Tw[((e & 3) + 8 * (e >> 2) + 4 * lh) * 36 + lr] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the wave's own LDS writes, then reads (in order)
            __builtin_amdgcn_wave_barrier();
            if (a.out_split) {
                // 32 pixels x 8 pieces of 16 bytes (4 hi chunks, 4 lo chunks of 8 channels), four per lane.  With
                // out_ldc == 32 a pixel is 128 contiguous bytes [hi 64][lo 64]: every store instruction of the wave
                // writes 1 KiB contiguously.  sc1 write-through stores: see conv_f16s3_common.h store_act16.
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g = lane + q * 64;
                    const int p = g >> 3, j = g & 7, c8 = (j & 3) * 8;
                    const int mo = tile * 32 + p;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(Tw + p * 36 + c8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(Tw + p * 36 + c8 + 4);
                    f16x8s pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = (e < 4 ? v0[e] : v1[e - 4]) * SPLIT_SCALE;
                        const _Float16 h = (_Float16)v;
                        pk[e] = j < 4 ? h : (_Float16)(v - (float)h);
                    }
                    if (mo < M) {
                        _Float16* o = reinterpret_cast<_Float16*>(a.out) + (int64_t)mo * 2 * a.out_ldc + a.out_coff + ct * 32 + c8 + (j < 4 ? 0 : a.out_ldc);
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(o), "v"(pk) : "memory");
                    }
                }
            } else {
                // 32 pixels x 4 groups of 8 channels = 128 groups, two per lane
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int g = lane + q * 64;
                    const int p = g >> 2, c8 = (g & 3) * 8;
                    const int mo = tile * 32 + p;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(Tw + p * 36 + c8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(Tw + p * 36 + c8 + 4);
                    if (mo < M) {
                        float* o = a.out + (int64_t)mo * a.out_ldc + a.out_coff + ct * 32 + c8;
                        *reinterpret_cast<f32x4*>(o) = v0;
                        *reinterpret_cast<f32x4*>(o + 4) = v1;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 14; ++s) av[s] = an[s];
        }
    }
}

// ---- split-f16 variant (precision f16s3).  The exact-fp32 MFMA above issues 14 x 64 cycles per 32x32 tile and, with
// the conversion VALU of the split-format store, makes the layer issue-bound (~140 us at 608x608x8 against a 63 us
// HBM-write floor).  Here the 27-tap dot product is ONE k32 step of v_mfma_f32_16x16x32_f16 per 16x16 tile in the
// same three-product split arithmetic as every other layer of the plan (conv_igemm_f16s3.hip): x*8 = xh + xl,
// w*2^e = wh + wl, acc += xl*wh + xh*wl + xh*wh in fp32; 12 MFMAs x 16 cycles per 32x32 tile.
// A operand: lane l -> pixel l%16 of the 16-row tile, k = 8*(l/16) + e, k = (ky*3+kx)*3 + c (k >= 27: zero).
// B operand: [Cout][32] f16 hi / lo planes (plan.cpp packs them), lane l -> channel l%16, the same k group.
struct StemSplitArgs {
    const float* x; const _Float16* wh; const _Float16* wl; const float* inv_scale; const float* bias;
    _Float16* out; int64_t out_ldc; int out_coff;
    int B, H, W, Ho, Wo, stride, Cout, leaky;
    unsigned x_bytes;
    int32_t* ovf;
};

__global__ __launch_bounds__(256)
void conv_stem_split_kernel(const StemSplitArgs a) {
    __shared__ __attribute__((aligned(16))) float T[4][32 * 36];       // per wave: 32 pixels x (32+4) floats
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lh = lane >> 4;
    const int M = a.B * a.Ho * a.Wo;
    const int n_ct = a.Cout / 32;
    const int64_t plane = (int64_t)a.H * a.W;
    const int hw = a.Ho * a.Wo;
    float* Tw = T[wave];
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    int koff[8], need[8];
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = lh * 8 + e;                                  // 0..31, k = (ky*3+kx)*3 + c
        const int tap = k / 3, c = k - tap * 3;
        const int ky = tap / 3, kx = tap - ky * 3;
        koff[e] = (c * (int)plane + ky * a.W + kx) * 4;
        need[e] = (ky == 0 ? 1 : 0) | (ky == 2 ? 2 : 0) | (kx == 0 ? 4 : 0) | (kx == 2 ? 8 : 0) | (k >= 27 ? 16 : 0) | 32;
    }

    for (int ct = 0; ct < n_ct; ++ct) {
        f16x8s bh[2], bl[2];
        float inv[2], bias[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = ct * 32 + j * 16 + lr;
            bh[j] = *reinterpret_cast<const f16x8s*>(a.wh + (int64_t)n * 32 + lh * 8);
            bl[j] = *reinterpret_cast<const f16x8s*>(a.wl + (int64_t)n * 32 + lh * 8);
            inv[j] = a.inv_scale[n]; bias[j] = a.bias[n];
        }
        // one tile's A operand: 2 x 8 input samples per lane, gathered one tile ahead of its MFMAs.  The layer is VALU-bound
        // if every sample decodes its own tap: the tap geometry of a lane's 8 k values is fixed, so it is folded once into
        // an element offset and a 'needs' mask (top / bottom / left / right neighbour, k >= 27, always); per tile only the
        // pixel's origin and edge mask are computed, and a masked sample is a buffer load out of range (-> 0).
        auto gather = [&](int tile, float (&av)[2][8]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = tile * 32 + i * 16 + lr;
                const bool mok = m < M;
                const int mm = mok ? m : 0;
                const int b = mm / hw, r = mm - b * hw;
                const int oy = r / a.Wo, ox = r - oy * a.Wo;
                const int iy0 = oy * a.stride - 1, ix0 = ox * a.stride - 1;
                const int edge = (iy0 < 0 ? 1 : 0) | (iy0 + 2 >= a.H ? 2 : 0) | (ix0 < 0 ? 4 : 0) | (ix0 + 2 >= a.W ? 8 : 0) | 16 | (mok ? 0 : 32);
                const int base = (b * 3 * (int)plane + iy0 * a.W + ix0) * 4;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned vo = (need[e] & edge) ? 0x80000000u : (unsigned)(base + koff[e]);
                    av[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, vo, 0, 0));
                }
            }
        };
        const int tstep = gridDim.x * 4;
        int tile = blockIdx.x * 4 + wave;
        float av[2][8], an[2][8];
        if (tile * 32 < M) gather(tile, av);
        for (; tile * 32 < M; tile += tstep) {
            if ((tile + tstep) * 32 < M) gather(tile + tstep, an);          // wave-uniform
            f16x8s ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = av[i][e] * SPLIT_SCALE;
                    const _Float16 h = (_Float16)v;
                    ah[i][e] = h; al[i][e] = (_Float16)(v - (float)h);
                }
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
            // D: col = lane%16 (channel), row = 4*(lane/16) + e (pixel)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][e] * inv[j] + bias[j];
                        v = apply_act(v, a.leaky);
                        Tw[(i * 16 + 4 * lh + e) * 36 + j * 16 + lr] = v;
                    }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the wave's own LDS writes, then reads (in order)
            __builtin_amdgcn_wave_barrier();
            // 32 pixels x 8 pieces of 16 bytes (4 hi chunks, 4 lo chunks), four per lane: see conv_stem_kernel
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = lane + q * 64;
                const int p = g >> 3, j = g & 7, c8 = (j & 3) * 8;
                const int mo = tile * 32 + p;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(Tw + p * 36 + c8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(Tw + p * 36 + c8 + 4);
                f16x8s pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 h, l;
                    split_f16((e < 4 ? v0[e] : v1[e - 4]) * SPLIT_SCALE, h, l, amax);
                    pk[e] = j < 4 ? h : l;
                }
                if (mo < M) {
                    _Float16* o = a.out + (int64_t)mo * 2 * a.out_ldc + a.out_coff + ct * 32 + c8 + (j < 4 ? 0 : a.out_ldc);
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(o), "v"(pk) : "memory");
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) av[i][e] = an[i][e];
        }
    }
    split_overflow_report(a.ovf, amax);
}

int launch_conv_stem_split(const float* x, const _Float16* wh, const _Float16* wl, const float* inv_scale, const float* bias,
                           const View& out, int B, int H, int W, int Ho, int Wo, int stride, int Cout, int leaky, int32_t* ovf, hipStream_t s) {
    if (!x || !wh || !wl || !inv_scale || !bias || !out.base) { set_error("conv_stem_split: null pointer"); return RTOD_E_ARG; }
    if (Cout % 32 || Cout < 32 || out.C != Cout || out.H != Ho || out.W != Wo || out.ldc % 8 || out.coff % 8 || !out.split) { set_error("conv_stem_split: bad output view"); return RTOD_E_ARG; }
    if ((int64_t)B * Ho * Wo >= (1ll << 31) || (int64_t)B * 3 * H * W * 4 >= (1ll << 31)) { set_error("conv_stem_split: input exceeds 2 GiB / int32 pixels"); return RTOD_E_ARG; }
    StemSplitArgs a;
    a.x_bytes = (unsigned)((int64_t)B * 3 * H * W * 4);
    a.ovf = ovf;
    a.x = x; a.wh = wh; a.wl = wl; a.inv_scale = inv_scale; a.bias = bias;
    a.out = reinterpret_cast<_Float16*>(out.base); a.out_ldc = out.ldc; a.out_coff = out.coff;
    a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.Cout = Cout; a.leaky = leaky;
    const int64_t tiles = ((int64_t)B * Ho * Wo + 31) / 32;
    int grid = (int)((tiles + 3) / 4);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(conv_stem_split_kernel, dim3(grid), dim3(256), 0, s, a);
    return hip_fail(hipGetLastError(), "conv_stem_split launch");
}

int launch_conv_stem(const float* x, const float* w, const float* bias, const View& out, int B, int H, int W,
                     int Ho, int Wo, int stride, int Cout, int leaky, hipStream_t s) {
    if (!x || !w || !bias || !out.base) { set_error("conv_stem: null pointer"); return RTOD_E_ARG; }
    if (Cout % 32 || Cout < 32 || out.C != Cout || out.H != Ho || out.W != Wo || out.ldc % 8 || out.coff % 8) { set_error("conv_stem: bad output view"); return RTOD_E_ARG; }
    if ((int64_t)B * Ho * Wo >= (1ll << 31)) { set_error("conv_stem: pixel count exceeds int32"); return RTOD_E_ARG; }
    StemArgs a;
    a.x = x; a.w = w; a.bias = bias; a.out = out.base; a.out_ldc = out.ldc; a.out_coff = out.coff; a.out_split = out.split;
    a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.Cout = Cout; a.leaky = leaky;
    const int64_t tiles = ((int64_t)B * Ho * Wo + 31) / 32;
    int grid = (int)((tiles + 3) / 4);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(conv_stem_kernel, dim3(grid), dim3(256), 0, s, a);
    return hip_fail(hipGetLastError(), "conv_stem launch");
}

}  // namespace rtod
