// Bandwidth-bound helper kernels of the Darknet forward (gfx950): input NCHW->NHWC pack, bilinear x2
// upsample, stand-alone shortcut add, max-pool, channel copy, NHWC->NCHW read-back, stand-alone head
// decode, confidence mask, IoU.  All are HBM-bound elementwise work: one float4 (16 B) per lane
// along the channel axis wherever the layout allows, grid capped at ~2048 blocks + grid stride.
#include "rtod_internal.h"
#include <algorithm>

namespace rtod {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(int64_t work, int block) {
    int64_t g = (work + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------
// input pack: x [B,C,H,W] (NCHW, C=3) -> [B,H,W,Cp] with channels C..Cp-1 zero.  The stem conv then
// runs through the generic implicit-GEMM with Cin = Cp = 4 (16-B pixel = one float4 gather).
__global__ void pack_input_kernel(const float* __restrict__ x, int B, int C, int H, int W,
                                  float* __restrict__ out, int Cp) {
    const int64_t npix = (int64_t)B * H * W;
    const int64_t hw = (int64_t)H * W;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p / hw, r = p - b * hw;
        const float* src = x + b * C * hw + r;
        for (int c0 = 0; c0 < Cp; c0 += 4) {
            f32x4 v;
            v[0] = c0 + 0 < C ? src[(c0 + 0) * hw] : 0.f;
            v[1] = c0 + 1 < C ? src[(c0 + 1) * hw] : 0.f;
            v[2] = c0 + 2 < C ? src[(c0 + 2) * hw] : 0.f;
            v[3] = c0 + 3 < C ? src[(c0 + 3) * hw] : 0.f;
            *reinterpret_cast<f32x4*>(out + p * Cp + c0) = v;
        }
    }
}

int launch_pack_input(const float* x, int B, int C, int H, int W, float* out, int Cp, hipStream_t s) {
    if (!x || !out || Cp % 4 || Cp < C) { set_error("pack_input: bad args"); return RTOD_E_ARG; }
    const int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, s, x, B, C, H, W, out, Cp);
    return hip_fail(hipGetLastError(), "pack_input launch");
}

// ---------------------------------------------------------------------------------------------
// bilinear x2, align_corners=False (reference: nn.Upsample(scale_factor=2, mode="bilinear"),
// src/darknet.py:587-593).  src = (dst+0.5)/2-0.5 clamped at 0; weights are
// exactly {1,0} on the borders and {0.25,0.75} inside.  Evaluated as wy0*(wx0*a+wx1*b)+wy1*(wx0*c+wx1*d),
// ATen's separable order.
__device__ __forceinline__ void up_coord(int o, int n, int& i0, int& i1, float& w1) {
    float src = ((float)o + 0.5f) * 0.5f - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    w1 = src - (float)i0;
}

__global__ void upsample2x_kernel(View in, View out, int B) {
    const int C4 = in.C / 4;
    const int64_t total = (int64_t)B * out.H * out.W * C4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C4) * 4;
        int64_t p = t / C4;
        const int ox = (int)(p % out.W); p /= out.W;
        const int oy = (int)(p % out.H);
        const int b = (int)(p / out.H);
        int y0, y1, x0, x1; float wy1, wx1;
        up_coord(oy, in.H, y0, y1, wy1);
        up_coord(ox, in.W, x0, x1, wx1);
        const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        const float* base = in.base + in.coff + c;
        const int64_t rb = (int64_t)b * in.H;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((rb + y0) * in.W + x0) * in.ldc);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((rb + y0) * in.W + x1) * in.ldc);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((rb + y1) * in.W + x0) * in.ldc);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((rb + y1) * in.W + x1) * in.ldc);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            r[e] = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
        *reinterpret_cast<f32x4*>(out.base + out.coff + c + (((int64_t)b * out.H + oy) * out.W + ox) * out.ldc) = r;
    }
}

// split-format variant: 8 channels (16 B of each plane) per thread; interpolation is linear, so it runs
// directly on hi+lo (the SPLIT_SCALE factor carries through) and the result is re-split.
typedef _Float16 f16x8a __attribute__((ext_vector_type(8)));
__global__ void upsample2x_split_kernel(View in, View out, int B) {
    const int C8 = in.C / 8;
    const int64_t total = (int64_t)B * out.H * out.W * C8;
    const _Float16* ib = reinterpret_cast<const _Float16*>(in.base);
    _Float16* ob = reinterpret_cast<_Float16*>(out.base);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C8) * 8;
        int64_t p = t / C8;
        const int ox = (int)(p % out.W); p /= out.W;
        const int oy = (int)(p % out.H);
        const int b = (int)(p / out.H);
        int y0, y1, x0, x1; float wy1, wx1;
        up_coord(oy, in.H, y0, y1, wy1);
        up_coord(ox, in.W, x0, x1, wx1);
        const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        const int64_t rb = (int64_t)b * in.H;
        auto ld = [&](int y, int x, float* v) {
            const _Float16* q = ib + ((rb + y) * in.W + x) * 2 * in.ldc + in.coff + c;
            const f16x8a h = *reinterpret_cast<const f16x8a*>(q);
            const f16x8a l = *reinterpret_cast<const f16x8a*>(q + in.ldc);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)h[e] + (float)l[e];
        };
        float v00[8], v01[8], v10[8], v11[8];
        ld(y0, x0, v00); ld(y0, x1, v01); ld(y1, x0, v10); ld(y1, x1, v11);
        f16x8a rh, rl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float r = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
            const _Float16 h = (_Float16)r;
            rh[e] = h; rl[e] = (_Float16)(r - (float)h);
        }
        _Float16* q = ob + (((int64_t)b * out.H + oy) * out.W + ox) * 2 * out.ldc + out.coff + c;
        *reinterpret_cast<f16x8a*>(q) = rh;
        *reinterpret_cast<f16x8a*>(q + out.ldc) = rl;
    }
}

static bool view_ok4(const View& v) { return v.base && v.C % 4 == 0 && v.ldc % 4 == 0 && v.coff % 4 == 0; }

int launch_upsample2x(const View& in, const View& out, int B, hipStream_t s) {
    if (in.split != out.split) { set_error("upsample2x: mixed formats"); return RTOD_E_ARG; }
    if (in.split) {
        if (!in.base || !out.base || in.C % 8 || in.ldc % 8 || in.coff % 8 || out.ldc % 8 || out.coff % 8 ||
            out.H != 2 * in.H || out.W != 2 * in.W || out.C != in.C) { set_error("upsample2x(split): bad views"); return RTOD_E_ARG; }
        const int64_t total = (int64_t)B * out.H * out.W * (in.C / 8);
        hipLaunchKernelGGL(upsample2x_split_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, out, B);
        return hip_fail(hipGetLastError(), "upsample2x(split) launch");
    }
    if (!view_ok4(in) || !view_ok4(out) || out.H != 2 * in.H || out.W != 2 * in.W || out.C != in.C) {
        set_error("upsample2x: bad views"); return RTOD_E_ARG;
    }
    const int64_t total = (int64_t)B * out.H * out.W * (in.C / 4);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, out, B);
    return hip_fail(hipGetLastError(), "upsample2x launch");
}

// ---------------------------------------------------------------------------------------------
// stand-alone shortcut (src/darknet.py:263-268) for cfgs where the add cannot ride a conv epilogue
__global__ void add_kernel(View a, View b, View out, int B) {
    const int C4 = a.C / 4;
    const int64_t total = (int64_t)B * a.H * a.W * C4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C4) * 4;
        const int64_t p = t / C4;
        const f32x4 x = *reinterpret_cast<const f32x4*>(a.base + a.coff + c + p * a.ldc);
        const f32x4 y = *reinterpret_cast<const f32x4*>(b.base + b.coff + c + p * b.ldc);
        *reinterpret_cast<f32x4*>(out.base + out.coff + c + p * out.ldc) = x + y;
    }
}

int launch_add(const View& a, const View& b, const View& out, int B, hipStream_t s) {
    if (a.split || b.split || out.split) { set_error("add: split-format views unsupported (shortcut must fuse into a conv)"); return RTOD_E_ARG; }
    if (!view_ok4(a) || !view_ok4(b) || !view_ok4(out) || a.C != b.C || a.C != out.C || a.H != b.H || a.W != b.W) {
        set_error("add: bad views"); return RTOD_E_ARG;
    }
    const int64_t total = (int64_t)B * a.H * a.W * (a.C / 4);
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, a, b, out, B);
    return hip_fail(hipGetLastError(), "add launch");
}

// channel-slice copy (fallback when a route concat cannot be zero-copy)
__global__ void copy_kernel(View a, View out, int B) {
    const int C4 = a.C / 4;
    const int64_t total = (int64_t)B * a.H * a.W * C4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C4) * 4;
        const int64_t p = t / C4;
        *reinterpret_cast<f32x4*>(out.base + out.coff + c + p * out.ldc) =
            *reinterpret_cast<const f32x4*>(a.base + a.coff + c + p * a.ldc);
    }
}

int launch_copy(const View& a, const View& out, int B, hipStream_t s) {
    if (a.split || out.split) { set_error("copy: split-format views unsupported (route concat must be zero-copy)"); return RTOD_E_ARG; }
    if (!view_ok4(a) || !view_ok4(out) || a.C != out.C || a.H != out.H || a.W != out.W) {
        set_error("copy: bad views"); return RTOD_E_ARG;
    }
    const int64_t total = (int64_t)B * a.H * a.W * (a.C / 4);
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, a, out, B);
    return hip_fail(hipGetLastError(), "copy launch");
}

// ---------------------------------------------------------------------------------------------
// max-pool: size/stride floor mode (nn.MaxPool2d, src/darknet.py:547-555); stride 1 = MaxPoolStride1
// (src/darknet.py:17-46): replicate-pad right/bottom by size-1 then size/1 pool == clamp the window.
// pad > 0 (cfg extension `symmetric=1`, SPPF-style pools): the window starts pad pixels up / left and out-of-image taps
// count as -inf (nn.MaxPool2d(size, stride, pad)).
__global__ void maxpool_kernel(View in, View out, int B, int size, int stride, int pad) {
    const int C4 = in.C / 4;
    const int64_t total = (int64_t)B * out.H * out.W * C4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C4) * 4;
        int64_t p = t / C4;
        const int ox = (int)(p % out.W); p /= out.W;
        const int oy = (int)(p % out.H);
        const int b = (int)(p / out.H);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int dy = 0; dy < size; ++dy) {
            int iy = oy * stride + dy - pad;
            if (pad) { if ((unsigned)iy >= (unsigned)in.H) continue; } else if (iy > in.H - 1) iy = in.H - 1;
            for (int dx = 0; dx < size; ++dx) {
                int ix = ox * stride + dx - pad;
                if (pad) { if ((unsigned)ix >= (unsigned)in.W) continue; } else if (ix > in.W - 1) ix = in.W - 1;
                const f32x4 v = *reinterpret_cast<const f32x4*>(in.base + in.coff + c + (((int64_t)b * in.H + iy) * in.W + ix) * in.ldc);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        *reinterpret_cast<f32x4*>(out.base + out.coff + c + (((int64_t)b * out.H + oy) * out.W + ox) * out.ldc) = m;
    }
}

// split-format variant: the maximum is taken over hi + lo (exact in fp32: 22 significant bits) and the winning element's
// (hi, lo) pair is stored as it stands — no re-split, no rounding.  8 channels (16 B of each plane) per thread.
__global__ void maxpool_split_kernel(View in, View out, int B, int size, int stride, int pad) {
    const int C8 = in.C / 8;
    const int64_t total = (int64_t)B * out.H * out.W * C8;
    const _Float16* ib = reinterpret_cast<const _Float16*>(in.base);
    _Float16* ob = reinterpret_cast<_Float16*>(out.base);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C8) * 8;
        int64_t p = t / C8;
        const int ox = (int)(p % out.W); p /= out.W;
        const int oy = (int)(p % out.H);
        const int b = (int)(p / out.H);
        float m[8];
        f16x8a mh, ml;
#pragma unroll
        for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; mh[e] = (_Float16)0.f; ml[e] = (_Float16)0.f; }
        for (int dy = 0; dy < size; ++dy) {
            int iy = oy * stride + dy - pad;
            if (pad) { if ((unsigned)iy >= (unsigned)in.H) continue; } else if (iy > in.H - 1) iy = in.H - 1;
            for (int dx = 0; dx < size; ++dx) {
                int ix = ox * stride + dx - pad;
                if (pad) { if ((unsigned)ix >= (unsigned)in.W) continue; } else if (ix > in.W - 1) ix = in.W - 1;
                const _Float16* q = ib + (((int64_t)b * in.H + iy) * in.W + ix) * 2 * in.ldc + in.coff + c;
                const f16x8a h = *reinterpret_cast<const f16x8a*>(q);
                const f16x8a l = *reinterpret_cast<const f16x8a*>(q + in.ldc);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (float)h[e] + (float)l[e];
                    if (v > m[e]) { m[e] = v; mh[e] = h[e]; ml[e] = l[e]; }
                }
            }
        }
        _Float16* o = ob + (((int64_t)b * out.H + oy) * out.W + ox) * 2 * out.ldc + out.coff + c;
        *reinterpret_cast<f16x8a*>(o) = mh;
        *reinterpret_cast<f16x8a*>(o + out.ldc) = ml;
    }
}

int launch_maxpool(const View& in, const View& out, int B, int size, int stride, int pad, hipStream_t s) {
    if (in.split != out.split) { set_error("maxpool: mixed activation formats"); return RTOD_E_ARG; }
    if (in.C != out.C || size < 1 || stride < 1 || pad < 0 || pad >= size) { set_error("maxpool: bad geometry"); return RTOD_E_ARG; }
    const int eh = pad ? (in.H + 2 * pad - size) / stride + 1 : (stride != 1 ? (in.H - size) / stride + 1 : in.H);
    const int ew = pad ? (in.W + 2 * pad - size) / stride + 1 : (stride != 1 ? (in.W - size) / stride + 1 : in.W);
    if (out.H != eh || out.W != ew) { set_error("maxpool: output %dx%d, expected %dx%d", out.H, out.W, eh, ew); return RTOD_E_ARG; }
    if (in.split) {
        if (!in.base || !out.base || in.C % 8 || in.coff % 8 || out.coff % 8 || in.ldc % 8 || out.ldc % 8) { set_error("maxpool: bad split views"); return RTOD_E_ARG; }
        const int64_t total = (int64_t)B * out.H * out.W * (in.C / 8);
        hipLaunchKernelGGL(maxpool_split_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, out, B, size, stride, pad);
        return hip_fail(hipGetLastError(), "maxpool_split launch");
    }
    if (!view_ok4(in) || !view_ok4(out)) { set_error("maxpool: bad views"); return RTOD_E_ARG; }
    const int64_t total = (int64_t)B * out.H * out.W * (in.C / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, out, B, size, stride, pad);
    return hip_fail(hipGetLastError(), "maxpool launch");
}

// ---------------------------------------------------------------------------------------------
// nearest x2 (cfg extension `[upsample] mode=nearest`; nn.Upsample(scale_factor=2, mode="nearest")): out(y, x) = in(y/2, x/2).
// A pure copy in either activation format (split: both planes move unchanged).
__global__ void upsample_nearest2x_kernel(View in, View out, int B) {
    const int per = in.split ? 8 : 4;                         // channels per thread: 16 bytes of a plane / of the fp32 pixel
    const int CV = in.C / per;
    const int64_t total = (int64_t)B * out.H * out.W * CV;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % CV) * per;
        int64_t p = t / CV;
        const int ox = (int)(p % out.W); p /= out.W;
        const int oy = (int)(p % out.H);
        const int b = (int)(p / out.H);
        const int64_t ip = ((int64_t)b * in.H + (oy >> 1)) * in.W + (ox >> 1), op = ((int64_t)b * out.H + oy) * out.W + ox;
        if (in.split) {
            const _Float16* q = reinterpret_cast<const _Float16*>(in.base) + ip * 2 * in.ldc + in.coff + c;
            _Float16* o = reinterpret_cast<_Float16*>(out.base) + op * 2 * out.ldc + out.coff + c;
            *reinterpret_cast<f16x8a*>(o) = *reinterpret_cast<const f16x8a*>(q);
            *reinterpret_cast<f16x8a*>(o + out.ldc) = *reinterpret_cast<const f16x8a*>(q + in.ldc);
        } else {
            *reinterpret_cast<f32x4*>(out.base + out.coff + c + op * out.ldc) = *reinterpret_cast<const f32x4*>(in.base + in.coff + c + ip * in.ldc);
        }
    }
}

int launch_upsample_nearest2x(const View& in, const View& out, int B, hipStream_t s) {
    if (!in.base || !out.base || in.split != out.split || in.C != out.C || out.H != 2 * in.H || out.W != 2 * in.W) { set_error("upsample_nearest: bad views"); return RTOD_E_ARG; }
    const int per = in.split ? 8 : 4;
    if (in.C % per || in.coff % per || out.coff % per || in.ldc % per || out.ldc % per) { set_error("upsample_nearest: channel alignment"); return RTOD_E_ARG; }
    const int64_t total = (int64_t)B * out.H * out.W * (in.C / per);
    hipLaunchKernelGGL(upsample_nearest2x_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, out, B);
    return hip_fail(hipGetLastError(), "upsample_nearest launch");
}

// ---------------------------------------------------------------------------------------------
// Batch-statistics BatchNorm ("as run" by the reference: detect.py never calls .eval(), so nn.BatchNorm2d normalises
// with the statistics of the batch, src/darknet.py:493-495).  Optional parity mode of the exact-fp32 plans
// (plan option bn_batch_stats): the conv writes its raw sums, bn_stats_kernel reduces per-channel mean and biased
// variance over (B, H, W) in double precision (PyTorch's CPU kernels accumulate float statistics in double), then
// bn_apply_kernel normalises, applies the activation and the fused shortcut in place.
// stats layout: [sstride] mean, [sstride] biased variance (doubles).
__global__ __launch_bounds__(256)
void bn_stats_kernel(View x, int B, double* __restrict__ stats, int sstride) {
    // one workgroup per group of 4 channels; threads stride over the pixels (deterministic: fixed partition, ordered tree)
    const int c = blockIdx.x * 4;
    const int64_t npix = (int64_t)B * x.H * x.W;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int64_t p = threadIdx.x; p < npix; p += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x.base + x.coff + c + p * x.ldc);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += (double)v[e]; q[e] += (double)v[e] * (double)v[e]; }
    }
    __shared__ double red[2][4][256];
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][e][threadIdx.x] = s[e]; red[1][e][threadIdx.x] = q[e]; }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[0][e][threadIdx.x] += red[0][e][threadIdx.x + w]; red[1][e][threadIdx.x] += red[1][e][threadIdx.x + w]; }
        }
        __syncthreads();
    }
    if (threadIdx.x < 4 && c + (int)threadIdx.x < x.C) {
        const int e = threadIdx.x;
        const double mean = red[0][e][0] / (double)npix;
        double var = red[1][e][0] / (double)npix - mean * mean;
        if (var < 0) var = 0;
        stats[c + e] = mean;
        stats[sstride + c + e] = var;
    }
}

// y = (x - mean) / sqrt(var + eps) * gamma + beta, activation, + shortcut; x and y may be the same view.
// bn: [gstride] beta, [gstride] gamma (the order of the .weights stream, src/darknet.py:356-376)
__global__ void bn_apply_kernel(View x, View y, View res, int has_res, int B, const double* __restrict__ stats, int sstride,
                                const float* __restrict__ bn, int gstride, int act) {
    const int C4 = x.C / 4;
    const int64_t total = (int64_t)B * x.H * x.W * C4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C4) * 4;
        const int64_t p = t / C4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x.base + x.coff + c + p * x.ldc);
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (has_res) r = *reinterpret_cast<const f32x4*>(res.base + res.coff + c + p * res.ldc);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double invstd = 1.0 / sqrt(stats[sstride + c + e] + 1e-5);
            const float w = (float)(invstd * (double)bn[gstride + c + e]);                      // gamma / sqrt(var + eps)
            const float b = (float)((double)bn[c + e] - stats[c + e] * invstd * (double)bn[gstride + c + e]);
            float u = v[e] * w + b;
            u = apply_act(u, act);
            o[e] = has_res ? u + r[e] : u;
        }
        *reinterpret_cast<f32x4*>(y.base + y.coff + c + p * y.ldc) = o;
    }
}

// Two-stage statistics (round 4): bn_stats_kernel above reads 16 bytes per pixel row from C/4 workgroups (8 for the first layer) —
// every line of the tensor fetched for one of its eight 16-byte pieces at a time.  Here the pixels are cut into `nparts` ranges, a
// workgroup reads its range with consecutive threads on consecutive channels (whole lines), reduces its 256 / (C/4) pixel lanes
// through LDS in a fixed tree and writes per-channel partial sums; bn_stats_final_kernel adds the parts in ascending order.
// Fixed partition, fixed order: deterministic.  Needs 256 % (C/4) == 0 or (C/4) % 256 == 0; other channel counts keep the old kernel.
constexpr int BN_PARTS_MAX = 1024;
__global__ __launch_bounds__(256)
void bn_stats_partial_kernel(View x, int B, double* __restrict__ partial, int pstride, int nparts) {
    const int C4 = x.C / 4;
    const int64_t npix = (int64_t)B * x.H * x.W;
    const int64_t per = (npix + nparts - 1) / nparts;
    const int64_t p0 = blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    __shared__ double red[2][4][256];
    for (int cb = 0; cb < C4; cb += 256) {
        const int W = C4 - cb < 256 ? C4 - cb : 256;            // channel quads handled in this pass (divides 256)
        const int PL = 256 / W;                                 // pixel lanes
        const int cq = cb + (int)threadIdx.x % W, pl = (int)threadIdx.x / W;
        double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        for (int64_t p = p0 + pl; p < p1; p += PL) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x.base + x.coff + cq * 4 + p * x.ldc);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += (double)v[e]; q[e] += (double)v[e] * (double)v[e]; }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][e][threadIdx.x] = s[e]; red[1][e][threadIdx.x] = q[e]; }
        __syncthreads();
        for (int w = PL / 2; w > 0; w >>= 1) {                  // lanes pl and pl + w of the same channel quad
            if (pl < w) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { red[0][e][threadIdx.x] += red[0][e][threadIdx.x + w * W]; red[1][e][threadIdx.x] += red[1][e][threadIdx.x + w * W]; }
            }
            __syncthreads();
        }
        if (pl == 0) {
            double* o = partial + (int64_t)blockIdx.x * 2 * pstride;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[cq * 4 + e] = red[0][e][threadIdx.x]; o[pstride + cq * 4 + e] = red[1][e][threadIdx.x]; }
        }
    }
}

// 16 channels per workgroup, 16 lanes per channel: lane k adds parts k, k + 16, ... in ascending order, then a fixed tree over the lanes
// (one thread per channel walking all parts was a chain of `nparts` dependent L2 round trips: 29 us per layer)
__global__ __launch_bounds__(256)
void bn_stats_final_kernel(const double* __restrict__ partial, int pstride, int nparts, int C, double npix, double* __restrict__ stats, int sstride,
                           const float* __restrict__ bn, int gstride, float* __restrict__ wb) {
    const int cl = threadIdx.x & 15, k0 = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0, q = 0;
    if (c < C)
        for (int k = k0; k < nparts; k += 16) { s += partial[(int64_t)k * 2 * pstride + c]; q += partial[(int64_t)k * 2 * pstride + pstride + c]; }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 8; w > 0; w >>= 1) {
        if (k0 < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w * 16]; red[1][threadIdx.x] += red[1][threadIdx.x + w * 16]; }
        __syncthreads();
    }
    if (k0 != 0 || c >= C) return;
    s = red[0][threadIdx.x]; q = red[1][threadIdx.x];
    const double mean = s / npix;
    double var = q / npix - mean * mean;
    if (var < 0) var = 0;
    stats[c] = mean;
    stats[sstride + c] = var;
    // the normalisation's per-channel constants, once per channel instead of once per element (same expressions as bn_apply_kernel)
    const double invstd = 1.0 / sqrt(var + 1e-5);
    wb[c] = (float)(invstd * (double)bn[gstride + c]);
    wb[sstride + c] = (float)((double)bn[c] - mean * invstd * (double)bn[gstride + c]);
}

// y = x * w + b, activation, + shortcut with the per-channel constants of bn_stats_final_kernel
__global__ void bn_apply_wb_kernel(View x, View y, View res, int has_res, int B, const float* __restrict__ wb, int sstride, int act) {
    const int C4 = x.C / 4;
    const int64_t total = (int64_t)B * x.H * x.W * C4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C4) * 4;
        const int64_t p = t / C4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x.base + x.coff + c + p * x.ldc);
        const f32x4 w = *reinterpret_cast<const f32x4*>(wb + c), b = *reinterpret_cast<const f32x4*>(wb + sstride + c);
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (has_res) r = *reinterpret_cast<const f32x4*>(res.base + res.coff + c + p * res.ldc);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = v[e] * w[e] + b[e];
            u = apply_act(u, act);
            o[e] = has_res ? u + r[e] : u;
        }
        *reinterpret_cast<f32x4*>(y.base + y.coff + c + p * y.ldc) = o;
    }
}

size_t bn_partial_doubles(int max_channels) { return (size_t)BN_PARTS_MAX * 2 * (size_t)max_channels + (size_t)max_channels; }   // partial sums + [2][C] float constants

int launch_bn_batch(const View& x, const View& y, const View* res, int B, double* stats, int sstride, const float* bn, int gstride, int act,
                    double* partial, int64_t partial_doubles, hipStream_t s) {
    if (x.split || y.split || (res && res->split)) { set_error("bn_batch: exact-fp32 plans only"); return RTOD_E_ARG; }
    if (!view_ok4(x) || !view_ok4(y) || x.C != y.C || x.H != y.H || x.W != y.W || !stats || !bn || gstride < x.C || sstride < x.C) { set_error("bn_batch: bad views"); return RTOD_E_ARG; }
    if (res && (!view_ok4(*res) || res->C != x.C || res->H != x.H || res->W != x.W)) { set_error("bn_batch: bad shortcut view"); return RTOD_E_ARG; }
    const int C4 = x.C / 4;
    const int64_t npix = (int64_t)B * x.H * x.W;
    int nparts = (int)std::min<int64_t>(BN_PARTS_MAX, std::max<int64_t>(1, npix / 128));
    if (partial && (256 % C4 == 0 || C4 % 256 == 0) && (int64_t)nparts * 2 * sstride + sstride <= partial_doubles) {
        hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nparts), dim3(256), 0, s, x, B, partial, sstride, nparts);
        if (hipGetLastError() != hipSuccess) return hip_fail(hipGetLastError(), "bn_stats_partial launch");
        float* wb = reinterpret_cast<float*>(partial + (int64_t)nparts * 2 * sstride);
        hipLaunchKernelGGL(bn_stats_final_kernel, dim3((x.C + 15) / 16), dim3(256), 0, s, partial, sstride, nparts, x.C, (double)npix, stats, sstride, bn, gstride, wb);
        if (hipGetLastError() != hipSuccess) return hip_fail(hipGetLastError(), "bn_stats_final launch");
        View r = res ? *res : x;
        hipLaunchKernelGGL(bn_apply_wb_kernel, dim3(grid_for(npix * C4, 256)), dim3(256), 0, s, x, y, r, res ? 1 : 0, B, wb, sstride, act);
        return hip_fail(hipGetLastError(), "bn_apply launch");
    } else {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(x.C / 4), dim3(256), 0, s, x, B, stats, sstride);
        if (hipGetLastError() != hipSuccess) return hip_fail(hipGetLastError(), "bn_stats launch");
    }
    const int64_t total = npix * C4;
    View r = res ? *res : x;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, x, y, r, res ? 1 : 0, B, stats, sstride, bn, gstride, act);
    return hip_fail(hipGetLastError(), "bn_apply launch");
}

// Training-mode side effect of nn.BatchNorm2d on its buffers, for ALL BatchNorm layers of a plan in one launch (the host class
// used to fetch every layer's statistics with a synchronising call and update the module buffers with two small copies each:
// 72 round trips + 144 copies per forward at YOLOv3).  running = (1 - momentum) * running + momentum * batch statistic in float32,
// the variance UNBIASED (n / (n - 1)), as torch does (momentum 0.1 by default; src/darknet.py:493-495 builds plain BatchNorm2d).
// One workgroup per (layer, 256 channels).
__global__ __launch_bounds__(256)
void bn_update_running_kernel(BnUpdateTable t, const double* __restrict__ stats, float momentum, double momentum_d) {
    const BnUpdateEntry e = t.e[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= e.channels) return;
    const double mean = stats[e.stats_off + c];
    const double var = stats[e.stats_off + e.sstride + c] * e.unbias;
    // mul_(1 - momentum) then add_(float32(momentum * statistic)): two rounded float32 operations, no contraction
    e.running_mean[c] = __fadd_rn(__fmul_rn(e.running_mean[c], 1.0f - momentum), (float)(momentum_d * mean));
    e.running_var[c] = __fadd_rn(__fmul_rn(e.running_var[c], 1.0f - momentum), (float)(momentum_d * var));
}

int launch_bn_update_running(const BnUpdateEntry* entries, int n, const double* stats, double momentum, hipStream_t s) {
    if (!entries || n < 1 || !stats) { set_error("bn_update_running: bad args"); return RTOD_E_ARG; }
    for (int i0 = 0; i0 < n; i0 += BN_UPDATE_MAX) {
        BnUpdateTable t;
        const int m = n - i0 < BN_UPDATE_MAX ? n - i0 : BN_UPDATE_MAX;
        int cmax = 0;
        for (int i = 0; i < m; ++i) { t.e[i] = entries[i0 + i]; if (t.e[i].channels > cmax) cmax = t.e[i].channels; }
        hipLaunchKernelGGL(bn_update_running_kernel, dim3((cmax + 255) / 256, m), dim3(256), 0, s, t, stats, (float)momentum, momentum);
        if (hipGetLastError() != hipSuccess) return hip_fail(hipGetLastError(), "bn_update_running launch");
    }
    return RTOD_OK;
}

// ---------------------------------------------------------------------------------------------
// NHWC view -> dense NCHW (test/debug read-back of a layer output)
__global__ void view_to_nchw_kernel(View in, int B, float* __restrict__ out) {
    const int64_t total = (int64_t)B * in.C * in.H * in.W;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = t;
        const int x = (int)(p % in.W); p /= in.W;
        const int y = (int)(p % in.H); p /= in.H;
        const int c = (int)(p % in.C);
        const int b = (int)(p / in.C);
        const int64_t pix = ((int64_t)b * in.H + y) * in.W + x;
        if (in.split) {
            const _Float16* q = reinterpret_cast<const _Float16*>(in.base) + pix * 2 * in.ldc + in.coff + c;
            out[t] = ((float)q[0] + (float)q[in.ldc]) * (1.0f / SPLIT_SCALE);
        } else {
            out[t] = in.base[pix * in.ldc + in.coff + c];
        }
    }
}

int launch_view_to_nchw(const View& in, int B, float* out, hipStream_t s) {
    if (!in.base || !out) { set_error("view_to_nchw: null"); return RTOD_E_ARG; }
    const int64_t total = (int64_t)B * in.C * in.H * in.W;
    hipLaunchKernelGGL(view_to_nchw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, B, out);
    return hip_fail(hipGetLastError(), "view_to_nchw launch");
}

// ---------------------------------------------------------------------------------------------
// stand-alone head decode = predict_transform (src/util.py:175-239) over a strided raw tensor.
// out row r = (gy*G + gx)*A + a, attribute c  <->  raw channel a*attrs + c at cell (gy, gx).
__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ void decode_kernel(const float* __restrict__ raw, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                              int B, DecodeArgs d, float* __restrict__ out) {
    const int AC = d.n_anchors * d.attrs;
    const int64_t per_img = (int64_t)d.G * d.G * AC;
    const int64_t total = (int64_t)B * per_img;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / per_img;
        const int64_t r = t - b * per_img;
        const int cell = (int)(r / AC);
        const int n = (int)(r - (int64_t)cell * AC);
        const int gy = cell / d.G, gx = cell - gy * d.G;
        const int a = n / d.attrs, c = n - a * d.attrs;
        float v = raw[b * sb + (int64_t)n * sc + (int64_t)gy * sy + (int64_t)gx * sx];
        if (c >= 4) v = sigm(v);
        else if (d.v5) {                                 // YOLOv5-style head (cfg extension)
            const float sg = sigm(v);
            if (c < 2) v = ((sg * 2.0f - 0.5f) + (float)(c == 0 ? gx : gy)) * d.stride;
            else { const float t2 = sg * 2.0f; v = (t2 * t2) * (c == 2 ? d.aw[a] : d.ah[a]); }
        }
        else if (c < 2) { v = sigm(v); if (!d.train) v = (v + (float)(c == 0 ? gx : gy)) * d.stride; }
        else if (!d.train) v = (expf(v) * (c == 2 ? d.aw[a] : d.ah[a])) * d.stride;
        out[b * d.img_stride + d.head_off + r] = v;
    }
}

int launch_decode(const float* raw, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B,
                  const DecodeArgs& d, float* out, hipStream_t s) {
    if (!raw || !out || d.G <= 0 || d.attrs < 5 || d.n_anchors < 1 || d.n_anchors > 4) { set_error("decode: bad args"); return RTOD_E_ARG; }
    const int64_t total = (int64_t)B * d.G * d.G * d.n_anchors * d.attrs;
    hipLaunchKernelGGL(decode_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, raw, sb, sc, sy, sx, B, d, out);
    return hip_fail(hipGetLastError(), "decode launch");
}

// ---------------------------------------------------------------------------------------------
// confidence_mask (src/util.py:106-117): out = pred * float(pred[...,4] > conf)
__global__ void confidence_mask_kernel(const float* __restrict__ pred, int64_t rows, int attrs, float conf, float* __restrict__ out) {
    const int64_t total = rows * attrs;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / attrs;
        const float m = pred[r * attrs + 4] > conf ? 1.0f : 0.0f;
        out[t] = pred[t] * m;
    }
}

int launch_confidence_mask(const float* pred, int64_t rows, int attrs, float conf, float* out, hipStream_t s) {
    if (!pred || !out || attrs < 5) { set_error("confidence_mask: bad args"); return RTOD_E_ARG; }
    hipLaunchKernelGGL(confidence_mask_kernel, dim3(grid_for(rows * attrs, 256)), dim3(256), 0, s, pred, rows, attrs, conf, out);
    return hip_fail(hipGetLastError(), "confidence_mask launch");
}

}  // namespace rtod
