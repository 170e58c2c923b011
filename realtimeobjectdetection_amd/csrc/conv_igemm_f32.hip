// Implicit-GEMM convolution on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces the reference's conv2d -> batch_norm(eval) -> leaky_relu_ module (src/darknet.py:467-501,
// run at src/darknet.py:292-295) with BN folded into the packed weights, plus the shortcut add
// (src/darknet.py:263-268) and the YOLO head decode (src/util.py:193-237) as optional epilogues.
//
// GEMM view:  D[m][n] = sum_k A[m][k] * Bw[n][k]
//   m = (b*Ho + oy)*Wo + ox            output pixel          (rows: MFMA A operand)
//   n = output channel                                       (cols: MFMA B operand, on the lane)
//   k = (ky*kw + kx)*Cin + c           NHWC im2col, channels contiguous
// A is gathered on the fly from the NHWC input view (zero padding by predication), Bw is the
// pre-packed [Npad][Kpad] K-major weight panel.  Both are staged through LDS as [row][32+4]
// float tiles (pad of one 16-B slot => conflict-free ds_read_b128 / ds_write_b128), register
// prefetch of the next K-chunk overlaps the MFMAs of the current one.
//
// Numerics: the f32 MFMA is a k-ordered fmaf chain (bit-exact fp32, no reduced precision).  Inside
// an 8-wide k group the two lane halves take k = {0..3} and {4..7} so one ds_read_b128 feeds four
// MFMAs; the chain order is a fixed permutation of k.
//
// K slices (MODE 1 / 2).  The deep small-grid layers (13x13 / 26x26 with K = 1024 ... 4608: YOLOv3-tiny at batch 1 is 3 x 16
// tiles of 64 x 64 on 256 CUs, each walking 144 K chunks at one load round trip per chunk) are summed in SLICES of
// `slice_chunks` K chunks: every slice starts from a zero accumulator and the slice sums are added in ascending order.
// That order is a property of the layer (plan.cpp), so two schedules give the same bits: MODE 1 runs the slices one after
// the other inside the workgroup (large batches: the grid fills the chip anyway), MODE 2 gives every slice its own
// workgroup (grid x slices), which writes its raw sum to a scratch panel, and conv_slice_reduce_kernel adds the panels in
// the same order and applies the epilogue.  Frame independence and variant agreement hold across the two.
#include "rtod_internal.h"
#include <cstdio>

namespace rtod {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;             // floats of K per LDS stage
constexpr int LDS_LD = BK + 4;     // padded row (floats)

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float decode_value(const DecodeArgs& d, float v, int n, int gx, int gy) {
    const int a = n / d.attrs;
    const int c = n - a * d.attrs;
    if (c >= 4) return sigmoidf_(v);
    if (d.v5) {                                     // YOLOv5-style head (cfg extension): operation order of the published Detect layer
        const float s = sigmoidf_(v);
        if (c < 2) return ((s * 2.0f - 0.5f) + (float)(c == 0 ? gx : gy)) * d.stride;
        const float t = s * 2.0f;
        return (t * t) * (c == 2 ? d.aw[a] : d.ah[a]);
    }
    if (c < 2) {
        float s = sigmoidf_(v);
        if (d.train) return s;
        return (s + (float)(c == 0 ? gx : gy)) * d.stride;
    }
    if (d.train) return v;
    const float anc = (c == 2) ? d.aw[a] : d.ah[a];
    return (expf(v) * anc) * d.stride;
}

// Epilogue of one output value (raw K sum -> bias, activation, shortcut, then head decode / split format / plain store).
// Shared by the convolution kernel and the slice reduction so that both apply the very same operations (no contraction:
// the two kernels must agree bit for bit).
__device__ __forceinline__ void conv_f32_store(const ConvArgs& a, int m, int n, float raw, int hw, float& amax) {
#pragma clang fp contract(off)
    float v = raw + a.bias[n];
    v = apply_act(v, a.leaky);
    if (a.res) v += a.res[(int64_t)m * a.res_ldc + a.res_coff + n];
    if (a.dec.enabled) {
        const int b = m / hw, cell = m - b * hw;
        const int gy = cell / a.dec.G, gx = cell - gy * a.dec.G;
        v = decode_value(a.dec, v, n, gx, gy);
        a.out[(int64_t)b * a.dec.img_stride + a.dec.head_off + (int64_t)cell * a.Cout + n] = v;
    } else if (a.out_split) {
        // split format for a following conv_igemm_f16s3 layer: 8*v as f16 hi + f16 lo planes
        _Float16 h, l;
        split_f16(v * SPLIT_SCALE, h, l, amax);
        _Float16* oh = reinterpret_cast<_Float16*>(a.out) + (int64_t)m * 2 * a.out_ldc + a.out_coff + n;
        oh[0] = h;
        oh[a.out_ldc] = l;
    } else {
        a.out[(int64_t)m * a.out_ldc + a.out_coff + n] = v;
    }
}

// MODE 0: one accumulation chain over K.  MODE 1: K slices summed in order inside the workgroup.  MODE 2: one workgroup per
// (tile, slice): raw slice sums to a.partial [slice][M][Npad].
template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64)
void conv_igemm_f32_kernel(const ConvArgs a, const int grid_m, const int grid_n) {
    constexpr int NWN = BN / WN;
    constexpr int NT = (BM / WM) * NWN * 64;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ROWS_PER_PASS = NT / 8;          // 8 float4 slots per 32-float row
    constexpr int A_SLOTS = BM / ROWS_PER_PASS;
    constexpr int B_SLOTS = BN / ROWS_PER_PASS;
    static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/threads mismatch");

    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_LD];
    float* sA = smem;
    float* sB = smem + BM * LDS_LD;

    // XCD-aware block remap (cdna guide T1, bijective form): blocks that share an A row-panel
    // (same bm) are made consecutive within one XCD so the panel is fetched into one L2.
    const int nwg = grid_m * grid_n;
    int bid = blockIdx.x;
    int slice = 0;
    if constexpr (MODE == 2) { slice = bid / nwg; bid -= slice * nwg; }
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int bm = bid / grid_n, bn = bid - bm * grid_n;

    const int tid = threadIdx.x;
    const int k4 = tid & 7;               // float4 slot inside the 32-float K chunk
    const int row0 = tid >> 3;
    const int M = a.B * a.Ho * a.Wo;

    // per-thread A rows: input-pixel origin of the receptive field
    int iy0[A_SLOTS], ix0[A_SLOTS], pb[A_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int m = bm * BM + row0 + i * ROWS_PER_PASS;
        if (m < M) {
            const int hw = a.Ho * a.Wo;
            const int b = m / hw, r = m - b * hw;
            const int oy = r / a.Wo, ox = r - oy * a.Wo;
            iy0[i] = oy * a.stride - a.pad;
            ix0[i] = ox * a.stride - a.pad;
            pb[i] = b * a.Hi * a.Wi;
        } else {
            iy0[i] = -(1 << 28); ix0[i] = 0; pb[i] = 0;     // never in bounds -> zeros
        }
    }
    const float* wrow = a.w + (int64_t)(bn * BN + row0) * a.Kpad + k4 * 4;
    const float* inb = a.in + a.in_coff;

    // One register stage: chunk kc + 1 is fetched while chunk kc computes.  (A second stage — chunk kc + 2 in flight — was
    // measured slower on every shape, YOLOv3 608x608 batch 8 -2.8 %, YOLOv3-tiny batch 1 -6 %: 30 more VGPRs cost more
    // occupancy than the longer prefetch distance hides.)
    f32x4 ra0[A_SLOTS], rb0[B_SLOTS];
    auto gload = [&](int kc, f32x4 (&ra)[A_SLOTS], f32x4 (&rb)[B_SLOTS]) {
        const int k = kc * BK + k4 * 4;
        const bool kok = k < a.K;
        const int tap = kok ? k / a.Cin : 0;
        const int c = k - tap * a.Cin;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = kok && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(inb + (int64_t)(pb[i] + iy * a.Wi + ix) * a.in_ldc + c);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i)
            rb[i] = *reinterpret_cast<const f32x4*>(wrow + (int64_t)i * ROWS_PER_PASS * a.Kpad + kc * BK);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / NWN, wn = wave - wm * NWN;
    const int lr = lane & 31, lh = lane >> 5;
    const float* pa = sA + (wm * WM + lr) * LDS_LD + lh * 4;
    const float* pbw = sB + (wn * WN + lr) * LDS_LD + lh * 4;

    const int nk = a.Kpad / BK;
    const int kc0 = MODE == 2 ? slice * a.slice_chunks : 0;
    const int kc1 = MODE == 2 ? (kc0 + a.slice_chunks < nk ? kc0 + a.slice_chunks : nk) : nk;
    f32x16 tot[MODE == 1 ? TM : 1][MODE == 1 ? TN : 1];
    int in_slice = 0;                                  // MODE 1: chunks accumulated in the current slice
    auto step = [&](int kc, f32x4 (&ra)[A_SLOTS], f32x4 (&rb)[B_SLOTS]) {
        __syncthreads();                           // previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i)
            *reinterpret_cast<f32x4*>(sA + (row0 + i * ROWS_PER_PASS) * LDS_LD + k4 * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i)
            *reinterpret_cast<f32x4*>(sB + (row0 + i * ROWS_PER_PASS) * LDS_LD + k4 * 4) = rb[i];
        __syncthreads();
        if (kc + 1 < kc1) gload(kc + 1, ra, rb);   // in flight while the MFMAs below run
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(pa + i * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(pbw + j * 32 * LDS_LD + q * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if constexpr (MODE == 1) {
            if (++in_slice == a.slice_chunks || kc + 1 == kc1) {     // slice boundary (uniform): total (+)= slice sum
                const bool first = kc < a.slice_chunks;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            tot[i][j][e] = first ? acc[i][j][e] : tot[i][j][e] + acc[i][j][e];
                            acc[i][j][e] = 0.f;
                        }
                in_slice = 0;
            }
        }
    };
    gload(kc0, ra0, rb0);
    for (int kc = kc0; kc < kc1; ++kc) step(kc, ra0, rb0);
    if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j];
    }
    if constexpr (MODE == 2) {                         // raw slice sums -> scratch panel [slice][M][Npad]
        float* part = a.partial + (int64_t)slice * M * a.Npad;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = bn * BN + wn * WN + j * 32 + lr;
            if (n >= a.Npad) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = bm * BM + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (m < M) part[(int64_t)m * a.Npad + n] = acc[i][j][e];
                }
        }
        return;
    }

    // epilogue: D col = lane&31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel)
    const int hw = a.Ho * a.Wo;
    float amax = 0.f;                                  // overflow sentinel when writing the split format
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = bn * BN + wn * WN + j * 32 + lr;
        if (n >= a.Cout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm * BM + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m >= M) continue;
                conv_f32_store(a, m, n, acc[i][j][e], hw, amax);
            }
        }
    }
    if (a.out_split) split_overflow_report(a.ovf, amax);
}

// Sum of the slice panels in ascending slice order + epilogue (MODE 2's second half).  One thread per (pixel, channel),
// channels fastest: coalesced panel reads.
__global__ __launch_bounds__(256)
void conv_slice_reduce_kernel(const ConvArgs a, const int n_slices, const int M) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int m = (int)(idx / a.Npad), n = (int)(idx - (int64_t)m * a.Npad);
    float amax = 0.f;
    if (m < M && n < a.Cout) {
        const float* p = a.partial + (int64_t)m * a.Npad + n;
        const int64_t panel = (int64_t)M * a.Npad;
        float v = p[0];
        for (int s = 1; s < n_slices; ++s) v = v + p[s * panel];
        conv_f32_store(a, m, n, v, a.Ho * a.Wo, amax);
    }
    if (a.out_split) split_overflow_report(a.ovf, amax);
}

static const ConvVariantInfo kVariants[CV_COUNT] = {
    {128, 128, "conv_igemm_f32<128x128,w64x64>"},
    {128, 64, "conv_igemm_f32<128x64,w64x32>"},
    {64, 64, "conv_igemm_f32<64x64,w32x32>"},
    {128, 32, "conv_igemm_f32<128x32,w32x32>"},
};

const ConvVariantInfo& conv_variant_info(int v) { return kVariants[v < 0 || v >= CV_COUNT ? 0 : v]; }

// variant ids of the exact-fp32 kernel: tile (0-3) + 10 * mode (0 plain, 1 K slices in the workgroup, 2 one workgroup per slice)
int conv_f32_kernel_name(int variant, char* buf, size_t len) {          // demangled instantiation name (rocprofv3)
    static const int t[CV_COUNT][4] = {{128, 128, 64, 64}, {128, 64, 64, 32}, {64, 64, 32, 32}, {128, 32, 32, 32}};   // launch_conv's template arguments
    const int tile = variant % 10, mode = variant / 10;
    if (variant < 0 || tile >= CV_COUNT || mode > 2) return -1;
    return snprintf(buf, len, "void rtod::conv_igemm_f32_kernel<%d, %d, %d, %d, %d>(rtod::ConvArgs, int, int)", t[tile][0], t[tile][1], t[tile][2], t[tile][3], mode);
}

int conv_f32_slices(const ConvArgs& a) { return a.slice_chunks > 0 ? (a.Kpad / BK + a.slice_chunks - 1) / a.slice_chunks : 1; }

template <int BM, int BN, int WM, int WN>
static int launch_t(const ConvArgs& a, hipStream_t s) {
    const int M = a.B * a.Ho * a.Wo;
    const int gm = (M + BM - 1) / BM, gn = (a.Cout + BN - 1) / BN;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    if (a.slice_chunks <= 0) hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, 0>), dim3(gm * gn), dim3(NT), 0, s, a, gm, gn);
    else if (!a.partial) hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, 1>), dim3(gm * gn), dim3(NT), 0, s, a, gm, gn);
    else {
        const int S = conv_f32_slices(a);
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, 2>), dim3(gm * gn * S), dim3(NT), 0, s, a, gm, gn);
        if (hipGetLastError() != hipSuccess) return hip_fail(hipGetLastError(), "conv_igemm_f32 slice launch");
        const int64_t n = (int64_t)M * a.Npad;
        hipLaunchKernelGGL(conv_slice_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, S, M);
    }
    return hip_fail(hipGetLastError(), "conv_igemm_f32 launch");
}

int launch_conv(const ConvArgs& a, int variant, hipStream_t s) {
    // host-side shape checks: the kernel assumes all of these (an out-of-bounds access on this
    // pool can reset the whole node)
    if (!a.in || !a.w || !a.bias || !a.out) { set_error("launch_conv: null pointer"); return RTOD_E_ARG; }
    if (a.Cin % 4 || a.in_ldc % 4 || a.in_coff % 4 || a.Kpad % BK || a.K > a.Kpad || a.K != a.kh * a.kw * a.Cin) {
        set_error("launch_conv: bad channel alignment (Cin=%d ldc=%ld coff=%d K=%d Kpad=%d)", a.Cin, (long)a.in_ldc, a.in_coff, a.K, a.Kpad);
        return RTOD_E_ARG;
    }
    if (a.B <= 0 || a.Ho <= 0 || a.Wo <= 0 || a.Cout <= 0) { set_error("launch_conv: empty shape"); return RTOD_E_ARG; }
    if ((int64_t)a.B * a.Hi * a.Wi >= (1ll << 31) || (int64_t)a.B * a.Ho * a.Wo >= (1ll << 31)) {
        set_error("launch_conv: pixel count exceeds int32"); return RTOD_E_ARG;
    }
    if (a.slice_chunks < 0 || (a.partial && a.slice_chunks <= 0) || a.Npad < a.Cout || a.Npad % 32) { set_error("launch_conv: bad K-slice arguments"); return RTOD_E_ARG; }
    if (a.partial && (int64_t)conv_f32_slices(a) * a.B * a.Ho * a.Wo * a.Npad > a.partial_floats) { set_error("launch_conv: slice scratch too small"); return RTOD_E_ARG; }
    switch (variant) {
        case CV_128x128: return launch_t<128, 128, 64, 64>(a, s);
        case CV_128x64: return launch_t<128, 64, 64, 32>(a, s);
        case CV_64x64: return launch_t<64, 64, 32, 32>(a, s);
        case CV_128x32: return launch_t<128, 32, 32, 32>(a, s);
    }
    set_error("launch_conv: unknown variant %d", variant);
    return RTOD_E_ARG;
}

}  // namespace rtod
