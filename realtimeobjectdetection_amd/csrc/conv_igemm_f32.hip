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
    if (c < 2) {
        float s = sigmoidf_(v);
        if (d.train) return s;
        return (s + (float)(c == 0 ? gx : gy)) * d.stride;
    }
    if (d.train) return v;
    const float anc = (c == 2) ? d.aw[a] : d.ah[a];
    return (expf(v) * anc) * d.stride;
}

template <int BM, int BN, int WM, int WN>
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

    f32x4 ra[A_SLOTS], rb[B_SLOTS];
    auto gload = [&](int kc) {
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
    gload(0);
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();                           // previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i)
            *reinterpret_cast<f32x4*>(sA + (row0 + i * ROWS_PER_PASS) * LDS_LD + k4 * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i)
            *reinterpret_cast<f32x4*>(sB + (row0 + i * ROWS_PER_PASS) * LDS_LD + k4 * 4) = rb[i];
        __syncthreads();
        if (kc + 1 < nk) gload(kc + 1);            // in flight while the MFMAs below run
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
    }

    // epilogue: D col = lane&31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel)
    const int hw = a.Ho * a.Wo;
    float amax = 0.f;                                  // overflow sentinel when writing the split format
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = bn * BN + wn * WN + j * 32 + lr;
        if (n >= a.Cout) continue;
        const float bias = a.bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm * BM + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m >= M) continue;
                float v = acc[i][j][e] + bias;
                if (a.leaky) v = v > 0.f ? v : v * 0.1f;
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
        }
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

int conv_f32_kernel_name(int variant, char* buf, size_t len) {          // demangled instantiation name (rocprofv3)
    static const int t[CV_COUNT][4] = {{128, 128, 64, 64}, {128, 64, 64, 32}, {64, 64, 32, 32}, {128, 32, 32, 32}};   // launch_conv's template arguments
    if (variant < 0 || variant >= CV_COUNT) return -1;
    return snprintf(buf, len, "void rtod::conv_igemm_f32_kernel<%d, %d, %d, %d>(rtod::ConvArgs, int, int)", t[variant][0], t[variant][1], t[variant][2], t[variant][3]);
}

template <int BM, int BN, int WM, int WN>
static int launch_t(const ConvArgs& a, hipStream_t s) {
    const int M = a.B * a.Ho * a.Wo;
    const int gm = (M + BM - 1) / BM, gn = (a.Cout + BN - 1) / BN;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN>), dim3(gm * gn), dim3(NT), 0, s, a, gm, gn);
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
