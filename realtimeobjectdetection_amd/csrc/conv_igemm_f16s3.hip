// Implicit-GEMM convolution on split-precision f16 MFMA (v_mfma_f32_32x32x16_f16), gfx950.
//
// Same contract, GEMM view and epilogues as conv_igemm_f32.hip (reference: conv -> BN(eval) -> leaky,
// src/darknet.py:467-501; shortcut 263-268; head decode src/util.py:193-237), but the contraction
// runs at the f16 matrix rate (16x the exact-fp32 MFMA per product) with fp32-class accuracy:
//
//   x = xh + xl  (xh = f16(x), xl = f16(x - xh): 22 significant bits), same for the weights
//   a*w ~= ah*wh + ah*wl + al*wh      three MFMA products, fp32 accumulate; al*wl ~ 2^-22 dropped
//
// Data format ("split" activations): every activation tensor of a precision-1 plan is NHWC with each
// pixel stored as [ldc halves: hi plane][ldc halves: lo plane] of the value PRE-SCALED by
// SPLIT_SCALE = 8 (power of two, keeps low parts of small activations out of the f16 subnormal
// range; |activation| must stay < 8188).  Same bytes per pixel as fp32.  Producers (this kernel's
// epilogue, the stem conv, upsample) write it, so staging an A tile is a pure 16-byte copy exactly
// like the pre-split weight planes: no conversion VALU in the main loop.
// Weights: [Npad][Kpad] hi and lo planes, per-output-channel power-of-two pre-scale undone (exactly)
// in the epilogue through inv_scale[n].
//
// Main loop: BMxBNx32 per stage; LDS double-buffered, four 64-byte-row f16 panels per stage with a
// 16-byte-chunk XOR swizzle (chunk ^= (row>>2)&3: conflict-free ds_read_b128 / ds_write_b128);
// two register stage sets keep two K-chunks of global loads in flight; one barrier per K-chunk.
// Addressing: the tap (ky,kx) and channel offset of a K-chunk are wave-uniform (Cin % 32 == 0) and
// advanced in scalar registers; loads are raw buffer loads whose per-lane voffset is
// pixel-origin + tap offset, forced out of range (-> zeros) for padding taps and rows >= M.
#include "rtod_internal.h"

namespace rtod {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HBK = 32;                  // K elements per LDS stage (64 bytes per panel row)
constexpr unsigned OOB = 0x80000000u;    // voffset beyond any buffer (< 2 GiB enforced on the host)

enum { EPI_SPLIT = 0, EPI_SPLIT_RES = 1, EPI_DECODE = 2 };

__device__ __forceinline__ float h_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float h_decode(const DecodeArgs& d, float v, int n, int gx, int gy) {
    const int a = n / d.attrs;
    const int c = n - a * d.attrs;
    if (c >= 4) return h_sigmoid(v);
    if (c < 2) {
        float s = h_sigmoid(v);
        if (d.train) return s;
        return (s + (float)(c == 0 ? gx : gy)) * d.stride;
    }
    if (d.train) return v;
    const float anc = (c == 2) ? d.aw[a] : d.ah[a];
    return (expf(v) * anc) * d.stride;
}

template <int ASL, int BSL>
struct StageRegs {
    u32x4 ah[ASL], al[ASL], bh[BSL], bl[BSL];
};

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, 2)      // <= 256 registers: two workgroups per CU
void conv_igemm_f16s3_kernel(const ConvArgs a, const int grid_m, const int grid_n) {
    constexpr int NWN = BN / WN;
    constexpr int NT = (BM / WM) * NWN * 64;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = NT / 4;                    // rows per pass: 4 x 16-B chunks per 64-B row
    constexpr int A_SLOTS = BM / RPP, B_SLOTS = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/threads mismatch");
    constexpr int PANEL_A = BM * 64, PANEL_B = BN * 64;        // bytes
    constexpr int STAGE = 2 * PANEL_A + 2 * PANEL_B;

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int nwg = grid_m * grid_n;
    int bid = blockIdx.x;
    {   // XCD-aware remap (bijective): blocks sharing an A row-panel run on one XCD / L2
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int bm = bid / grid_n, bn = bid - bm * grid_n;

    const int tid = threadIdx.x;
    const int M = a.B * a.Ho * a.Wo;
    const int c16 = tid & 3, row0 = tid >> 2;
    const unsigned PS = (unsigned)a.in_ldc * 4u;                 // bytes per pixel (hi plane + lo plane)
    const unsigned lo_plane = (unsigned)a.in_ldc * 2u;

    // ---- A: per-slot pixel origin (receptive-field corner), byte offset may be "negative" (wraps)
    int iy0[A_SLOTS], ix0[A_SLOTS];
    unsigned pbase[A_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int m = bm * BM + row0 + i * RPP;
        if (m < M) {
            const int hw = a.Ho * a.Wo;
            const int b = m / hw, r = m - b * hw;
            const int oy = r / a.Wo, ox = r - oy * a.Wo;
            iy0[i] = oy * a.stride - a.pad;
            ix0[i] = ox * a.stride - a.pad;
            pbase[i] = (unsigned)((b * a.Hi + iy0[i]) * a.Wi + ix0[i]) * PS + (unsigned)(a.in_coff + c16 * 8) * 2u;
        } else {
            iy0[i] = -(1 << 28); ix0[i] = 0; pbase[i] = 0;
        }
    }
    // ---- B: per-slot row offset in the weight planes
    unsigned wbase[B_SLOTS];
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) wbase[i] = (unsigned)((bn * BN + row0 + i * RPP) * a.Kpad + c16 * 8) * 2u;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);

    // wave-uniform K-chunk cursor: tap (ky,kx) and first channel c0 of the chunk to be LOADED next
    int ld_kc = 0, ld_c0 = 0, ld_ky = 0, ld_kx = 0;
    const int nk = a.Kpad / HBK;

    StageRegs<A_SLOTS, B_SLOTS> S0, S1;
    auto gload = [&](StageRegs<A_SLOTS, B_SLOTS>& S) {
        const unsigned tap_off = (unsigned)(ld_ky * a.Wi + ld_kx) * PS + (unsigned)ld_c0 * 2u;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const bool ok = (unsigned)(iy0[i] + ld_ky) < (unsigned)a.Hi && (unsigned)(ix0[i] + ld_kx) < (unsigned)a.Wi;
            const unsigned vo = ok ? pbase[i] + tap_off : OOB;
            S.ah[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo, 0, 0);
            S.al[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo, lo_plane, 0);
        }
        const unsigned koff = (unsigned)ld_kc * (HBK * 2);
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            S.bh[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wh, wbase[i], koff, 0);
            S.bl[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wl, wbase[i], koff, 0);
        }
        // advance the cursor (scalar)
        ++ld_kc;
        ld_c0 += HBK;
        if (ld_c0 >= a.Cin) { ld_c0 = 0; if (++ld_kx == a.kw) { ld_kx = 0; ++ld_ky; } }
    };
    // LDS image: panel row r, 16-B chunk c at byte r*64 + ((c ^ ((r>>2)&3)) << 4)
    const int wr_swz = (c16 ^ ((row0 >> 2) & 3)) << 4;           // RPP % 16 == 0 -> same swizzle for every slot
    auto lds_write = [&](const StageRegs<A_SLOTS, B_SLOTS>& S, int buf) {
        unsigned char* st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int o = (row0 + i * RPP) * 64 + wr_swz;
            *reinterpret_cast<u32x4*>(st + o) = S.ah[i];
            *reinterpret_cast<u32x4*>(st + PANEL_A + o) = S.al[i];
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int o = (row0 + i * RPP) * 64 + wr_swz;
            *reinterpret_cast<u32x4*>(st + 2 * PANEL_A + o) = S.bh[i];
            *reinterpret_cast<u32x4*>(st + 2 * PANEL_A + PANEL_B + o) = S.bl[i];
        }
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
    const int rd_swz = (lr >> 2) & 3;
    const int a_row = (wm * WM + lr) * 64, b_row = (wn * WN + lr) * 64;

    auto compute = [&](int buf) {
        const unsigned char* st = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < HBK / 16; ++ks) {
            const int co = ((ks * 2 + lh) ^ rd_swz) << 4;
            f16x8 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fah[i] = *reinterpret_cast<const f16x8*>(st + a_row + i * 32 * 64 + co);
                fal[i] = *reinterpret_cast<const f16x8*>(st + PANEL_A + a_row + i * 32 * 64 + co);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fbh[j] = *reinterpret_cast<const f16x8*>(st + 2 * PANEL_A + b_row + j * 32 * 64 + co);
                fbl[j] = *reinterpret_cast<const f16x8*>(st + 2 * PANEL_A + PANEL_B + b_row + j * 32 * 64 + co);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // prologue: chunks 0 and 1 in flight, chunk 0 staged
    gload(S0);
    if (nk > 1) gload(S1);
    lds_write(S0, 0);
    if (nk > 2) gload(S0);
    __syncthreads();
    for (int t = 0; t < nk; t += 2) {
        compute(0);                                   // chunk t
        if (t + 1 < nk) {
            lds_write(S1, 1);                         // chunk t+1 (waits only for S1's loads)
            if (t + 3 < nk) gload(S1);                // chunk t+3
        }
        __syncthreads();
        if (t + 1 >= nk) break;
        compute(1);                                   // chunk t+1
        if (t + 2 < nk) {
            lds_write(S0, 0);                         // chunk t+2
            if (t + 4 < nk) gload(S0);                // chunk t+4
        }
        __syncthreads();
    }

    // ---- epilogue: D col = lane&31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel)
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = bn * BN + wn * WN + j * 32 + lr;
        if (n >= a.Cout) continue;
        if constexpr (EPI == EPI_DECODE) {
            const float bias = a.bias[n], inv = a.inv_scale[n];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = bm * BM + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (m >= M) continue;
                    float v = acc[i][j][e] * inv + bias;
                    if (a.leaky) v = v > 0.f ? v : v * 0.1f;
                    const int b = m / hw, cell = m - b * hw;
                    const int gy = cell / a.dec.G, gx = cell - gy * a.dec.G;
                    a.out[(int64_t)b * a.dec.img_stride + a.dec.head_off + (int64_t)cell * a.Cout + n] = h_decode(a.dec, v, n, gx, gy);
                }
        } else {
            // everything in the SPLIT_SCALE domain: (acc*inv + bias)*8 == acc*(8 inv) + 8 bias exactly
            const float bias = a.bias[n] * SPLIT_SCALE, inv = a.inv_scale[n] * SPLIT_SCALE;
            _Float16* oh = reinterpret_cast<_Float16*>(a.out) + a.out_coff + n;
            const int64_t ops = 2 * a.out_ldc, olo = a.out_ldc;                 // halves per pixel, lo-plane offset
            const _Float16* rh = reinterpret_cast<const _Float16*>(a.res) + a.res_coff + n;
            const int64_t rps = 2 * a.res_ldc, rlo = a.res_ldc;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float r[16];
                if constexpr (EPI == EPI_SPLIT_RES) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {                                // all residual loads first, one wait
                        const int m = bm * BM + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const int64_t mm = m < M ? m : 0;
                        r[e] = (float)rh[mm * rps] + (float)rh[mm * rps + rlo];
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = bm * BM + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (m >= M) continue;
                    float v = acc[i][j][e] * inv + bias;
                    if (a.leaky) v = v > 0.f ? v : v * 0.1f;
                    if constexpr (EPI == EPI_SPLIT_RES) v += r[e];
                    const _Float16 h = (_Float16)v;
                    oh[(int64_t)m * ops] = h;
                    oh[(int64_t)m * ops + olo] = (_Float16)(v - (float)h);
                }
            }
        }
    }
}

static const ConvVariantInfo kHVariants[HV_COUNT] = {
    {128, 128, "conv_igemm_f16s3<128x128,w64x64>"},
    {128, 64, "conv_igemm_f16s3<128x64,w64x32>"},
    {64, 64, "conv_igemm_f16s3<64x64,w32x32>"},
    {64, 128, "conv_igemm_f16s3<64x128,w32x64>"},
};

const ConvVariantInfo& conv_f16s3_variant_info(int v) { return kHVariants[v < 0 || v >= HV_COUNT ? 0 : v]; }

template <int BM, int BN, int WM, int WN>
static int launch_h(const ConvArgs& a, hipStream_t s) {
    const int M = a.B * a.Ho * a.Wo;
    const int gm = (M + BM - 1) / BM, gn = (a.Cout + BN - 1) / BN;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    if (a.dec.enabled)
        hipLaunchKernelGGL((conv_igemm_f16s3_kernel<BM, BN, WM, WN, EPI_DECODE>), dim3(gm * gn), dim3(NT), 0, s, a, gm, gn);
    else if (a.res)
        hipLaunchKernelGGL((conv_igemm_f16s3_kernel<BM, BN, WM, WN, EPI_SPLIT_RES>), dim3(gm * gn), dim3(NT), 0, s, a, gm, gn);
    else
        hipLaunchKernelGGL((conv_igemm_f16s3_kernel<BM, BN, WM, WN, EPI_SPLIT>), dim3(gm * gn), dim3(NT), 0, s, a, gm, gn);
    return hip_fail(hipGetLastError(), "conv_igemm_f16s3 launch");
}

int launch_conv_f16s3(const ConvArgs& a, int variant, hipStream_t s) {
    if (!a.in || !a.w_hi || !a.w_lo || !a.bias || !a.inv_scale || !a.out) { set_error("launch_conv_f16s3: null pointer"); return RTOD_E_ARG; }
    if (a.Cin % HBK || a.in_ldc % 8 || a.in_coff % 8 || a.Kpad % HBK || a.K != a.Kpad || a.K != a.kh * a.kw * a.Cin) {
        set_error("launch_conv_f16s3: needs Cin %% 32 == 0 and 8-channel aligned views (Cin=%d ldc=%ld coff=%d K=%d Kpad=%d)", a.Cin, (long)a.in_ldc, a.in_coff, a.K, a.Kpad);
        return RTOD_E_ARG;
    }
    if (!a.dec.enabled && (a.out_ldc % 1 || a.out_ldc <= 0)) { set_error("launch_conv_f16s3: bad output view"); return RTOD_E_ARG; }
    if (a.B <= 0 || a.Ho <= 0 || a.Wo <= 0 || a.Cout <= 0) { set_error("launch_conv_f16s3: empty shape"); return RTOD_E_ARG; }
    if (a.in_bytes == 0 || a.in_bytes >= OOB || a.w_bytes == 0 || a.w_bytes >= OOB) {
        set_error("launch_conv_f16s3: buffer of %u / %u bytes outside (0, 2 GiB)", a.in_bytes, a.w_bytes); return RTOD_E_ARG;
    }
    if ((uint64_t)a.B * a.Hi * a.Wi * a.in_ldc * 4ull > (uint64_t)a.in_bytes) { set_error("launch_conv_f16s3: input view exceeds its buffer"); return RTOD_E_ARG; }
    switch (variant) {
        case HV_128x128: return launch_h<128, 128, 64, 64>(a, s);
        case HV_128x64: return launch_h<128, 64, 64, 32>(a, s);
        case HV_64x64: return launch_h<64, 64, 32, 32>(a, s);
        case HV_64x128: return launch_h<64, 128, 32, 64>(a, s);
    }
    set_error("launch_conv_f16s3: unknown variant %d", variant);
    return RTOD_E_ARG;
}

}  // namespace rtod
