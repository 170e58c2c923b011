// Shared pieces of the split-precision f16 convolution kernels (conv_igemm_f16s3.hip, conv_band_f16s3.hip).
#pragma once
#include "rtod_internal.h"
#include <type_traits>

namespace rtod {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HBK = 32;                  // K elements per LDS stage (64 bytes per panel row)
constexpr unsigned OOB = 0x80000000u;    // voffset beyond any buffer (< 2 GiB enforced on the host)

enum { EPI_SPLIT = 0, EPI_SPLIT_RES = 1, EPI_DECODE = 2, EPI_SPLIT_PW = 3, EPI_SPLIT_RES_PW = 4 };
constexpr int PW_MAX_COUT = 64, PW_MAX_K = 64;    // measured: hosts with 128 output channels gain nothing over the stand-alone 1x1 kernel

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

// Raw buffer load issued through inline asm so that hipcc's s_waitcnt insertion does not see it: the
// main loop keeps two K-chunks of loads in flight across barriers and waits with hand-counted
// vmcnt(N) (cdna guide 5.7: loads hidden from the compiler need their own counted wait, and every
// destination must be named by the wait statement before its first use).
__device__ __forceinline__ u32x4 asm_buffer_load_b128(const __amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
    return v;
}

// 16-byte activation store, sc1 (write-through): a layer's output is read by the NEXT kernel, on all XCDs, so it has
// to reach memory anyway; written back line by line while the kernel runs it leaves the kernel-end release little
// dirty L2 to drain (measured +1.2 % frames/s over plain write-back stores; non-temporal stores: -2.7 %).
// `plain` (diagnostic build, RTOD_DBG_ZERO bit 8) keeps the write-back store for that comparison.
__device__ __forceinline__ void store_act16(_Float16* p, const f16x8& v, bool plain) {
#ifdef RTOD_DIAG
    if (plain) { *reinterpret_cast<f16x8*>(p) = v; return; }
#endif
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

// SiLU on a value in the split format's domain (8 x): 8 * silu(x) = v / (1 + exp(-v / 8)); hardware exp2 / rcp (~1e-6 relative).
// The epilogues select it with a uniform branch AROUND their loops (a per-value select made the compiler evaluate both
// activations for every element: 2.5 % on the YOLOv3 forward, measured).
__device__ __forceinline__ float silu_scaled(float v, float inv_domain) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v * inv_domain)); }

// ---- transposed product (out^T = W * act^T): the MFMA's first operand is the weight fragment, the second the activation
// fragment, so a lane's accumulator holds 4 consecutive output CHANNELS (rows 4*lh + e) of ONE pixel (column lr).  With the
// weight rows of a 32-channel group interleaved over two 16-row tiles (tr_chan_of_row) the two tiles of a pair give the lane
// 8 consecutive channels: one 16-byte store per plane straight from the accumulators — no LDS transpose, no barrier.
// Panel row rho of an N tile (tile t = rho/16, row r = rho%16) <-> channel 32*(t/2) + 8*(r/4) + 4*(t%2) + r%4 of the tile.
__host__ __device__ __forceinline__ int tr_chan_of_row(int rho) {
    const int t = rho >> 4, r = rho & 15;
    return 32 * (t >> 1) + 8 * (r >> 2) + 4 * (t & 1) + (r & 3);
}

// Register epilogue of the transposed product.  acc[i][j]: pixel tile i (16 pixels), channel tile j of the wave.
// mrow[i] = linear output pixel (b*Ho + y)*Wo + x of this lane's column in pixel tile i, or -1 if it lies outside the tensor;
// cbase = first output channel of the wave's channel tiles.
// KG == 2 (in-workgroup split-K): group 1 parks its accumulators in LDS (`smem`, 16 bytes per lane and tile: the MFMA layout
// as it stands), one barrier, group 0 adds them to its own and stores.
template <int WM, int WN, bool RES, int KG = 1>
__device__ __forceinline__ void conv_f16s3_epilogue_regs(const ConvArgs& a, f32x4 (&acc)[WM / 16][WN / 16], unsigned char* smem,
                                                         const int (&mrow)[WM / 16], int cbase, int tidg, int lh, int kg = 0) {
    constexpr int TM = WM / 16, TN = WN / 16, NP = TN / 2;
    static_assert(TN % 2 == 0, "channel tiles come in pairs");
    if constexpr (KG == 2) {
        f32x4* T = reinterpret_cast<f32x4*>(smem);
        const int base = ((tidg >> 6) * TM * TN) * 64 + (tidg & 63);          // [wave of the group][i][j][lane]
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) T[base + (i * TN + j) * 64] = acc[i][j];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 o = T[base + (i * TN + j) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] += o[e];
            }
    }
    float amax = 0.f;
    _Float16* const oh = reinterpret_cast<_Float16*>(a.out) + a.out_coff;
    const _Float16* const rh = reinterpret_cast<const _Float16*>(a.res) + a.res_coff;
#pragma unroll
    for (int P = 0; P < NP; ++P) {
        const int c0 = cbase + 32 * P + 8 * lh;
        const bool cok = c0 < a.Cout;                                        // Cout % 8 == 0 for every split-format tensor
        const int cc = cok ? c0 : 0;
        // residual operands of this channel group first: their latency runs under the scale / bias loads and the conversions
        f16x8 rq_h[RES ? TM : 1], rq_l[RES ? TM : 1];
        if constexpr (RES) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mrow[i];
                const bool ok = m >= 0 && cok;
                const _Float16* q = rh + (int64_t)(ok ? m : 0) * 2 * a.res_ldc + cc;
                rq_h[i] = ok ? *reinterpret_cast<const f16x8*>(q) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};     // (unconditional loads + select, as in the
                rq_l[i] = ok ? *reinterpret_cast<const f16x8*>(q + a.res_ldc) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};   //  LDS epilogue below: +6 % on the patch kernel, measured)
            }
        }
        f32x4 iv0 = *reinterpret_cast<const f32x4*>(a.inv_scale + cc), iv1 = *reinterpret_cast<const f32x4*>(a.inv_scale + cc + 4);
        f32x4 bs0 = *reinterpret_cast<const f32x4*>(a.bias + cc), bs1 = *reinterpret_cast<const f32x4*>(a.bias + cc + 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mrow[i];
            f16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = e < 4 ? acc[i][2 * P][e] : acc[i][2 * P + 1][e - 4];
                // (acc*inv + bias)*8 == acc*(8 inv) + 8 bias exactly (power of two)
                float v = s * ((e < 4 ? iv0[e] : iv1[e - 4]) * SPLIT_SCALE) + (e < 4 ? bs0[e] : bs1[e - 4]) * SPLIT_SCALE;
                if (a.leaky) v = v > 0.f ? v : v * 0.1f;              // linear / leaky only: SiLU layers run on the LDS-transposed epilogue (plan.cpp)
                if constexpr (RES) v += (float)rq_h[i][e] + (float)rq_l[i][e];
                _Float16 h, l;
                split_f16(v, h, l, amax);
                ph[e] = h; pl[e] = l;
            }
            if (m >= 0 && cok) {
                _Float16* q = oh + (int64_t)m * 2 * a.out_ldc + c0;
                store_act16(q, ph, false);
                store_act16(q + a.out_ldc, pl, false);
            }
        }
    }
    split_overflow_report(a.ovf, amax);
}

constexpr int epi_row_group(int bm, int wm, int rg_max) {
    int best = wm;
    for (int r = wm; r <= bm && r <= rg_max; r += wm) if (bm % r == 0) best = r;
    return best;
}


// ---- common epilogue: scale / bias / leaky, LDS transpose, then split-format store (+ residual) or head decode.
// `smem` is the kernel's whole LDS allocation (dead after the main loop, which must end on a barrier).
// Accumulators are v_mfma_f32_16x16x32_f16 tiles: lane holds column lr = lane%16, rows e + 4*lh, lh = lane/16.
// KG = 2 (in-workgroup split-K, conv_band_f16s3.hip): NT counts both wave groups, `tid` is the workgroup-wide thread
// index, wm / wn are positions inside the group `kg`; group 1 deposits its raw accumulators in the tile first and
// group 0 adds its own before scale / bias / activation.
// PRE: the caller hands over bias[n] / inv_scale[n] of its WN / 16 column groups in registers (conv_ring_f16s3.hip loads them at
// the start of a tile: a compiler-visible global load inside the epilogue would be waited for with vmcnt(0), which also
// drains the LDS-DMA ring that is prefetching the next tile).
template <int BM, int BN, int WM, int WN, int NT, int EPI, int SMEM_BYTES, int KG = 1, bool PRE = false>
__device__ __forceinline__ void conv_f16s3_epilogue(const ConvArgs& a, f32x4 (&acc)[WM / 16][WN / 16], unsigned char* smem,
                                                    int bm, int bn, int tid, int wm, int wn, int lr, int lh, int M, int kg = 0,
                                                    const float* pre_bias = nullptr, const float* pre_inv = nullptr) {
    constexpr int MT = 16, TM = WM / MT, TN = WN / MT, NE = 4;
    constexpr bool PW = EPI == EPI_SPLIT_PW || EPI == EPI_SPLIT_RES_PW;
    constexpr bool RES = EPI == EPI_SPLIT_RES || EPI == EPI_SPLIT_RES_PW;
    // fused pointwise conv: the transpose tile doubles as the A operand of the second GEMM (row stride + 4 floats: the
    // 16 rows x 16 bytes of a fragment read then cover all 64 banks), and a second tile T2 takes its result
    constexpr int TS = PW ? BN + 4 : BN;
    constexpr int T2S = PW_MAX_COUT + 4;
    // ---- epilogue.  The accumulators (MFMA layout: channel on the lane, 16 pixel rows per register set)
    // are scaled / biased / activated and transposed through LDS (the stage buffers are dead: the main
    // loop ended on a barrier) as an fp32 [rows][BN] tile, so that the residual loads and the output
    // stores are row-contiguous 16-byte accesses (split format) or 256-byte row segments (decode).
    constexpr int RG_MAX = SMEM_BYTES / ((TS + (PW ? T2S : 0)) * 4);
    constexpr int RG = epi_row_group(BM, WM, RG_MAX);                     // rows per pass: multiple of WM dividing BM
    static_assert(RG >= WM && BM % RG == 0, "epilogue row group");
    float* T = reinterpret_cast<float*>(smem);
    const int hw = a.Ho * a.Wo;
#ifdef RTOD_DIAG
    const bool st_plain = (a.dbg & 8) != 0;
#else
    constexpr bool st_plain = false;
#endif
    float amax = 0.f;                                                    // overflow sentinel of the split-format stores
    // fused pointwise conv: a wave owns one 16-column group of the second GEMM; its B fragments (the whole K of the
    // 1x1 conv, <= 4 k32 steps) are fetched once here, their latency hidden behind the transpose and the first store pass
    constexpr int PWK = PW_MAX_K / 32;
    f16x8 b2h[PW ? PWK : 1], b2l[PW ? PWK : 1];
    float bias2 = 0.f, inv2 = 0.f;
    int pw_n2t = 1, pw_j = 0;
    if constexpr (PW) {
        pw_n2t = a.pw_cout / 16;                                  // 1, 2 or 4: divides the wave count
        pw_j = (tid >> 6) % pw_n2t;
        const int n2 = pw_j * 16 + (tid & 15);
        const _Float16* w2h = a.pw_wh + (int64_t)n2 * 32 + ((tid & 63) >> 4) * 8;      // [chunk][pw_npad][32]
        const _Float16* w2l = a.pw_wl + (int64_t)n2 * 32 + ((tid & 63) >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < PWK; ++ks) {
            const bool ok = ks * 32 < a.pw_k;
            b2h[ks] = ok ? *reinterpret_cast<const f16x8*>(w2h + (int64_t)ks * a.pw_npad * 32) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            b2l[ks] = ok ? *reinterpret_cast<const f16x8*>(w2l + (int64_t)ks * a.pw_npad * 32) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        bias2 = a.pw_bias[n2] * SPLIT_SCALE; inv2 = a.pw_inv_scale[n2] * SPLIT_SCALE;
    }
    const float escale = (EPI == EPI_DECODE) ? 1.0f : SPLIT_SCALE;   // (acc*inv + bias)*8 == acc*(8 inv) + 8 bias exactly
    constexpr int GPR_ = BN / 8;                                          // 8-channel (16-byte) groups per row
    constexpr int NG = (RG * GPR_ + NT - 1) / NT;                         // groups per thread and pass
    // bias / scale of the wave's TN column groups: all loads first and ahead of the residual loads (vmcnt retires in order: the
    // transpose must not wait for those), from a clamped index and selected afterwards — the conditional form
    // `n < Cout ? bias[n] : 0` cost a branch and a full memory wait per value, 2 TN of them in a row
    // (loaded at the top of every pass: held across the passes of a multi-pass epilogue they cost the generic tiles 10-60 registers)
#pragma unroll 1
    for (int rg = 0; rg < BM; rg += RG) {
        float bias_j[TN], inv_j[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if constexpr (PRE) { bias_j[j] = pre_bias[j]; inv_j[j] = pre_inv[j]; }
            else {
                const int n = bn * BN + wn * WN + j * MT + lr;
                const int nc = n < a.Cout ? n : 0;
                const float b = a.bias[nc], iv = a.inv_scale[nc];
                bias_j[j] = n < a.Cout ? b : 0.f; inv_j[j] = n < a.Cout ? iv : 0.f;
            }
        }
        // residual operands of this pass: issued ahead of the transpose so that their latency (HBM / L2, ~2 us when
        // waited for inside the store loop: in-kernel stamps showed the epilogue at 15 % of a 76x76 tile) overlaps the
        // accumulator -> LDS phase and the barrier
        f16x8 rq_h[RES ? NG : 1], rq_l[RES ? NG : 1];
        if constexpr (RES) {
            const _Float16* rh0 = reinterpret_cast<const _Float16*>(a.res) + a.res_coff + bn * BN;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int g = tid + i * NT;
                const int r = g / GPR_, c8 = (g - r * GPR_) * 8;
                const int m = bm * BM + rg + r;
                const bool ok = g < RG * GPR_ && m < M && bn * BN + c8 < a.Cout;
                // unconditional loads from a clamped address, selected afterwards: written as `ok ? *q : 0` each load became a
                // branch with its own s_waitcnt vmcnt(0) — NG pairs of memory latencies in a row at the head of every epilogue
                const _Float16* q = rh0 + (int64_t)(ok ? m : 0) * 2 * a.res_ldc + (ok ? c8 : 0);
                const f16x8 th = *reinterpret_cast<const f16x8*>(q), tl = *reinterpret_cast<const f16x8*>(q + a.res_ldc);
                rq_h[i] = ok ? th : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                rq_l[i] = ok ? tl : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        if constexpr (KG == 2) {
            if (kg == 1 && wm * WM >= rg && wm * WM < rg + RG) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < NE; ++e)
                            T[(wm * WM - rg + i * MT + e + 4 * lh) * TS + wn * WN + j * MT + lr] = acc[i][j][e];
            }
            __syncthreads();
        }
        if ((KG == 1 || kg == 0) && (RG == BM || (wm * WM >= rg && wm * WM < rg + RG))) {     // single pass, one K group: no branch
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = wn * WN + j * MT + lr;
                const float bias = bias_j[j] * escale, inv = inv_j[j] * escale;
                auto col = [&](auto act) {                                   // 0 linear, 1 leaky, 2 SiLU
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < NE; ++e) {
                            const int rl = wm * WM - rg + i * MT + e + 4 * lh;
                            float s = acc[i][j][e];
                            if constexpr (KG == 2) s += T[rl * TS + nl];
                            float v = s * inv + bias;
                            if constexpr (decltype(act)::value == 2) v = silu_scaled(v, 1.0f / escale);
                            else if constexpr (decltype(act)::value == 1) v = __builtin_fmaxf(v, v * 0.1f);   // == v > 0 ? v : 0.1 v, bit for bit (signed zeros, infinities, NaN)
                            T[rl * TS + nl] = v;
                        }
                };
                // uniform three-way branch around the loop: a per-value `if (a.leaky)` was a compare, a mask OR and a select per element
                if (a.leaky == 2) col(std::integral_constant<int, 2>{});
                else if (a.leaky) col(std::integral_constant<int, 1>{});
                else col(std::integral_constant<int, 0>{});
            }
        }
        __syncthreads();
        if constexpr (EPI == EPI_DECODE) {
            // thread <-> fixed channel n: anchor / attribute kind are per-thread constants; rows advance by NT/BN
            constexpr int RSTEP = NT / BN > 0 ? NT / BN : 1;
            static_assert(NT % BN == 0 || BN % NT == 0, "decode mapping");
            for (int nl = tid % BN; nl < BN; nl += NT) {
                const int n = bn * BN + nl;
                if (n >= a.Cout) continue;
                const int an = n / a.dec.attrs, c = n - an * a.dec.attrs;
                const bool is_wh = (c == 2 || c == 3) && !a.dec.train;
                const bool is_raw = (c == 2 || c == 3) && a.dec.train;
                const float anc = (c == 2 ? a.dec.aw[an] : a.dec.ah[an]);
                const int r0 = (NT >= BN) ? tid / BN : 0;
                int m = bm * BM + rg + r0;
                const int b = m / hw;
                int cell = m - b * hw;
                int gy = cell / a.dec.G, gx = cell - gy * a.dec.G;
                // The loop is VALU-bound (1 600 instructions per thread per 128-row tile with libm expf, an IEEE divide and
                // 64-bit index arithmetic per element: the decode epilogue was 58 % of the 76x76 head kernel): hardware
                // exp2 / rcp (1 ulp; the result has to meet 1e-4) and a running output pointer.
                float* op = a.out + (int64_t)b * a.dec.img_stride + a.dec.head_off + (int64_t)cell * a.Cout + n;
                const int64_t row_step = (int64_t)RSTEP * a.Cout;
                const int64_t img_fix = a.dec.img_stride - (int64_t)hw * a.Cout;      // first cell of the next image
                if (a.dec.v5) {                                          // YOLOv5-style head (cfg extension); uniform
                    for (int r = r0; r < RG && m < M; r += RSTEP, m += RSTEP) {
                        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-T[r * TS + nl]));
                        float o = sg;
                        if (c < 2) o = ((sg * 2.0f - 0.5f) + (float)(c == 0 ? gx : gy)) * a.dec.stride;
                        else if (c < 4) { const float t2 = sg * 2.0f; o = (t2 * t2) * anc; }
                        *op = o;
                        op += row_step; cell += RSTEP; gx += RSTEP;
                        while (gx >= a.dec.G) { gx -= a.dec.G; ++gy; }
                        while (cell >= hw) { cell -= hw; gy -= a.dec.G; op += img_fix; }
                    }
                } else
                for (int r = r0; r < RG && m < M; r += RSTEP, m += RSTEP) {
                    const float v = T[r * TS + nl];
                    float o;
                    if (is_raw) o = v;
                    else {
                        const float ex = __expf(is_wh ? v : -v);          // one exp serves both kinds
                        if (is_wh) o = (ex * anc) * a.dec.stride;
                        else {
                            o = __builtin_amdgcn_rcpf(1.0f + ex);
                            if (c < 2 && !a.dec.train) o = (o + (float)(c == 0 ? gx : gy)) * a.dec.stride;
                        }
                    }
                    *op = o;
                    op += row_step; cell += RSTEP; gx += RSTEP;
                    while (gx >= a.dec.G) { gx -= a.dec.G; ++gy; }
                    while (cell >= hw) { cell -= hw; gy -= a.dec.G; op += img_fix; }   // G*G < RSTEP: a step may cross several images
                }
            }
        } else {
            constexpr int GPR = GPR_;
            _Float16* oh = reinterpret_cast<_Float16*>(a.out) + a.out_coff + bn * BN;
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int g = tid + gi * NT;
                if (g >= RG * GPR) continue;
                const int r = g / GPR, c8 = (g - r * GPR) * 8;
                const int m = bm * BM + rg + r;
                if (m >= M || bn * BN + c8 >= a.Cout) continue;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(T + r * TS + c8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(T + r * TS + c8 + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if constexpr (RES) {
                    const f16x8 qh = rq_h[gi], ql = rq_l[gi];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)qh[e] + (float)ql[e];
                    if constexpr (PW) {                                   // the second GEMM reads the sum
                        *reinterpret_cast<f32x4*>(T + r * TS + c8) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(T + r * TS + c8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                }
                f16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 h, l; split_f16(v[e], h, l, amax); ph[e] = h; pl[e] = l; }
                _Float16* q = oh + (int64_t)m * 2 * a.out_ldc + c8;
                store_act16(q, ph, st_plain);
                store_act16(q + a.out_ldc, pl, st_plain);
            }
            if constexpr (PW) {
                // ---- fused 1x1 conv: out2[RG x pw_cout] = act[RG x pw_k] * W2^T, same three-product split arithmetic.
                // T holds this conv's activations x SPLIT_SCALE (the stored format), so hi / lo come from it directly.
                // Work item = one 16x16 output tile; A from T (ds_read_b128 pairs), B straight from the 1x1 conv's packed
                // planes (a few KiB, L1/L2 resident).
                __syncthreads();
                float* T2 = T + RG * TS;
                const int lane = tid & 63, wave = tid >> 6;
                const int lr16 = lane & 15, lh16 = lane >> 4;
                const int k2s = a.pw_k / 32;
                const int n2 = pw_j * 16 + lr16;
                for (int sl = wave / pw_n2t; sl < RG / 16; sl += (NT / 64) / pw_n2t) {
                    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
                    const float* trow = T + (sl * 16 + lr16) * TS + lh16 * 8;
#pragma unroll
                    for (int ks = 0; ks < PWK; ++ks) {
                        if (ks < k2s) {                                   // uniform
                            const f32x4 x0 = *reinterpret_cast<const f32x4*>(trow + ks * 32);
                            const f32x4 x1 = *reinterpret_cast<const f32x4*>(trow + ks * 32 + 4);
                            f16x8 ah, al;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float x = e < 4 ? x0[e] : x1[e - 4];
                                const _Float16 h = (_Float16)x;
                                ah[e] = h; al[e] = (_Float16)(x - (float)h);
                            }
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b2h[ks], acc2, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b2l[ks], acc2, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b2h[ks], acc2, 0, 0, 0);
                        }
                    }
                    if (a.pw_leaky == 2) {                                   // uniform
#pragma unroll
                        for (int e = 0; e < 4; ++e) T2[(sl * 16 + 4 * lh16 + e) * T2S + n2] = silu_scaled(acc2[e] * inv2 + bias2, 1.0f / SPLIT_SCALE);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc2[e] * inv2 + bias2;
                            if (a.pw_leaky) v = v > 0.f ? v : v * 0.1f;
                            T2[(sl * 16 + 4 * lh16 + e) * T2S + n2] = v;
                        }
                    }
                }
                __syncthreads();
                const int gpr2 = a.pw_cout / 8;
                _Float16* oh2 = reinterpret_cast<_Float16*>(a.pw_out) + a.pw_out_coff;
                for (int g = tid; g < RG * gpr2; g += NT) {
                    const int r = g / gpr2, c8 = (g - r * gpr2) * 8;
                    const int m = bm * BM + rg + r;
                    if (m >= M) continue;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(T2 + r * T2S + c8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(T2 + r * T2S + c8 + 4);
                    f16x8 ph, pl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { _Float16 h, l; split_f16(e < 4 ? v0[e] : v1[e - 4], h, l, amax); ph[e] = h; pl[e] = l; }
                    _Float16* q = oh2 + (int64_t)m * 2 * a.pw_out_ldc + c8;
                    store_act16(q, ph, st_plain);
                    store_act16(q + a.pw_out_ldc, pl, st_plain);
                }
            }
        }
        if (rg + RG < BM) __syncthreads();
    }
    if constexpr (EPI != EPI_DECODE) split_overflow_report(a.ovf, amax);
}

}  // namespace rtod
