// Stem + first down-sampling convolution in one kernel (split-precision f16 MFMA), gfx950.
//
// Layers 0 and 1 of Darknet-53 (reference: conv -> BN(eval) -> leaky twice, src/darknet.py:467-501; cfg/yolov3.cfg blocks 1-2):
//     s  = leaky(conv3x3 s1 p1 (x [B,3,H,W] fp32 NCHW) -> 32 channels)          608x608x32 at the benchmark size
//     o1 = leaky(conv3x3 s2 p1 (s) -> 64 channels)                               304x304x64
//     o2 = leaky(conv1x1 (o1) -> 32 channels)          (optional: the 1x1 conv the plan hosts in layer 1's epilogue)
// Stand-alone, the stem writes s to HBM (378 MB at 608x608 batch 8) and layer 1 reads it back through a 9-tap gather:
// 0.12 + 0.19 ms of a 4.2 ms forward for 3 % of its FLOPs.  Here s never leaves the CU.
//
// A persistent workgroup (8 waves, one per CU) is two halves of four waves; each half walks its own 4 x 16 tiles of o1 with its
// own LDS patches, one barrier interval out of phase with the other half, so that on every SIMD one wave is in its
// vector-heavy phase (1) while its partner is in the matrix-heavy phase (2).  Per tile of a half:
//   0. input patch: the 3 x 11 x 35 input values under the tile are fetched two tiles ahead (coalesced rows, 5 values per
//      thread), split ONCE into (hi | lo << 16) f16 pairs of 8*x and kept in LDS (5 KB) — the stem's 27 taps re-read every
//      input value 9 times, so converting at the gather costs 9x the vector work;
//   1. stem phase: the 9 x 33 pixels of s the tile touches are computed 16 pixels at a time — the 27-tap dot product is
//      one k32 step of v_mfma_f32_16x16x32_f16 in the split arithmetic of conv_stem.hip (x*8 = xh + xl, three products),
//      operands gathered from the input patch (8 ds_read_b32 + 8 v_perm per 16 pixels, read one group ahead) —
//      scaled / biased / activated / split exactly as conv_stem_split_kernel does, and written to an LDS patch in the
//      activation format (hi plane, lo plane, 64-byte rows, chunk swizzle).  Pixels outside the image are written as zeros
//      (layer 1's padding).  Columns are stored even-then-odd per patch row, so the stride-2 taps read 16 CONSECUTIVE rows;
//   2. layer-1 phase: wave w of the half owns output row w of the tile: 9 taps x (1 pixel tile x 4 channel tiles x 3 products),
//      weights of all 9 taps resident in LDS for the lifetime of the workgroup (73 KB, shared by both halves) — no staging, no
//      barrier between taps, tap t + 1's fragments read while tap t's MFMAs run;
//   3. epilogue from the accumulators (transposed product, conv_f16s3_common.h): the lane's 8 consecutive channels of a pixel
//      are stored as 16 bytes per plane AND are, as they stand, the activation operand of the hosted 1x1 conv (k = 8*lh + j
//      of 32-channel chunk P), whose two k32 steps run on them directly (its weights live in registers).
// Measured (MI355X, 608x608 batch 8, s_memtime stamps of the RTOD_DIAG build): 152-160 us against 126 + 190 us for the two
// stand-alone kernels; per barrier interval a wave spends ~5.5k cycles in phase 1 or ~5.8k in phases 0 + 2 + 3 — vector-issue
// bound (≈750 VALU instructions against 150 MFMAs per SIMD and interval; an MFMA holds the SIMD's vector issue for half
// of its 16 cycles), not MFMA- or LDS-bound.
// Arithmetic, K order and rounding points are those of the stand-alone kernels (conv_stem.hip, conv_igemm_f16s3.hip with
// EPI_SPLIT_PW): the outputs are bit-identical to the unfused plan's.
#include "conv_f16s3_common.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace rtod {

constexpr int S2_TH = 4, S2_TW = 16;                                   // o1 tile of one wave half
constexpr int S2_PR = 2 * S2_TH + 1, S2_PC = 2 * S2_TW + 1;            // 9 x 33 pixels of s
constexpr int S2_EVEN = S2_TW + 1;                                     // even columns first (17), then the 16 odd ones
constexpr int S2_PE = S2_PR * S2_PC;                                   // 297
constexpr int S2_GROUPS = (S2_PE + 15) / 16;                           // 19 groups of 16 patch pixels
constexpr int S2_PROWS = S2_GROUPS * 16, S2_PLANE = S2_PROWS * 64;     // 304 rows, 19 456 bytes per plane
constexpr int S2_W1 = 9 * 64 * 64;                                     // one plane of layer 1's weights in LDS: [tap][64 rows][64 B]
constexpr int S2_WAVES = 8, S2_NT = S2_WAVES * 64;
constexpr int S2_HW = S2_WAVES / 2, S2_HT = S2_HW * 64;                // waves / threads of one half
constexpr int S2_GPW = (S2_GROUPS + S2_HW - 1) / S2_HW;                // pixel groups per wave (5; the last wave has 4)
constexpr int S2_IR = S2_PR + 2, S2_IC = S2_PC + 2;                     // 11 x 35 input pixels under the patch
constexpr int S2_IN = 3 * S2_IR * S2_IC;                               // 1155 input values
constexpr int S2_VPT = (S2_IN + 1 + S2_HT - 1) / S2_HT;                // input values per thread of a half (5)
constexpr int S2_INP = S2_VPT * S2_HT;                                 // padded: dwords S2_IN.. stay zero (operand of the k >= 27 lanes)
constexpr int S2_TAB = (64 + 64 + 32 + 32) * 4;
constexpr int S2_ZERO = ((S2_PR - 1) * S2_IC + S2_PC - 1) * 4 + 4;         // zero region: reads at (receptive-field corner) + (zero-lane offset) for every corner
constexpr int S2_LDS = 2 * 2 * S2_PLANE + 2 * S2_W1 + 2 * S2_INP * 4 + S2_TAB + (S2_ZERO + 15) / 16 * 16;
static_assert(S2_INP >= S2_IN + 1, "input patch padding");
static_assert(S2_TH == S2_HW, "one output row per wave of a half");

#ifdef RTOD_DIAG
constexpr int S2_SLOTS = 6, S2_SBLOCKS = 64;
__device__ unsigned long long g_s2_stamps[S2_SBLOCKS * S2_WAVES * (S2_SLOTS + 1)];
#define S2_STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ts_[i] += tn_ - tprev_; tprev_ = tn_; }
#else
#define S2_STAMP(i)
#endif

struct Stem2Args {
    const float* x; unsigned x_bytes;                 // [B,3,H,W] fp32
    const _Float16* w0h; const _Float16* w0l;         // stem weights [32][32] f16 hi / lo, k = (ky*3+kx)*3 + c (plan.cpp, stem split packing)
    const float* inv0; const float* bias0; int leaky0;
    ConvArgs c1;                                      // layer 1 (weights, scale, bias, output view) with the hosted 1x1 conv's pw_* fields
    int B, H, W;                                      // input / stem geometry; c1.Ho, c1.Wo = layer 1's output
    int tiles_x, tiles_y;
    int dbg;                                          // RTOD_DIAG builds only: phase ablation bits (timing experiments)
};

__device__ __forceinline__ void s2_dma_pair(const __amdgpu_buffer_rsrc_t rsrc_hi, const __amdgpu_buffer_rsrc_t rsrc_lo, unsigned voffset,
                                            unsigned soff, unsigned lds_hi, unsigned lds_lo) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(rsrc_hi), "s"(rsrc_lo), "s"(soff), "s"(lds_hi), "s"(lds_lo)
        : "memory");
}

template <bool PW>
__global__ __launch_bounds__(S2_NT, 2)
void conv_stem2_f16s3_kernel(const Stem2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int half = wave >> 2, wq = wave & 3;                          // wave half (0 / 1) and the wave's index inside it
    const int th = tid & (S2_HT - 1);
    const int lr = lane & 15, lh = lane >> 4;
    unsigned char* const patch = smem + half * 2 * S2_PLANE;            // this half's stem patch: [hi | lo][304][64]
    unsigned char* const w1 = smem + 2 * 2 * S2_PLANE;                  // [hi | lo][9][64][64], shared by both halves
    unsigned* const inp = reinterpret_cast<unsigned*>(w1 + 2 * S2_W1) + half * S2_INP;   // this half's [3][11][35] (hi | lo << 16) of 8*x, zero tail
    float* const tab = reinterpret_cast<float*>(w1 + 2 * S2_W1 + 2 * S2_INP * 4);        // inv1*8 [64], bias1*8 [64], inv2*8 [32], bias2*8 [32]
    unsigned char* const zreg = w1 + 2 * S2_W1 + 2 * S2_INP * 4 + S2_TAB;                 // S2_ZERO bytes of zeros (operand of the k >= 27 lanes)
    const ConvArgs& c = a.c1;
    const int H = a.H, W = a.W;
    const int64_t plane = (int64_t)H * W;
    const int n_tiles = a.B * a.tiles_y * a.tiles_x;
    const int n_pairs = (n_tiles + 1) / 2;                              // a workgroup takes tiles in pairs: tile 2p for half 0, 2p + 1 for half 1

    // ---- once per workgroup: layer 1's weights -> LDS, scale / bias tables
    {
        const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)c.w_hi, 0, c.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)c.w_lo, 0, c.w_bytes, 0x00020000);
        const unsigned lds0 = (unsigned)(size_t)w1;
        const int lrow = lane >> 2;
        for (int p = wave; p < 9 * 4; p += S2_WAVES) {                  // (tap, 16-row block): one hi + one lo piece each
            const int tap = p >> 2, rb = p & 3;
            const int rho = rb * 16 + lrow;
            const int ch = (lane & 3) ^ ((rho >> 1) & 3);
            const unsigned vo = (unsigned)(tr_chan_of_row(rho) * 32 + ch * 8) * 2u;      // planes are [chunk*9 + tap][Npad][32], Cin = 32: chunk 0
            const unsigned l = lds0 + (unsigned)(tap * 64 * 64 + rb * 1024);
            s2_dma_pair(rs_wh, rs_wl, vo, (unsigned)tap * (unsigned)c.Npad * 64u, l, l + S2_W1);
        }
        for (int i = tid; i < 64; i += S2_NT) { tab[i] = c.inv_scale[i] * SPLIT_SCALE; tab[64 + i] = c.bias[i] * SPLIT_SCALE; }
        if constexpr (PW)
            for (int i = tid; i < 32; i += S2_NT) { tab[128 + i] = c.pw_inv_scale[i] * SPLIT_SCALE; tab[160 + i] = c.pw_bias[i] * SPLIT_SCALE; }
        for (int i = tid; i < (S2_ZERO + 3) / 4; i += S2_NT) reinterpret_cast<unsigned*>(zreg)[i] = 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- stem constants of this lane: weight fragments (rows in the transposed product's channel order), scale, bias
    // (x SPLIT_SCALE folded in: a power of two, exact)
    f16x8 w0h[2], w0l[2];
    float inv0[8], bias0[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = tr_chan_of_row(t * 16 + lr);
        w0h[t] = *reinterpret_cast<const f16x8*>(a.w0h + n * 32 + lh * 8);
        w0l[t] = *reinterpret_cast<const f16x8*>(a.w0l + n * 32 + lh * 8);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { inv0[e] = a.inv0[8 * lh + e] * SPLIT_SCALE; bias0[e] = a.bias0[8 * lh + e] * SPLIT_SCALE; }
    // leaky(v) = max(v, 0.1 v): bit for bit (v > 0 ? v : 0.1 v)
    const float slope0 = a.leaky0 ? 0.1f : 1.0f, slope1 = c.leaky ? 0.1f : 1.0f, slope2 = (PW && c.pw_leaky) ? 0.1f : 1.0f;
    // hosted 1x1 conv: its weight fragments live in registers (2 k chunks x 2 channel tiles, hi / lo)
    f16x8 w2h[2][2], w2l[2][2];
    if constexpr (PW) {
#pragma unroll
        for (int P = 0; P < 2; ++P)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t o = ((int64_t)P * c.pw_npad + tr_chan_of_row(t * 16 + lr)) * 32 + lh * 8;
                w2h[P][t] = *reinterpret_cast<const f16x8*>(c.pw_wh + o);
                w2l[P][t] = *reinterpret_cast<const f16x8*>(c.pw_wl + o);
            }
    }
    // input-patch dword of this lane's 8 k values (k = 8*lh + e = (ky*3 + kx)*3 + c) relative to the stem pixel's patch
    // position; for the k >= 27 lanes the offset points from this half's input patch into the zero region, so that corner + offset
    // reads a zero for every corner WITHOUT a per-value select (round 4: 8 selects per 16 pixels less)
    int koff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = lh * 8 + e;
        const int tap = k / 3, cc = k - tap * 3;
        const int ky = tap / 3, kx = tap - ky * 3;
        koff[e] = k < 27 ? ((cc * S2_IR + ky) * S2_IC + kx) * 4 : (int)(zreg - reinterpret_cast<unsigned char*>(inp));
    }
    // patch pixels of this lane's groups: group g = wq + 4 j, patch row q = 16*g + lr = py*33 + column slot
    int gpyx[S2_GPW], gpb[S2_GPW];                                      // (py << 8 | px), or -1 beyond the patch; input-patch byte offset
#pragma unroll
    for (int j = 0; j < S2_GPW; ++j) {
        const int g = wq + j * S2_HW;
        const int q = g * 16 + lr;
        const int py = q / S2_PC, cs = q - py * S2_PC;
        const int px = cs < S2_EVEN ? 2 * cs : 2 * (cs - S2_EVEN) + 1;
        const bool in_patch = g < S2_GROUPS && q < S2_PE;              // beyond the patch: never in the image -> zeros
        gpyx[j] = in_patch ? (py << 8) | px : -1;
        gpb[j] = in_patch ? (py * S2_IC + px) * 4 : 0;
    }
    // input values of this thread: i = th + 256 r -> (channel, row, column) of the 3 x 11 x 35 patch (i >= 1155: the zero tail)
    int in_off[S2_VPT], in_y[S2_VPT], in_x[S2_VPT];
#pragma unroll
    for (int r = 0; r < S2_VPT; ++r) {
        const int i = th + S2_HT * r;
        const int cc = i / (S2_IR * S2_IC), rem = i - cc * (S2_IR * S2_IC);
        const int iy = rem / S2_IC, ix = rem - iy * S2_IC;
        in_off[r] = (cc * (int)plane + iy * W + ix) * 4;
        in_y[r] = i < S2_IN ? iy : -(1 << 20);
        in_x[r] = ix;
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const int w_lane = lr * 64 + ((lh ^ ((lr >> 1) & 3)) << 4);         // fragment address inside a 16-row weight block
    float amax = 0.f;
#ifdef RTOD_DIAG
    unsigned long long ts_[S2_SLOTS] = {0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart_ = tprev_;
#endif
    _Float16* const o1 = reinterpret_cast<_Float16*>(c.out) + c.out_coff;
    _Float16* const o2 = PW ? reinterpret_cast<_Float16*>(c.pw_out) + c.pw_out_coff : nullptr;

    // Tile coordinates are carried incrementally (integer division by run-time values costs ~200 dependent cycles per use;
    // two uses per barrier interval were a third of the interval): packed (b << 20 | ty << 10 | tx), advanced by the
    // workgroup's stride of 2 * gridDim.x tiles with carries; a tile beyond the last one has b >= B.
    const int step = 2 * (int)gridDim.x;
    const int d_tx = step % a.tiles_x, d_ty = (step / a.tiles_x) % a.tiles_y, d_b = step / a.tiles_x / a.tiles_y;
    auto advance = [&](int pk) -> int {
        int tx = (pk & 1023) + d_tx, ty = ((pk >> 10) & 1023) + d_ty, b = (pk >> 20) + d_b;
        if (tx >= a.tiles_x) { tx -= a.tiles_x; ++ty; }
        if (ty >= a.tiles_y) { ty -= a.tiles_y; ++b; }
        if (b > 1024) b = 1024;                                          // (stays past the end; keeps the packed field in range)
        return (b << 20) | (ty << 10) | tx;
    };
    auto valid = [&](int pk) -> bool { return (pk >> 20) < a.B; };
    // Input gather of one tile: raw fp32, issued two tiles ahead of its use as MFMA operand (one tile in registers, one in LDS)
    // so that the round trip runs under the MFMA phases.  Values outside the image read as zero (the stem's padding).
    float xin[S2_VPT];
    auto gather = [&](int pk) {
        const int x0 = (pk & 1023) * S2_TW, y0 = ((pk >> 10) & 1023) * S2_TH, b = pk >> 20;
        const int gy0 = 2 * y0 - 2, gx0 = 2 * x0 - 2;                   // input pixel of patch (0, 0)
        const int base = (b * 3 * (int)plane + gy0 * W + gx0) * 4;
#pragma unroll
        for (int r = 0; r < S2_VPT; ++r) {
            const bool ok = (unsigned)(gy0 + in_y[r]) < (unsigned)H && (unsigned)(gx0 + in_x[r]) < (unsigned)W;
            const unsigned vo = ok ? (unsigned)(base + in_off[r]) : 0x80000000u;
            xin[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, vo, 0, 0));
        }
    };
    auto stage_input = [&]() {                                          // registers -> (hi | lo << 16) pairs in LDS
#pragma unroll
        for (int r = 0; r < S2_VPT; ++r) {
            const float v = xin[r] * SPLIT_SCALE;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            inp[th + S2_HT * r] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
        }
    };

    // ---- stem phase of one tile: input patch -> 9 x 33 pixels of s in this half's LDS patch
    auto stem_phase_impl = [&](int pk, auto interior_c) {
        constexpr bool INTERIOR = decltype(interior_c)::value;         // the whole 9 x 33 patch lies inside the image: no zero-padding selects
        const int x0 = (pk & 1023) * S2_TW, y0 = ((pk >> 10) & 1023) * S2_TH;
        const int sy0 = 2 * y0 - 1, sx0 = 2 * x0 - 1;                   // stem pixel of patch (0, 0)
        // the 8 operand dwords of group j + 1 are read while group j computes
        unsigned d[2][8];
        auto read_group = [&](int j, int buf) {
            int pb = gpb[j];                                             // input-patch dword of the pixel's receptive-field corner
            asm volatile("" : "+v"(pb));                                 // (recomputed sums: 40 hoisted addresses would cost 40 VGPRs)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                d[buf][e] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(inp) + (pb + koff[e]));
        };
        read_group(0, 0);
#pragma unroll
        for (int j = 0; j < S2_GPW; ++j) {
            if (wq + j * S2_HW >= S2_GROUPS) break;                      // wave-uniform
            const int cur = j & 1;
            if (j + 1 < S2_GPW) read_group(j + 1, cur ^ 1);              // (the last wave's fifth group does not exist: it reads pixel 0, unused)
            __builtin_amdgcn_sched_barrier(0);
            const int py = gpyx[j] >> 8, px = gpyx[j] & 255;
            const bool inimg = gpyx[j] >= 0 && (unsigned)(sy0 + py) < (unsigned)H && (unsigned)(sx0 + px) < (unsigned)W;
            u32x4 uh, ul;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uh[i] = __builtin_amdgcn_perm(d[cur][2 * i + 1], d[cur][2 * i], 0x05040100u);
                ul[i] = __builtin_amdgcn_perm(d[cur][2 * i + 1], d[cur][2 * i], 0x07060302u);
            }
            const f16x8 xh = __builtin_bit_cast(f16x8, uh), xl = __builtin_bit_cast(f16x8, ul);
            f16x8 ph, pl;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                s = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h[t], xl, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0l[t], xh, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h[t], xh, s, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = s[e] * inv0[4 * t + e] + bias0[4 * t + e];
                    v = fmaxf(v, v * slope0);
                    if constexpr (!INTERIOR) v = inimg ? v : 0.f;       // outside the image: layer 1's zero padding (patch rows past the 297th are never read)
                    _Float16 h, l;
                    split_f16(v, h, l, amax);
                    ph[4 * t + e] = h; pl[4 * t + e] = l;
                }
            }
            const int q = (wq + j * S2_HW) * 16 + lr;
            const int o = q * 64 + ((lh ^ ((q >> 1) & 3)) << 4);
            *reinterpret_cast<f16x8*>(patch + o) = ph;
            *reinterpret_cast<f16x8*>(patch + S2_PLANE + o) = pl;
        }
    };

    auto stem_phase = [&](int pk) {
        const int x0 = (pk & 1023) * S2_TW, y0 = ((pk >> 10) & 1023) * S2_TH;
        const int sy0 = 2 * y0 - 1, sx0 = 2 * x0 - 1;
        if (sy0 >= 0 && sx0 >= 0 && sy0 + S2_PR <= H && sx0 + S2_PC <= W) stem_phase_impl(pk, std::true_type{});
        else stem_phase_impl(pk, std::false_type{});
    };

    // ---- layer-1 phase of one tile: this wave's output row (ty = wq), 64 channels, + epilogue (+ hosted 1x1 conv)
    auto conv_phase = [&](int pk) {
        const int x0 = (pk & 1023) * S2_TW, y0 = ((pk >> 10) & 1023) * S2_TH, b = pk >> 20;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // operands of tap t + 1 are read from LDS while the 12 MFMAs of tap t run (explicit double buffer; the scheduling
        // barriers keep the compiler from sinking the reads back to their uses)
        f16x8 fxh[2], fxl[2], fwh[2][4], fwl[2][4];
        auto load_tap = [&](int tap, int buf) {
            const int ky = tap / 3, kx = tap % 3;
            const int q = (2 * wq + ky) * S2_PC + (kx == 1 ? S2_EVEN : (kx >> 1)) + lr;      // even / odd / next even column slot
            const int o = q * 64 + ((lh ^ ((q >> 1) & 3)) << 4);
            fxh[buf] = *reinterpret_cast<const f16x8*>(patch + o);
            fxl[buf] = *reinterpret_cast<const f16x8*>(patch + S2_PLANE + o);
            const unsigned char* wp = w1 + tap * 64 * 64 + w_lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fwh[buf][t] = *reinterpret_cast<const f16x8*>(wp + t * 1024);
                fwl[buf][t] = *reinterpret_cast<const f16x8*>(wp + S2_W1 + t * 1024);
            }
        };
#ifdef RTOD_DIAG
        if (!(a.dbg & 16)) {
#endif
        load_tap(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < 9) load_tap(tap + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwh[cur][t], fxl[cur], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwl[cur][t], fxh[cur], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwh[cur][t], fxh[cur], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef RTOD_DIAG
        }
        if (a.dbg & 32) { amax += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]; return; }
        if (a.dbg & 64) { asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); S2_STAMP(2) }
#endif
        // epilogue: pixel (y0 + wq, x0 + lr), channels 32P + 8*lh + {0..7}
        const int oy = y0 + wq, ox = x0 + lr;
#ifdef RTOD_DIAG
        const bool pok = oy < c.Ho && ox < c.Wo && !(a.dbg & 4);
#else
        const bool pok = oy < c.Ho && ox < c.Wo;
#endif
        const int64_t m = pok ? ((int64_t)b * c.Ho + oy) * c.Wo + ox : 0;
        f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            const int c0 = 32 * P + 8 * lh;
            const f32x4 i0 = *reinterpret_cast<const f32x4*>(tab + c0), i1 = *reinterpret_cast<const f32x4*>(tab + c0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(tab + 64 + c0), b1 = *reinterpret_cast<const f32x4*>(tab + 64 + c0 + 4);
            f16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = e < 4 ? acc[2 * P][e] : acc[2 * P + 1][e - 4];
                float v = s * (e < 4 ? i0[e] : i1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
                v = fmaxf(v, v * slope1);
                _Float16 h, l;
                split_f16(v, h, l, amax);
                ph[e] = h; pl[e] = l;
            }
            if (pok) {
                _Float16* q = o1 + m * 2 * c.out_ldc + c0;
                store_act16(q, ph, false);
                store_act16(q + c.out_ldc, pl, false);
            }
            if constexpr (PW) {                                          // the stored values are the 1x1 conv's operand of k chunk P
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[P][t], pl, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[P][t], ph, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[P][t], ph, acc2[t], 0, 0, 0);
                }
            }
        }
        if constexpr (PW) {
            const int c0 = 8 * lh;
            const f32x4 i0 = *reinterpret_cast<const f32x4*>(tab + 128 + c0), i1 = *reinterpret_cast<const f32x4*>(tab + 128 + c0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(tab + 160 + c0), b1 = *reinterpret_cast<const f32x4*>(tab + 160 + c0 + 4);
            f16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = e < 4 ? acc2[0][e] : acc2[1][e - 4];
                float v = s * (e < 4 ? i0[e] : i1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
                v = fmaxf(v, v * slope2);
                _Float16 h, l;
                split_f16(v, h, l, amax);
                ph[e] = h; pl[e] = l;
            }
            if (pok) {
                _Float16* q = o2 + m * 2 * c.pw_out_ldc + c0;
                store_act16(q, ph, false);
                store_act16(q + c.pw_out_ldc, pl, false);
            }
        }
    };

    // ---- schedule.  The two halves run the same program one barrier interval apart: while half 0 computes the stem patch of
    // its tile n (vector work: operand gather, scale / activation / split), half 1 runs layer 1 on its tile n - 1 (matrix work)
    // on the same four SIMDs, and vice versa.  Interval `it`: half h is in its stem phase when (it + h) is even.
    //   half 0:  S(0) | C(0) | S(1) | C(1) | ...            half 1:  -  | S(0) | C(0) | S(1) | ...
    // Every wave passes the same 2 * N + 1 barriers (N = pairs of this workgroup, uniform); a half without a tile idles.
#ifdef RTOD_DIAG
    tprev_ = __builtin_amdgcn_s_memtime();
#endif
    const int n_mine = (int)blockIdx.x < n_pairs ? (n_pairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    int pk0, pk1, pk2;                                                   // this half's tile now / staged next / being fetched
    {
        const int t0 = 2 * (int)blockIdx.x + half;
        const int tx = t0 % a.tiles_x, r = t0 / a.tiles_x;
        pk0 = ((r / a.tiles_y) << 20) | ((r % a.tiles_y) << 10) | tx;
        pk1 = advance(pk0); pk2 = advance(pk1);
        if (valid(pk0)) { gather(pk0); stage_input(); }
        if (valid(pk1)) gather(pk1);
    }
    __syncthreads();                                                     // weights, tables, first input patches
    for (int it = 0; it <= 2 * n_mine; ++it) {
        const int ph = it - half;                                        // this half's own interval count
        if (ph >= 0 && ph < 2 * n_mine) {
            if ((ph & 1) == 0) {
#ifdef RTOD_DIAG
                if (!(a.dbg & 1))
#endif
                if (valid(pk0)) stem_phase(pk0);
                S2_STAMP(0)
            } else {
                if (valid(pk0)) {
                    // the input patch is dead: stage the next tile's (fetched during the previous tile), fetch the one after it
#ifdef RTOD_DIAG
                    if (!(a.dbg & 8))
#endif
                    if (valid(pk1)) {
                        stage_input();
                        if (valid(pk2)) gather(pk2);
                    }
                    S2_STAMP(1)
#ifdef RTOD_DIAG
                    if (!(a.dbg & 2))
#endif
                    conv_phase(pk0);
                    S2_STAMP(3)
                }
                pk0 = pk1; pk1 = pk2; pk2 = advance(pk2);
            }
        }
        S2_STAMP(5)
        __syncthreads();
        S2_STAMP(4)
    }
#ifdef RTOD_DIAG
    if (lane == 0 && blockIdx.x < S2_SBLOCKS) {
        unsigned long long* o = g_s2_stamps + (blockIdx.x * S2_WAVES + wave) * (S2_SLOTS + 1);
        for (int i = 0; i < S2_SLOTS; ++i) o[i] = ts_[i];
        o[S2_SLOTS] = tprev_ - tstart_;
    }
#endif
    split_overflow_report(c.ovf, amax);
}

bool conv_stem2_supported(int k0, int s0, int p0, int cin0, int cout0, int k1, int s1, int p1, int cout1, int pw_cout) {
    return k0 == 3 && s0 == 1 && p0 == 1 && cin0 == 3 && cout0 == 32 && k1 == 3 && s1 == 2 && p1 == 1 && cout1 == 64 &&
           (pw_cout == 0 || pw_cout == 32);
}

int conv_stem2_kernel_name(int pw, char* buf, size_t len) {
    return snprintf(buf, len, "void rtod::conv_stem2_f16s3_kernel<%s>(rtod::Stem2Args)", pw ? "true" : "false");
}

int launch_conv_stem2_f16s3(const float* x, int B, int H, int W, const _Float16* w0h, const _Float16* w0l, const float* inv0, const float* bias0,
                            int leaky0, const ConvArgs& c1, hipStream_t s) {
    if (!x || !w0h || !w0l || !inv0 || !bias0 || !c1.w_hi || !c1.w_lo || !c1.inv_scale || !c1.bias || !c1.out) { set_error("conv_stem2: null pointer"); return RTOD_E_ARG; }
    const bool pw = c1.pw_wh != nullptr;
    if (c1.Cin != 32 || c1.Cout != 64 || c1.kh != 3 || c1.kw != 3 || c1.stride != 2 || c1.pad != 1 || c1.Hi != H || c1.Wi != W ||
        c1.Ho != (H - 1) / 2 + 1 || c1.Wo != (W - 1) / 2 + 1 || c1.res || c1.dec.enabled || c1.out_ldc % 8 || c1.out_coff % 8 || c1.Npad < 64) {
        set_error("conv_stem2: layer 1 is not a 3x3 stride-2 32 -> 64 convolution over the stem's output"); return RTOD_E_ARG;
    }
    if (pw && (!c1.pw_wl || !c1.pw_inv_scale || !c1.pw_bias || !c1.pw_out || c1.pw_k != 64 || c1.pw_cout != 32 || c1.pw_npad < 32 || c1.pw_out_ldc % 8 || c1.pw_out_coff % 8)) {
        set_error("conv_stem2: hosted 1x1 conv must be 64 -> 32"); return RTOD_E_ARG;
    }
    if ((int64_t)B * 3 * H * W * 4 >= (1ll << 31) || (int64_t)B * c1.Ho * c1.Wo >= (1ll << 31)) { set_error("conv_stem2: input exceeds 2 GiB / int32 pixels"); return RTOD_E_ARG; }
    if (c1.Wo > 1023 * S2_TW || c1.Ho > 1023 * S2_TH || B > 1023) { set_error("conv_stem2: tile coordinates exceed the packed 10-bit fields"); return RTOD_E_ARG; }
    Stem2Args a;
    a.x = x; a.x_bytes = (unsigned)((int64_t)B * 3 * H * W * 4);
    a.w0h = w0h; a.w0l = w0l; a.inv0 = inv0; a.bias0 = bias0; a.leaky0 = leaky0;
    a.c1 = c1; a.B = B; a.H = H; a.W = W;
    a.dbg = 0;
#ifdef RTOD_DIAG
    if (const char* e = getenv("RTOD_S2_DBG")) a.dbg = atoi(e);
#endif
    a.tiles_x = (c1.Wo + S2_TW - 1) / S2_TW; a.tiles_y = (c1.Ho + S2_TH - 1) / S2_TH;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return hip_fail(hipGetLastError(), "conv_stem2 device query");
    const int64_t pairs = ((int64_t)B * a.tiles_x * a.tiles_y + 1) / 2;      // one tile per wave half
    const int grid = (int)(pairs < cus ? pairs : cus);
    auto k_pw = conv_stem2_f16s3_kernel<true>;
    auto k_plain = conv_stem2_f16s3_kernel<false>;
    static std::atomic<unsigned long long> attr_done{0};
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_pw), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_plain), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return hip_fail(hipGetLastError(), "conv_stem2 LDS attribute");
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    static_assert(S2_LDS <= 160 * 1024, "LDS budget");
    if (pw) hipLaunchKernelGGL(k_pw, dim3(grid), dim3(S2_NT), S2_LDS, s, a);
    else hipLaunchKernelGGL(k_plain, dim3(grid), dim3(S2_NT), S2_LDS, s, a);
#ifdef RTOD_DIAG
    if (getenv("RTOD_S2_STAMPS")) {
        static int printed = 0;
        if (printed < 3 && hipDeviceSynchronize() == hipSuccess) {
            static unsigned long long h[S2_SBLOCKS * S2_WAVES * (S2_SLOTS + 1)];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_s2_stamps), sizeof(h)) == hipSuccess) {
                const int nb = grid < S2_SBLOCKS ? grid : S2_SBLOCKS;
                for (int hf = 0; hf < 2; ++hf) {
                    double sum[S2_SLOTS + 1] = {0};
                    for (int b = 0; b < nb; ++b) for (int w = hf * 4; w < hf * 4 + 4; ++w) for (int i = 0; i <= S2_SLOTS; ++i)
                        sum[i] += (double)h[(b * S2_WAVES + w) * (S2_SLOTS + 1) + i];
                    fprintf(stderr, "[s2 stamps] half %d cycles/wave: stem=%.0f stage=%.0f taps=%.0f epi=%.0f barrier=%.0f other=%.0f total=%.0f (pairs/wg %.1f)\n", hf,
                            sum[0] / (nb * 4), sum[1] / (nb * 4), sum[2] / (nb * 4), sum[3] / (nb * 4), sum[4] / (nb * 4), sum[5] / (nb * 4), sum[6] / (nb * 4), (double)pairs / grid);
                }
                ++printed;
            }
        }
    }
#endif
    return hip_fail(hipGetLastError(), "conv_stem2 launch");
}

}  // namespace rtod
