// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input band (split-precision f16 MFMA), gfx950.
//
// Same contract as conv_igemm_f16s3.hip for the layers it covers (reference: conv -> BN -> leaky,
// src/darknet.py:467-501, shortcut fused, 263-268).  The generic implicit-GEMM re-gathers every input
// pixel once per tap: 9x the L2->CU traffic and 9x the LDS writes for the A operand (measured: 900 MB of
// L1->L2 requests per launch for a 72 MB layer, ~1.3 ms of load stall over the 3x3 layers of YOLOv3).
// Here a workgroup owns BM CONSECUTIVE output pixels (linear index over batch, y, x) and stages,
// once per 32-channel chunk, the band of input pixels those outputs can touch:
//
//     band row r  <->  input pixel (linear)  m0 - W - 1 + r,   r in [0, BM + 2W + 2)
//
// so the A fragment of output row p for tap (ky,kx) is band row p + ky*W + kx: a constant shift per tap,
// read straight from LDS (im2col never materialised).  Taps that fall outside the image (left/right edge,
// top/bottom, other image of the batch) are redirected per lane to an all-zero LDS row by a 9-bit validity
// mask computed once.  K order is (channel chunk outer, tap inner): the weight planes are packed to match.
// B (weights) is streamed per (chunk, tap): double-buffered LDS, two register stage sets, asm buffer loads
// with counted vmcnt.
//
// MFMA shape: v_mfma_f32_16x16x32_f16, one k32 step per 32-channel stage.  Against 32x32x16 it moves the same
// LDS bytes per flop, but (a) the chip holds a higher clock on it (MI355X_MICROARCH 'DVFS give-back' (7);
// measured here +3..5 % on the 76x76 / 38x38 layers) and (b) wave tiles are multiples of 16, not 32, which is
// what makes the 192- and 96-row workgroup tiles below possible with 8 waves: the grid of a layer is a fixed
// number of pixels, and 128-row tiles leave the 256 CUs 70 % filled on every YOLOv3 scale at batch 8
// (722 / 364 / 184 tiles on 512 resident slots).  Each lane reads 16-byte chunk lane/16 of row lane%16; the
// chunk swizzle (row >> 1) & 3 keeps that read conflict-free for EVERY band shift (tools/lds_bank_sim.py).
//
// A layer that this kernel supports always runs on it, at every batch size, whatever tile autotune picks:
// all tiles accumulate a pixel's K products in the same order (split-K is decided by the layer's shape, never by
// the batch), so the frame-independence guarantee (bitwise equal outputs for a frame whatever batch it rides in)
// holds under autotuning.
//
// Limits: stride 1, pad 1, 3x3, Cin % 32 == 0, W <= 94.
#include "conv_f16s3_common.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace rtod {

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 30 && N % 2 == 0, "vmcnt literal");
#define RTOD_VMCNT_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RTOD_VMCNT_CASE(2) RTOD_VMCNT_CASE(4) RTOD_VMCNT_CASE(6) RTOD_VMCNT_CASE(8) RTOD_VMCNT_CASE(10) RTOD_VMCNT_CASE(12) RTOD_VMCNT_CASE(14)
    RTOD_VMCNT_CASE(16) RTOD_VMCNT_CASE(18) RTOD_VMCNT_CASE(20) RTOD_VMCNT_CASE(22) RTOD_VMCNT_CASE(24) RTOD_VMCNT_CASE(26) RTOD_VMCNT_CASE(28) RTOD_VMCNT_CASE(30)
#undef RTOD_VMCNT_CASE
}
template <typename T, int N> __device__ __forceinline__ void tie_regs(T (&r)[N]) {   // pins later uses below a preceding wait
    static_assert(N >= 1 && N <= 5, "tie_regs");
    if constexpr (N == 1) asm volatile("" : "+v"(r[0]) :: "memory");
    else if constexpr (N == 2) asm volatile("" : "+v"(r[0]), "+v"(r[1]) :: "memory");
    else if constexpr (N == 3) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]) :: "memory");
    else if constexpr (N == 4) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
    else asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) :: "memory");
}

__device__ __forceinline__ int band_swz(int row) { return (row >> 1) & 3; }

// Diagnostic build only (make stamps -> librtod_stamps.so, -DRTOD_STAMPS): per-wave s_memtime attribution of the main
// loop's phases, written to a device table the launcher prints.  The product library compiles none of it.
#ifdef RTOD_STAMPS
constexpr int STAMP_SLOTS = 10, STAMP_BLOCKS = 128, STAMP_WAVES = 16;    // 8 / 9: slot 1 of the steps 2 / 0-1 steps after a band prefetch was issued
__device__ unsigned long long g_band_stamps[STAMP_BLOCKS * STAMP_WAVES * (STAMP_SLOTS + 1)];
__device__ unsigned long long g_band_real[STAMP_BLOCKS * STAMP_WAVES];     // s_memrealtime ticks (100 MHz) over the same span: the clock the chip held
#define RTOD_STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ts_[i] += tn_ - tprev_; tprev_ = tn_; }
#else
#define RTOD_STAMP(i)
#endif

// Diagnostic build only (make timeline -> librtod_tl.so, -DRTOD_TIMELINE): start / end wall clock (s_memrealtime, 100 MHz), shader
// cycles (s_memtime) and HW_ID of every workgroup of a launch — three clock reads per workgroup, nothing inside the loop —
// summarised by the launcher: round structure, per-round workgroup durations, tail, the clock the chip held.
#ifdef RTOD_TIMELINE
constexpr int TL_BLOCKS = 2048;
__device__ unsigned long long g_band_tl[TL_BLOCKS * 4];
#endif

// (Round 3's timing-only ablation knobs of this main loop — RTOD_ABL: no epilogue / no loads / no LDS writes / fragments read once /
//  no barrier — are kept as profiles/experiments/r03_band_ablation_knobs.patch: apply it, then `make abl ABL=<bits>`.)

constexpr int BAND_MAX_W = 94;
// Epilogue flavour.  0 (default): pixel-major accumulators, LDS-transposed stores (256-byte runs per pixel).  1: transposed
// product + register epilogue (conv_f16s3_epilogue_regs; 64-byte segments per pixel, no LDS pass) — bit-identical, measured
// 0.6-3 % slower on the band layers (A/B on one box, gpurun_out/ab1), so it is a build-time experiment only.
#ifndef RTOD_BAND_TR
#define RTOD_BAND_TR 0
#endif
#ifndef RTOD_BFRAG_AHEAD
#define RTOD_BFRAG_AHEAD 1
#endif
#ifndef RTOD_BFRAG_LA
#define RTOD_BFRAG_LA(minw, tn) 2
#endif
constexpr bool BAND_TR = RTOD_BAND_TR != 0;
// LDS-transposed epilogue: the launch allocates at least the transpose tile — the whole BM x BN fp32 tile, capped at 64 KiB (more rows go in passes)
__host__ __device__ constexpr int band_epi_bytes(int bm, int bn) { return BAND_TR ? 0 : (bm * bn * 4 < 65536 ? bm * bn * 4 : 65536); }
__host__ __device__ constexpr int band_rows(int bm, int w) { return (bm + 2 * w + 2 + 15) / 16 * 16; }   // zero row follows

// BM x BN workgroup tile, NWM x NWN waves of (BM/NWM) x (BN/NWN); MINW = waves/SIMD the register budget must allow.
//   128x128, 4x2: 8 waves of 32x64, 2 workgroups per CU (4 waves/SIMD) — tools/ubench_tiles.hip: occupancy buys more
//                 MFMA utilisation than a larger wave tile
//   192x128, 4x2 / 6x2: 8 waves of 48x64 or 12 of 32x64 — 76x76x8 pixels are 482 tiles instead of 722, 38x38x8 244
//   96x128, 2x4: 8 waves of 48x32 — 19x19x8 pixels x 1024 channels are 248 tiles instead of 184
// KG = 2: in-workgroup split-K.  Two groups of NWM x NWN waves each run the whole pipeline (own band, own weight
// stages) on the even / odd 32-channel chunks and the epilogue sums the two accumulator sets through LDS.  For layers
// whose grid cannot give a CU two workgroups (19x19: 248 tiles) this is what puts 16 waves on the CU.
template <int BM, int BN, int NWM, int NWN, int MINW, int EPI, int KG>
__global__ __launch_bounds__(NWM * NWN * 64 * KG, MINW)
void conv_band_f16s3_kernel(const ConvArgs a, const int grid_m, const int grid_n) {
    constexpr int WM = BM / NWM, WN = BN / NWN, NT = NWM * NWN * 64;   // NT = threads of one K group
    static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % NWM == 0 && BN % NWN == 0, "wave tile");
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int RPP = NT / 4;                               // rows per staging pass (4 x 16-B chunks per row)
    constexpr int B_SLOTS = (BN + RPP - 1) / RPP;
    constexpr int BAND_MAX = band_rows(BM, BAND_MAX_W);
    constexpr int BAND_SLOTS = (BAND_MAX + RPP - 1) / RPP;
    static_assert(RPP % 16 == 0, "a pass that runs past a panel is predicated per 16-row wave slice; swizzle period 8");
    constexpr int B_LOADS = 2 * B_SLOTS, BAND_LOADS = 2 * BAND_SLOTS;
    static_assert(B_LOADS + BAND_LOADS <= 30 && 2 * B_LOADS <= 30, "vmcnt literals");
    constexpr int PANEL_B = BN * 64;
    constexpr int BSTAGE = 2 * PANEL_B;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef RTOD_TIMELINE
    const unsigned long long tl_start_ = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tl_cstart_ = __builtin_amdgcn_s_memtime();
#endif
    const int W = a.Wi, H = a.Hi;
    const int NBR = band_rows(BM, W);                          // band rows incl. padding; the zero row is row NBR
    const int plane = (NBR + 1) * 64;
    const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);   // K group of this wave (uniform -> SGPR)
    unsigned char* bst = smem + kg * (2 * BSTAGE + 2 * plane); // per group: [2 stages][hi, lo][BN][64], then the band
    unsigned char* bandh = bst + 2 * BSTAGE;
    unsigned char* bandl = bandh + plane;
    const int zero_off = NBR * 64;

    const int nwg = grid_m * grid_n;
    int bid = blockIdx.x;
    if (!a.xcd_by_n) {                                         // XCD x (= blockIdx % 8) takes a contiguous range of pixel tiles
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }                                                          // else: grid_n % 8 == 0, bn = bid % grid_n -> XCD = bn % 8
    const int bm = bid / grid_n, bn = bid - bm * grid_n;

    const int tid = KG == 1 ? (int)threadIdx.x : (int)threadIdx.x - kg * NT;   // thread index within the K group
    const int M = a.B * H * W;                                 // Ho == Hi, Wo == Wi
    const int m0 = bm * BM;
    const int NB = BM + 2 * W + 2;
    const int c16 = tid & 3, row0 = tid >> 2;
    const unsigned PS = (unsigned)a.in_ldc * 4u;
    const unsigned lo_plane = (unsigned)a.in_ldc * 2u;

    // zero row (both planes)
    if (tid < 8) *reinterpret_cast<u32x4*>((tid < 4 ? bandh : bandl) + zero_off + (tid & 3) * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- band loads: per-thread constant voffset, channel chunk through soffset
    unsigned bvo[BAND_SLOTS];
#pragma unroll
    for (int j = 0; j < BAND_SLOTS; ++j) {
        const int r = row0 + j * RPP;
        const int q = m0 - W - 1 + r;
        bvo[j] = (r < NB && q >= 0 && q < M) ? (unsigned)q * PS + (unsigned)(a.in_coff + c16 * 8) * 2u : OOB;
    }
    unsigned wbase[B_SLOTS];
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i)
        wbase[i] = (row0 + i * RPP < BN) ? (unsigned)((bn * BN + (BAND_TR ? tr_chan_of_row(row0 + i * RPP) : row0 + i * RPP)) * 32 + c16 * 8) * 2u : OOB;
    const unsigned wchunk = (unsigned)a.Npad * (HBK * 2);        // bytes of one (chunk, tap) panel of a weight plane ([chunk][Npad][32])

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);

    const int n_cc = a.Cin / 32 / KG;                          // channel chunks of this group: kg, kg + KG, ...
    const int nsteps = 9 * n_cc;

    // ---- per-lane validity of the 9 taps for the TM 16-row tiles this wave reads
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / NWN, wn = wave - wm * NWN;
    const int lr = lane & 15, lh = lane >> 4;
    unsigned vmask[TM];
    int prow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = wm * WM + i * 16 + lr;
        prow[i] = p;
        const int m = m0 + p;
        unsigned vm = 0;
        if (m < M) {
            const int hw = H * W;
            const int b = m / hw, r = m - b * hw;
            const int oy = r / W, ox = r - oy * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) vm |= 1u << t;
            }
        }
        vmask[i] = vm;
    }

    struct BStage { u32x4 bh[B_SLOTS], bl[B_SLOTS]; };
    BStage S0, S1;
    u32x4 BRh[BAND_SLOTS], BRl[BAND_SLOTS];

    int ld_step = 0, ld_cc = 0, ld_tap = 0;                    // B chunk to be loaded next: local step, its chunk and tap
    auto gload_b = [&](BStage& S) {
        const bool live = ld_step < nsteps;
        const unsigned koff = (unsigned)((ld_cc * KG + kg) * 9 + ld_tap) * wchunk;
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const unsigned wo = live ? wbase[i] : OOB;
            S.bh[i] = asm_buffer_load_b128(rs_wh, wo, koff);
            S.bl[i] = asm_buffer_load_b128(rs_wl, wo, koff);
        }
        ++ld_step;
        if (++ld_tap == 9) { ld_tap = 0; ++ld_cc; }
    };
    auto gload_band = [&](int cc) {
        const bool live = cc < n_cc;
        const unsigned soff = (unsigned)(cc * KG + kg) * 64u;
#pragma unroll
        for (int j = 0; j < BAND_SLOTS; ++j) {
            const unsigned vo = live ? bvo[j] : OOB;
            BRh[j] = asm_buffer_load_b128(rs_a, vo, soff);
            BRl[j] = asm_buffer_load_b128(rs_a, vo, lo_plane + soff);
        }
    };
    // vmcnt bookkeeping (in-order completion).  Steady-state issue pattern: [B set x][B set y] and, right after a
    // channel-chunk boundary, [B x][B y][band].  wait_b: the older B set has landed, the younger B set (and for the
    // two steps after a band prefetch the band loads in between) may stay in flight.  The counted wait carries no
    // register operands (two alternative asm statements with tied operands make the compiler unify their outputs
    // with copies placed BEFORE the wait); one tying statement after the uniform branch pins the uses instead.
    // (Round 3: a third register set — weights loaded three steps ahead — changed nothing, A/B on one box: what the weight
    // loads cost, 14-21 % of these kernels in the timing-only builds of tools/run_abl.sh, is throughput, not latency.)
    int band_age = 0;                                          // steps since the last band prefetch was issued
    auto wait_b = [&](BStage& S) {
        if (band_age < 2) wait_vmcnt<B_LOADS + BAND_LOADS>(); else wait_vmcnt<B_LOADS>();
        tie_regs(S.bh); tie_regs(S.bl);
        ++band_age;
        __builtin_amdgcn_sched_barrier(0);
    };
    auto wait_band = [&]() {                                   // everything issued so far except the two B sets
        wait_vmcnt<2 * B_LOADS>();
        tie_regs(BRh); tie_regs(BRl);
        __builtin_amdgcn_sched_barrier(0);
    };
    const int wr_swz = (c16 ^ band_swz(row0)) << 4;            // RPP % 8 == 0: the same for every pass
    auto write_b = [&](const BStage& S, int buf) {
        unsigned char* st = bst + buf * BSTAGE;
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int o = (row0 + i * RPP) * 64 + wr_swz;
            if ((i + 1) * RPP <= BN || row0 + i * RPP < BN) {
                *reinterpret_cast<u32x4*>(st + o) = S.bh[i];
                *reinterpret_cast<u32x4*>(st + PANEL_B + o) = S.bl[i];
            }
        }
    };
    auto write_band = [&]() {
#pragma unroll
        for (int j = 0; j < BAND_SLOTS; ++j) {
            const int o = (row0 + j * RPP) * 64 + wr_swz;
            if (row0 + j * RPP < NBR) {                        // never touch the zero row
                *reinterpret_cast<u32x4*>(bandh + o) = BRh[j];
                *reinterpret_cast<u32x4*>(bandl + o) = BRl[j];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int b_lane = (wn * WN + lr) * 64 + ((lh ^ band_swz(lr)) << 4);     // WN % 16 == 0: the row's swizzle is the lane's
    f16x8 ah[TM], al[TM];
    // Branch-free: the offset of a valid tap and the zero row are merged by a bit select (bit `tap` of vmask, sign-extended,
    // is the mask) — the divergent-branch form cost 6 registers and a basic-block split per 16-row tile.
    const int lh4 = lh << 4;
    auto read_a = [&](int tap) {
        const int shift = (tap / 3) * W + (tap % 3);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = prow[i] + shift;
            const int x = (row << 6) | (((row << 3) ^ lh4) & 0x30);          // row * 64 + ((lh ^ band_swz(row)) << 4)
            const int sel = __builtin_amdgcn_sbfe((int)vmask[i], tap, 1);     // 0 or -1
            const int o = (x & sel) | (zero_off & ~sel);
            ah[i] = *reinterpret_cast<const f16x8*>(bandh + o);
            al[i] = *reinterpret_cast<const f16x8*>(bandl + o);
        }
    };
    // per 16-column group: two B fragments, then 3 TM products (lo*hi, hi*lo, hi*hi: small terms first)
    auto compute = [&](int buf) {
        const unsigned char* st = bst + buf * BSTAGE + b_lane;
        f16x8 bh[TN], bl[TN];
#if RTOD_BFRAG_AHEAD
        // B fragments LA = 2 column groups ahead of their products: a read is covered by the 3 TM MFMAs of the group before it
        // (left to itself the scheduler reads each pair just in time — 8 registers of fragments, a wait in front of every MFMA
        // pair).  A/B on one box, 76x76 / 38x38 layers: -3.3 % / -2.0 %; LA = 1: as before; LA = 3 or all four groups up front
        // on the 12-wave tile (registers allow it there): +2-3 % at 38x38; the first two groups at the top of the step, ahead of
        // the weight wait: no change; s_setprio 1 around the MFMA block: +2-7 %  (profiles/experiments/r03_bfrag_ahead.log).
        constexpr int LA = RTOD_BFRAG_LA(MINW, TN) < TN ? RTOD_BFRAG_LA(MINW, TN) : TN;     // column groups read ahead
        auto read_bj = [&](int j) {
            bh[j] = *reinterpret_cast<const f16x8*>(st + j * 16 * 64);
            bl[j] = *reinterpret_cast<const f16x8*>(st + PANEL_B + j * 16 * 64);
        };
#pragma unroll
        for (int j = 0; j < LA; ++j) read_bj(j);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * LA, 0);
#else
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const f16x8*>(st + j * 16 * 64);
            bl[j] = *reinterpret_cast<const f16x8*>(st + PANEL_B + j * 16 * 64);
        }
#endif
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (BAND_TR) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
#if RTOD_BFRAG_AHEAD
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * TM, 0);
            if (j + LA < TN) { read_bj(j + LA); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#endif
        }
    };

#ifdef RTOD_STAMPS
    unsigned long long ts_[STAMP_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart_ = tprev_;
    const unsigned long long rstart_ = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- prologue.  Issue order (vmcnt is in-order): band(0), B0, B1 | band written, B0 staged | B2, band(1)
    gload_band(0);
    gload_b(S0);
    gload_b(S1);
    wait_band();                                               // band(0) landed (B0, B1 may be in flight)
    write_band();
    band_age = 2;
    wait_b(S0);                                                // B0 landed (B1 in flight)
    write_b(S0, 0);
    gload_b(S0);                                               // B chunk 2
    gload_band(1);                                             // prefetch of the next channel chunk (OOB if none)
    band_age = 0;                                              // same issue pattern as at a chunk boundary: [B, B, band]
    __syncthreads();

    RTOD_STAMP(0)                                              // 0: prologue
    // Main loop above epilogues in the issue arbitration (back to 0 after the drain): -1.2...1.5 % on every band layer class on
    // one box, -0.1...0.5 % on another (tools/exp_ab_libs.py, profiles/experiments/r03_loop_priority_ab.log); levels 1 and 3 the same,
    // the same two lines in the generic and ring kernels nothing.
    __builtin_amdgcn_s_setprio(2);
    int tap = 0, cc = 0;
    // one step = one (channel chunk, tap): compute chunk t from B buffer t&1, stage chunk t+1, load chunk t+3.
    // After the last tap of a channel chunk the band is replaced (all waves have read it: the step's barrier).
    auto step = [&](int buf, BStage& Snext) {
#ifdef RTOD_STAMPS
        const int age_ = band_age;
#endif
        read_a(tap);
        wait_b(Snext);
#ifdef RTOD_STAMPS
        // vmcnt retires in order: the B set waited for 2 steps after a band prefetch was issued BEHIND those band loads (slot 8)
        if (age_ == 2) RTOD_STAMP(8) else if (age_ < 2) RTOD_STAMP(9) else
#endif
        RTOD_STAMP(1)                                          // 1: A reads issued + wait for the staged B set
        write_b(Snext, buf ^ 1);
        gload_b(Snext);
        __builtin_amdgcn_sched_barrier(0);
        RTOD_STAMP(2)                                          // 2: B LDS writes + next loads issued
        compute(buf);
        RTOD_STAMP(3)                                          // 3: B reads + MFMA issue
        __syncthreads();
        RTOD_STAMP(4)                                          // 4: barrier
        if (++tap == 9) {
            tap = 0; ++cc;
            if (cc < n_cc) {                                   // uniform
                wait_band();                                   // band(cc) landed long ago; keeps the two B sets in flight
                write_band();
                gload_band(cc + 1);
                band_age = 0;
                __syncthreads();
                RTOD_STAMP(5)                                  // 5: band replacement
            }
        }
    };
    for (int t = 0; t < nsteps; t += 2) {
        step(0, S1);
        if (t + 1 < nsteps) step(1, S0);
    }
    // drain: the trailing loads' destination registers must stay allocated until they have landed (see conv_igemm_f16s3.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie_regs(S0.bh); tie_regs(S0.bl); tie_regs(S1.bh); tie_regs(S1.bl); tie_regs(BRh); tie_regs(BRl);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    RTOD_STAMP(6)                                              // 6: drain
    __builtin_amdgcn_s_setprio(0);

#ifdef RTOD_DIAG
    if (a.dbg & 4) return;
#endif
    if constexpr (BAND_TR) {
        int mrow[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int m = bm * BM + wm * WM + i * 16 + lr; mrow[i] = m < M ? m : -1; }
        conv_f16s3_epilogue_regs<WM, WN, EPI == EPI_SPLIT_RES, KG>(a, acc, smem, mrow, bn * BN + wn * WN, tid, lh, kg);
    } else conv_f16s3_epilogue<BM, BN, WM, WN, NT * KG, EPI, band_epi_bytes(BM, BN), KG>(a, acc, smem, bm, bn, (int)threadIdx.x, wm, wn, lr, lh, M, kg);
#ifdef RTOD_STAMPS
    RTOD_STAMP(7)                                              // 7: epilogue
    if ((threadIdx.x & 63) == 0 && blockIdx.x < STAMP_BLOCKS && (threadIdx.x >> 6) < STAMP_WAVES) {
        unsigned long long* o = g_band_stamps + (blockIdx.x * STAMP_WAVES + (threadIdx.x >> 6)) * (STAMP_SLOTS + 1);
        for (int i = 0; i < STAMP_SLOTS; ++i) o[i] = ts_[i];
        o[STAMP_SLOTS] = tprev_ - tstart_;
        g_band_real[blockIdx.x * STAMP_WAVES + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime() - rstart_;
    }
#endif
#ifdef RTOD_TIMELINE
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < TL_BLOCKS) {
        g_band_tl[blockIdx.x * 4] = tl_start_;
        g_band_tl[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        g_band_tl[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);   // HW_ID, XCC_ID
        g_band_tl[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - tl_cstart_;
    }
#endif
}

template <int BM, int BN, int NWM, int NWN, int MINW, int KG = 1>
static int launch_band(const ConvArgs& a, hipStream_t s) {
    const int M = a.B * a.Ho * a.Wo;
    const int gm = (M + BM - 1) / BM, gn = (a.Cout + BN - 1) / BN;
    if (a.Cin % (32 * KG)) { set_error("launch_conv_band: Cin=%d not a multiple of %d", a.Cin, 32 * KG); return RTOD_E_ARG; }
    // Which operand should an XCD's L2 (4 MiB) keep?  Each XCD streams its workgroups' operands from memory once.
    // Pixel-tile-major: an XCD reads 1/8 of the activations and ALL weights; channel-tile-major (possible when the N
    // tiles split evenly over the 8 XCDs): 1/8 of the weights and all activations.  The deep 19x19 layers hold 18.9 MB of
    // weights against 5.9 MB of activations: their main loop waited on weight loads 50 % longer than the other scales'.
    ConvArgs ax = a;
    ax.xcd_by_n = (gn % 8 == 0 && (int64_t)a.Cout * a.K > (int64_t)M * a.Cin) ? 1 : 0;
    const int main_bytes = KG * (4 * BN * 64 + 2 * (band_rows(BM, a.Wi) + 1) * 64);
    const int lds = main_bytes > band_epi_bytes(BM, BN) ? main_bytes : band_epi_bytes(BM, BN);
    auto k_res = conv_band_f16s3_kernel<BM, BN, NWM, NWN, MINW, EPI_SPLIT_RES, KG>;
    auto k_plain = conv_band_f16s3_kernel<BM, BN, NWM, NWN, MINW, EPI_SPLIT, KG>;
    static std::atomic<unsigned long long> attr_done{0};       // per instantiation, one bit per device; > 64 KiB of dynamic LDS needs the opt-in
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError(), "conv_band_f16s3 hipGetDevice");
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {       // idempotent: a second thread may repeat the calls
        const int cap = KG * (4 * BN * 64 + 2 * (band_rows(BM, BAND_MAX_W) + 1) * 64);
        const int mx = cap > band_epi_bytes(BM, BN) ? cap : band_epi_bytes(BM, BN);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_res), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_plain), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
            return hip_fail(hipGetLastError(), "conv_band_f16s3 LDS attribute");
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    if (a.res) hipLaunchKernelGGL(k_res, dim3(gm * gn), dim3(NWM * NWN * 64 * KG), lds, s, ax, gm, gn);
    else hipLaunchKernelGGL(k_plain, dim3(gm * gn), dim3(NWM * NWN * 64 * KG), lds, s, ax, gm, gn);
#ifdef RTOD_TIMELINE
    {   // diagnostic build: synchronises, prints the launch's workgroup timeline
        static int printed = 0;
        const int nb = gm * gn < TL_BLOCKS ? gm * gn : TL_BLOCKS;
        if (printed < 60 && hipDeviceSynchronize() == hipSuccess) {
            static unsigned long long h[TL_BLOCKS * 4];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_band_tl), sizeof(unsigned long long) * 4 * nb) == hipSuccess) {
                unsigned long long t0 = ~0ull, t1 = 0;
                double cyc = 0, rt = 0;
                for (int b = 0; b < nb; ++b) {
                    if (h[b * 4] < t0) t0 = h[b * 4];
                    if (h[b * 4 + 1] > t1) t1 = h[b * 4 + 1];
                    cyc += (double)h[b * 4 + 3]; rt += (double)(h[b * 4 + 1] - h[b * 4]);
                }
                // first round = started within 2 us of the first start
                double d1 = 0, d2 = 0, s2min = 1e9, s2max = 0, e1min = 1e9, e1max = 0; int n1 = 0, n2 = 0;
                for (int b = 0; b < nb; ++b) {
                    const double st = (h[b * 4] - t0) / 100.0, en = (h[b * 4 + 1] - t0) / 100.0;
                    if (st < 2.0) { ++n1; d1 += en - st; if (en < e1min) e1min = en; if (en > e1max) e1max = en; }
                    else { ++n2; d2 += en - st; if (st < s2min) s2min = st; if (st > s2max) s2max = st; }
                }
                fprintf(stderr, "[timeline] band<%d,%d,%dx%d,k%d> W=%d Cin=%d Cout=%d res=%d tiles=%d | span %.1f us | round 1: %d wgs, mean %.1f us, ends %.1f..%.1f | later: %d wgs, starts %.1f..%.1f, mean %.1f us | clock %.0f MHz\n",
                        BM, BN, NWM, NWN, KG, a.Wi, a.Cin, a.Cout, a.res ? 1 : 0, gm * gn, (t1 - t0) / 100.0, n1, n1 ? d1 / n1 : 0.0, e1min, e1max, n2, n2 ? s2min : 0.0, n2 ? s2max : 0.0, n2 ? d2 / n2 : 0.0,
                        rt > 0 ? cyc / rt * 100.0 : 0.0);
                if (printed < 2) for (int b = 0; b < nb; b += 37) {
                    const unsigned w = (unsigned)h[b * 4 + 2];
                    fprintf(stderr, "[timeline]   wg %4d xcc %u se %u cu %2u tg %u  %.1f .. %.1f us\n", b, (unsigned)(h[b * 4 + 2] >> 32), (w >> 13) & 7, (w >> 8) & 15, (w >> 16) & 15, (h[b * 4] - t0) / 100.0, (h[b * 4 + 1] - t0) / 100.0);
                }
                ++printed;
            }
        }
    }
#endif
#ifdef RTOD_STAMPS
    {   // print the mean cycles per phase over the first blocks' waves (diagnostic build: synchronises)
        static int printed = 0;
        if (printed < 400 && hipDeviceSynchronize() == hipSuccess) {
            static unsigned long long h[STAMP_BLOCKS * STAMP_WAVES * (STAMP_SLOTS + 1)];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_band_stamps), sizeof(h)) == hipSuccess) {
                const int nb = gm * gn < STAMP_BLOCKS ? gm * gn : STAMP_BLOCKS, nw = NWM * NWN * KG;
                double sum[STAMP_SLOTS + 1] = {0};
                for (int b = 0; b < nb; ++b) for (int w = 0; w < nw; ++w) for (int i = 0; i <= STAMP_SLOTS; ++i)
                    sum[i] += (double)h[(b * STAMP_WAVES + w) * (STAMP_SLOTS + 1) + i];
                fprintf(stderr, "[stamps] band<%d,%d,%dx%d,k%d> W=%d Cin=%d Cout=%d tiles=%d steps=%d | cycles/wave:", BM, BN, NWM, NWN, KG, a.Wi, a.Cin, a.Cout, gm * gn, 9 * a.Cin / 32 / KG);
                for (int i = 0; i <= STAMP_SLOTS; ++i) fprintf(stderr, " %s%.0f", i == STAMP_SLOTS ? "total=" : "", sum[i] / (nb * nw));
                static unsigned long long hr[STAMP_BLOCKS * STAMP_WAVES];
                if (hipMemcpyFromSymbol(hr, HIP_SYMBOL(g_band_real), sizeof(hr)) == hipSuccess) {
                    double real = 0;
                    for (int b = 0; b < nb; ++b) for (int w = 0; w < nw; ++w) real += (double)hr[b * STAMP_WAVES + w];
                    if (real > 0) fprintf(stderr, " | clock %.0f MHz", sum[STAMP_SLOTS] / real * 100.0);     // s_memtime / s_memrealtime x 100 MHz
                }
                fprintf(stderr, "\n");
                ++printed;
            }
        }
    }
#endif
    return hip_fail(hipGetLastError(), "conv_band_f16s3 launch");
}

// Split-K changes the summation order, and a layer's result must not depend on the batch it runs in (tile choice does):
// whether a layer runs split-K is therefore a property of the layer alone — deep (Cin >= 512) and small (<= 400 pixels:
// the 19x19 / 13x13 stages, whose grid cannot give a CU two workgroups at any realistic batch).
int conv_band_layer_kg(int cin, int h, int w) { return (cin % 64 == 0 && cin >= 512 && h * w <= 400) ? 2 : 1; }
bool conv_band_mode_valid(int mode, int cin, int h, int w) {
    if (mode < 0 || mode >= BAND_MODES) return false;
    if (mode == BANDD_WIDE_MODE) return false;                     // the wide tile belongs to the non-band layers (94 < W <= 160)
    if (mode >= BAND_LDS_MODES) return conv_bandd_mode_kg(mode - BAND_LDS_MODES) == conv_band_layer_kg(cin, h, w);
    return (mode >= BAND_K2_MODE0 ? 2 : 1) == conv_band_layer_kg(cin, h, w);
}
int conv_band_default_mode(int cin, int h, int w) { return conv_band_layer_kg(cin, h, w) == 2 ? BAND_K2_MODE0 : 0; }

bool conv_band_supported(int ksize, int stride, int pad, int cin, int w_in) {
    return ksize == 3 && stride == 1 && pad == 1 && cin % 32 == 0 && w_in <= BAND_MAX_W;
}

// One list drives the mode table, the launch switch and the kernel names rocprofv3 prints:
//   X(mode, BM, BN, waves along M, waves along N, MINW, K groups, name suffix)
#define RTOD_BAND_TILES(X) \
    X(0, 128, 128, 4, 2, 4, 1, "") X(1, 128, 64, 4, 2, 4, 1, "") X(2, 192, 128, 4, 2, 2, 1, "") X(3, 192, 128, 6, 2, 3, 1, "") \
    X(4, 96, 128, 2, 4, 4, 1, "") X(5, 128, 128, 2, 2, 2, 1, "") X(6, 64, 128, 2, 4, 4, 1, "") \
    X(7, 96, 128, 2, 4, 4, 2, ",k2") X(8, 128, 128, 4, 2, 4, 2, ",k2") X(9, 64, 128, 2, 4, 4, 2, ",k2") X(10, 128, 64, 4, 2, 4, 2, ",k2")

#define RTOD_X_INFO(mode, bm, bn, nwm, nwn, minw, kg, sfx) {bm, bn, "conv_band_f16s3<" #bm "x" #bn "," #nwm "x" #nwn sfx ">"},
static const ConvVariantInfo kBandModes[BAND_LDS_MODES] = { RTOD_BAND_TILES(RTOD_X_INFO) };
#undef RTOD_X_INFO

int conv_band_kernel_name(int mode, int epi, char* buf, size_t len) {
    if (mode >= BAND_LDS_MODES) return conv_bandd_kernel_name(mode - BAND_LDS_MODES, epi, buf, len);
#define RTOD_X_NAME(m, bm, bn, nwm, nwn, minw, kg, sfx) \
    if (mode == m) return snprintf(buf, len, "void rtod::conv_band_f16s3_kernel<" #bm ", " #bn ", " #nwm ", " #nwn ", " #minw ", %d, " #kg ">(rtod::ConvArgs, int, int)", epi);
    RTOD_BAND_TILES(RTOD_X_NAME)
#undef RTOD_X_NAME
    return -1;
}

const ConvVariantInfo& conv_band_mode_info(int mode) {
    if (mode >= BAND_LDS_MODES && mode < BAND_MODES) return conv_bandd_mode_info(mode - BAND_LDS_MODES);
    return kBandModes[mode < 0 || mode >= BAND_LDS_MODES ? 0 : mode];
}

int launch_conv_band_f16s3(const ConvArgs& a_in, int mode, hipStream_t s) {
    if (mode >= BAND_LDS_MODES && mode < BAND_MODES) return launch_conv_bandd_f16s3(a_in, mode - BAND_LDS_MODES, s);
    ConvArgs a = a_in;
    if (!a.in || !a.w_hi || !a.w_lo || !a.bias || !a.inv_scale || !a.out) { set_error("launch_conv_band: null pointer"); return RTOD_E_ARG; }
    if (!conv_band_supported(a.kh, a.stride, a.pad, a.Cin, a.Wi) || a.kw != 3 || a.Ho != a.Hi || a.Wo != a.Wi || a.dec.enabled) {
        set_error("launch_conv_band: unsupported shape (k=%d s=%d pad=%d Cin=%d W=%d)", a.kh, a.stride, a.pad, a.Cin, a.Wi); return RTOD_E_ARG;
    }
    if (a.in_ldc % 8 || a.in_coff % 8 || a.K != a.Kpad || a.K != 9 * a.Cin) { set_error("launch_conv_band: bad view / K"); return RTOD_E_ARG; }
    if (a.in_bytes == 0 || a.in_bytes >= OOB || a.w_bytes == 0 || a.w_bytes >= OOB) { set_error("launch_conv_band: buffer extents"); return RTOD_E_ARG; }
    if ((uint64_t)a.B * a.Hi * a.Wi * a.in_ldc * 4ull > (uint64_t)a.in_bytes) { set_error("launch_conv_band: input view exceeds its buffer"); return RTOD_E_ARG; }
#ifdef RTOD_DIAG
    static const int dbg_zero = getenv("RTOD_DBG_ZERO") ? atoi(getenv("RTOD_DBG_ZERO")) : 0;   // diagnostic build only
    if (dbg_zero & 1) a.in_bytes = 1;
    if (dbg_zero & 2) a.w_bytes = 1;
    a.dbg = dbg_zero;
#endif
    switch (mode) {
#define RTOD_X_CASE(m, bm, bn, nwm, nwn, minw, kg, sfx) case m: return launch_band<bm, bn, nwm, nwn, minw, kg>(a, s);
        RTOD_BAND_TILES(RTOD_X_CASE)
#undef RTOD_X_CASE
    }
    set_error("launch_conv_band: mode %d unsupported", mode);
    return RTOD_E_ARG;
}

}  // namespace rtod
