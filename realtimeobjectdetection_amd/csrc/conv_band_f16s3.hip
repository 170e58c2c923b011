// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input band (split-precision f16 MFMA), gfx950.
//
// Same contract as conv_igemm_f16s3.hip for the layers it covers (reference: conv -> BN -> leaky,
// src/darknet.py:467-501, shortcut fused, 263-268).  The generic implicit-GEMM re-gathers every input
// pixel once per tap: 9x the L2->CU traffic and 9x the LDS writes for the A operand (measured: 900 MB of
// L1->L2 requests per launch for a 72 MB layer, ~1.3 ms of load stall over the 3x3 layers of YOLOv3).
// Here a workgroup owns BM = 128 CONSECUTIVE output pixels (linear index over batch, y, x) and stages,
// once per 32-channel chunk, the band of input pixels those outputs can touch:
//
//     band row r  <->  input pixel (linear)  m0 - W - 1 + r,   r in [0, 128 + 2W + 2)
//
// so the A fragment of output row p for tap (ky,kx) is band row p + ky*W + kx: a constant shift per tap,
// read straight from LDS (im2col never materialised).  Taps that fall outside the image (left/right edge,
// top/bottom, other image of the batch) are redirected per lane to an all-zero LDS row by a 9-bit validity
// mask computed once.  Consecutive pixels on consecutive lanes keep the 16-byte XOR swizzle conflict-free
// for every shift.  K order is (channel chunk outer, tap inner): the weight planes are packed to match.
// B (weights) is streamed per (chunk, tap) exactly like the generic kernel: double-buffered LDS, two
// register stage sets, asm buffer loads with counted vmcnt.
//
// Limits: stride 1, pad 1, 3x3, Cin % 32 == 0, W <= 94 (band <= 320 rows = 40 KiB).
#include "conv_f16s3_common.h"
#include <cstdlib>

namespace rtod {

constexpr int BAND_ROWS = 320;                 // max band rows (128 + 2W + 2 <= 320)
constexpr int BAND_ZERO = 320;                 // index of the all-zero row
constexpr int BAND_PANEL = 336 * 64;           // bytes per plane (rows 321..335 unused padding)

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 14 && N % 2 == 0, "vmcnt literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
}
template <typename T, int N> __device__ __forceinline__ void tie_regs(T (&r)[N]) {   // pins later uses below a preceding wait
    static_assert(N >= 1 && N <= 5, "tie_regs");
    if constexpr (N == 1) asm volatile("" : "+v"(r[0]) :: "memory");
    else if constexpr (N == 2) asm volatile("" : "+v"(r[0]), "+v"(r[1]) :: "memory");
    else if constexpr (N == 3) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]) :: "memory");
    else if constexpr (N == 4) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
    else asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) :: "memory");
}

// BM = output pixels per workgroup, NWM = waves along M (x 2 waves along N).
//   BM 128, NWM 2: 4 waves of 64 x BN/2 (2 waves/SIMD)
//   BM 128, NWM 4: 8 waves of 32 x BN/2 at 4 waves/SIMD — tools/ubench_tiles.hip: occupancy buys more MFMA
//                  utilisation than a larger wave tile
//   BM  96, NWM 3: 6 waves of 32 x BN/2 at 3 waves/SIMD — a tile height that fills whole rounds of the 512 resident
//                  workgroup slots where 128 does not (38x38x8 x 512 ch: 484 tiles instead of 364)
constexpr int band_min_waves(int nwm) { return nwm == 4 ? 4 : (nwm == 3 ? 3 : 2); }

template <int BM, int BN, int NWM, int EPI>
__global__ __launch_bounds__(NWM * 128, band_min_waves(NWM))
void conv_band_f16s3_kernel(const ConvArgs a, const int grid_m, const int grid_n) {
    constexpr int WM = BM / NWM, WN = BN / 2, NT = NWM * 128;
    static_assert(WM % 32 == 0 && BM + 2 * 94 + 2 <= BAND_ROWS + 64, "band tile");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPP = NT / 4;                               // rows per pass (4 x 16-B chunks per row)
    constexpr int B_SLOTS = (BN + RPP - 1) / RPP;
    constexpr int BAND_SLOTS = (BAND_ROWS + RPP - 1) / RPP;   // 5 (4 waves) / 3 (8 waves)
    static_assert(RPP % 16 == 0, "a pass that runs past a panel is predicated per 16-row wave slice");
    constexpr int B_LOADS = 2 * B_SLOTS, BAND_LOADS = 2 * BAND_SLOTS;
    constexpr int PANEL_B = BN * 64;
    constexpr int BSTAGE = 2 * PANEL_B;
    constexpr int SMEM = 2 * BAND_PANEL + 2 * BSTAGE;
    static_assert(SMEM >= WM * BN * 4, "one epilogue row group must fit");

    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* bandh = smem;
    unsigned char* bandl = smem + BAND_PANEL;
    unsigned char* bst = smem + 2 * BAND_PANEL;

    const int nwg = grid_m * grid_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int bm = bid / grid_n, bn = bid - bm * grid_n;

    const int tid = threadIdx.x;
    const int W = a.Wi, H = a.Hi;
    const int M = a.B * H * W;                                 // Ho == Hi, Wo == Wi
    const int m0 = bm * BM;
    const int NB = BM + 2 * W + 2;
    const int c16 = tid & 3, row0 = tid >> 2;
    const unsigned PS = (unsigned)a.in_ldc * 4u;
    const unsigned lo_plane = (unsigned)a.in_ldc * 2u;

    // zero row (both planes)
    if (tid < 8) *reinterpret_cast<u32x4*>((tid < 4 ? bandh : bandl) + BAND_ZERO * 64 + (tid & 3) * 16) = u32x4{0u, 0u, 0u, 0u};

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
        wbase[i] = (row0 + i * RPP < BN) ? (unsigned)((bn * BN + row0 + i * RPP) * a.Kpad + c16 * 8) * 2u : OOB;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);

    const int n_cc = a.Cin / 32;
    const int nsteps = 9 * n_cc;

    // ---- per-lane validity of the 9 taps for the two 32-row tiles this wave reads
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;                      // NWM x 2 waves
    const int lr = lane & 31, lh = lane >> 5;
    unsigned vmask[TM];
    int prow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = wm * WM + i * 32 + lr;
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

    int ld_step = 0;                                           // B chunk to be loaded next (== (cc*9 + tap))
    auto gload_b = [&](BStage& S) {
        const bool live = ld_step < nsteps;
        const unsigned koff = (unsigned)ld_step * (HBK * 2);
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const unsigned wo = live ? wbase[i] : OOB;
            S.bh[i] = asm_buffer_load_b128(rs_wh, wo, koff);
            S.bl[i] = asm_buffer_load_b128(rs_wl, wo, koff);
        }
        ++ld_step;
    };
    auto gload_band = [&](int cc) {
        const bool live = cc < n_cc;
        const unsigned soff = (unsigned)cc * 64u;
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
    const int wr_swz = (c16 ^ ((row0 >> 2) & 3)) << 4;
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
            if ((j + 1) * RPP <= BAND_ROWS || row0 + j * RPP < BAND_ROWS) {     // never touch the zero row / padding
                *reinterpret_cast<u32x4*>(bandh + o) = BRh[j];
                *reinterpret_cast<u32x4*>(bandl + o) = BRl[j];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int rd_swz = (lr >> 2) & 3;
    const int b_row = (wn * WN + lr) * 64;
    // 4-wave blocks read the fragments of both k16 steps up front (64 VGPRs); 8-wave blocks must stay within
    // 128 VGPRs for 4 waves/SIMD and read one k16 step at a time (the other waves of the SIMD cover the latency)
    constexpr int FK = (NWM >= 3) ? 1 : 2;
    struct Frags { f16x8 ah[FK][TM], al[FK][TM], bh[FK][TN], bl[FK][TN]; };
    auto read_frags = [&](int tap, int buf, int ks0, Frags& F) {
        const int shift = (tap / 3) * W + (tap % 3);
        const unsigned char* st = bst + buf * BSTAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = prow[i] + shift;
            const bool ok = (vmask[i] >> tap) & 1u;
            const int swz = (row >> 2) & 3;
#pragma unroll
            for (int k = 0; k < FK; ++k) {
                const int o = ok ? row * 64 + ((((ks0 + k) * 2 + lh) ^ swz) << 4) : BAND_ZERO * 64;
                F.ah[k][i] = *reinterpret_cast<const f16x8*>(bandh + o);
                F.al[k][i] = *reinterpret_cast<const f16x8*>(bandl + o);
            }
        }
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            const int co = (((ks0 + k) * 2 + lh) ^ rd_swz) << 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                F.bh[k][j] = *reinterpret_cast<const f16x8*>(st + b_row + j * 32 * 64 + co);
                F.bl[k][j] = *reinterpret_cast<const f16x8*>(st + PANEL_B + b_row + j * 32 * 64 + co);
            }
        }
    };
    auto mfma_block = [&](const Frags& F) {
#pragma unroll
        for (int k = 0; k < FK; ++k)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.al[k][i], F.bh[k][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.ah[k][i], F.bl[k][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.ah[k][i], F.bh[k][j], acc[i][j], 0, 0, 0);
                }
    };

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

    Frags F;
    int tap = 0, cc = 0;
    // one step = one (channel chunk, tap): compute chunk t from B buffer t&1, stage chunk t+1, load chunk t+3.
    // After the last tap of a channel chunk the band is replaced (all waves have read it: the step's barrier).
    auto step = [&](int buf, BStage& Snext) {
        read_frags(tap, buf, 0, F);
        wait_b(Snext);
        write_b(Snext, buf ^ 1);
        gload_b(Snext);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(F);
        if constexpr (FK == 1) { read_frags(tap, buf, 1, F); mfma_block(F); }
        __syncthreads();
        if (++tap == 9) {
            tap = 0; ++cc;
            if (cc < n_cc) {                                   // uniform
                wait_band();                                   // band(cc) landed long ago; keeps the two B sets in flight
                write_band();
                gload_band(cc + 1);
                band_age = 0;
                __syncthreads();
            }
        }
    };
    for (int t = 0; t < nsteps; t += 2) {
        step(0, S1);
        if (t + 1 < nsteps) step(1, S0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (a.dbg & 4) return;
    conv_f16s3_epilogue<BM, BN, WM, WN, NT, EPI, SMEM>(a, acc, smem, bm, bn, tid, wm, wn, lr, lh, M);
}

template <int BM, int BN, int NWM>
static int launch_band(const ConvArgs& a, hipStream_t s) {
    const int M = a.B * a.Ho * a.Wo;
    const int gm = (M + BM - 1) / BM, gn = (a.Cout + BN - 1) / BN;
    if (a.res)
        hipLaunchKernelGGL((conv_band_f16s3_kernel<BM, BN, NWM, EPI_SPLIT_RES>), dim3(gm * gn), dim3(NWM * 128), 0, s, a, gm, gn);
    else
        hipLaunchKernelGGL((conv_band_f16s3_kernel<BM, BN, NWM, EPI_SPLIT>), dim3(gm * gn), dim3(NWM * 128), 0, s, a, gm, gn);
    return hip_fail(hipGetLastError(), "conv_band_f16s3 launch");
}

bool conv_band_supported(int ksize, int stride, int pad, int cin, int w_in) {
    return ksize == 3 && stride == 1 && pad == 1 && cin % 32 == 0 && 128 + 2 * w_in + 2 <= BAND_ROWS;   // BM <= 128
}

// mode: 0 128x128/4w, 1 128x64/4w, 2 128x128/8w, 3 128x64/8w, 4 96x128/6w, 5 96x64/6w
int launch_conv_band_f16s3(const ConvArgs& a_in, int mode, hipStream_t s) {
    ConvArgs a = a_in;
    if (!a.in || !a.w_hi || !a.w_lo || !a.bias || !a.inv_scale || !a.out) { set_error("launch_conv_band: null pointer"); return RTOD_E_ARG; }
    if (!conv_band_supported(a.kh, a.stride, a.pad, a.Cin, a.Wi) || a.kw != 3 || a.Ho != a.Hi || a.Wo != a.Wi || a.dec.enabled) {
        set_error("launch_conv_band: unsupported shape (k=%d s=%d pad=%d Cin=%d W=%d)", a.kh, a.stride, a.pad, a.Cin, a.Wi); return RTOD_E_ARG;
    }
    if (a.in_ldc % 8 || a.in_coff % 8 || a.K != a.Kpad || a.K != 9 * a.Cin) { set_error("launch_conv_band: bad view / K"); return RTOD_E_ARG; }
    if (a.in_bytes == 0 || a.in_bytes >= OOB || a.w_bytes == 0 || a.w_bytes >= OOB) { set_error("launch_conv_band: buffer extents"); return RTOD_E_ARG; }
    if ((uint64_t)a.B * a.Hi * a.Wi * a.in_ldc * 4ull > (uint64_t)a.in_bytes) { set_error("launch_conv_band: input view exceeds its buffer"); return RTOD_E_ARG; }
    static const int dbg_zero = getenv("RTOD_DBG_ZERO") ? atoi(getenv("RTOD_DBG_ZERO")) : 0;
    if (dbg_zero & 1) a.in_bytes = 1;
    if (dbg_zero & 2) a.w_bytes = 1;
    a.dbg = dbg_zero;
    if (mode == 0) return launch_band<128, 128, 2>(a, s);
    if (mode == 1) return launch_band<128, 64, 2>(a, s);
    if (mode == 2) return launch_band<128, 128, 4>(a, s);
    if (mode == 3) return launch_band<128, 64, 4>(a, s);
    if (mode == 4) return launch_band<96, 128, 3>(a, s);
    if (mode == 5) return launch_band<96, 64, 3>(a, s);
    set_error("launch_conv_band: mode %d unsupported", mode);
    return RTOD_E_ARG;
}

}  // namespace rtod
