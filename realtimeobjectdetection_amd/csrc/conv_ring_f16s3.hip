// Persistent implicit-GEMM convolution with an LDS-DMA stage ring (split-precision f16 MFMA), gfx950.
//
// Same contract, GEMM view, data formats, K order and epilogues as conv_igemm_f16s3.hip (reference: conv -> BN(eval)
// -> leaky, src/darknet.py:467-501; shortcut 263-268; head decode src/util.py:193-237).  It exists for the layers whose
// K loop is SHORT against everything around it: the 1x1 convolutions (K = 128 ... 1024 = 4 ... 32 k32 steps) and the
// small-grid layers.  Measured on the register-staged kernel (profiles/r01_layers.json): the 37 1x1 layers of YOLOv3 took
// ~23 us each at 14-17 % of the MFMA roof and 10-39 % of HBM — 13 us of that with every load and the epilogue removed
// (launch, prologue latency, two stages of prefetch = at most 45 KB in flight per CU, one full drain per tile).
//
// What is different here:
//   * operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB = 16 rows x 64 B per wave-instruction):
//     no staging registers, no ds_write pass, so the ring can be STAGES deep (3-5 k32 stages, 70-130 KB in flight per CU)
//     at no register cost.  The LDS image is lane-linear per instruction; the 16-byte-chunk XOR swizzle that keeps the
//     fragment reads conflict-free is applied on the per-lane SOURCE address (cdna guide 5.4 rule 21);
//   * workgroups are persistent: a workgroup walks its tiles (XCD-aware tile order) and the ring runs ACROSS tile
//     boundaries — the first stages of tile t+1 are in flight while tile t's epilogue transposes and stores through the one
//     ring slot the loader does not own at that moment.  No load result lives in registers across the epilogue (the failure
//     mode of the register-staged attempt);
//   * one s_barrier per k32 step: wait for this wave's own DMA pieces of the step (counted vmcnt), barrier (everybody's
//     pieces landed, everybody finished reading the slot that is recycled next), issue the DMA STAGES-1 steps ahead, then
//     fragments + MFMAs.
// K order and MFMA order are those of conv_igemm_f16s3.hip / conv_band_f16s3.hip (32-channel chunk outer, tap inner; per
// chunk al*bh, ah*bl, ah*bh), so a layer gives the same bits on any tile of this kernel.
// Measured and not kept (round 2): two k32 chunks per ring stage (one barrier per 64 channels; 64x128 / 64x64 / 128x64 tiles,
// bit-identical) — 0.94-0.96 ms for the 36 1x1 layers of YOLOv3 against 0.83 for the same tiles with one chunk per stage:
// the doubled stage costs the second workgroup per CU, which hides more latency than the halved barrier count saves.
// Three workgroups per CU (64x64 with a 3-stage ring) or shallower rings on the other tiles: no change (0.84 vs 0.84).  With every
// load and the epilogue switched off (RTOD_DIAG, tools/ablate_1x1.sh) these layers still take 14-18 of their 19-23 us: what a
// step costs is issuing its LDS-DMA pieces (4-6 per wave and step at ~100-185 cycles each, MI355X_MICROARCH 'LDS-DMA piece issue
// cost') against 288-384 cycles of MFMAs — tiles with more MFMA work per staged kilobyte are the lever, not the schedule.
// In-workgroup split-K (two 8-wave groups on the even / odd chunks of a 64x128 tile, 184 workgroups at 19x19): correct
// (1.5e-5 against the oracle) and exactly as fast as the autotuned tiles (19.9 vs 19.8-20.1 us) — halving the K loop does not
// shorten these kernels: rocprofv3's per-dispatch trace shows ~4.5 us of every launch is dispatch + drain (a one-workgroup
// kernel that writes 28 bytes takes 4.8 us there) and the weights of every layer arrive cold from HBM.
#include "conv_f16s3_common.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace rtod {

// this wave's vector-memory operations except the N youngest have completed (loads, stores and LDS-DMA count together,
// in issue order)
template <int N> __device__ __forceinline__ void ring_wait_vmcnt() {
    static_assert(N >= 0 && N <= 24, "vmcnt literal");
#define RTOD_VMCNT_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RTOD_VMCNT_CASE(1) RTOD_VMCNT_CASE(2) RTOD_VMCNT_CASE(3) RTOD_VMCNT_CASE(4) RTOD_VMCNT_CASE(5) RTOD_VMCNT_CASE(6)
    RTOD_VMCNT_CASE(7) RTOD_VMCNT_CASE(8) RTOD_VMCNT_CASE(9) RTOD_VMCNT_CASE(10) RTOD_VMCNT_CASE(11) RTOD_VMCNT_CASE(12)
    RTOD_VMCNT_CASE(13) RTOD_VMCNT_CASE(14) RTOD_VMCNT_CASE(15) RTOD_VMCNT_CASE(16) RTOD_VMCNT_CASE(17) RTOD_VMCNT_CASE(18)
    RTOD_VMCNT_CASE(19) RTOD_VMCNT_CASE(20) RTOD_VMCNT_CASE(21) RTOD_VMCNT_CASE(22) RTOD_VMCNT_CASE(23) RTOD_VMCNT_CASE(24)
#undef RTOD_VMCNT_CASE
}

// Two LDS-DMA pieces (the hi and the lo plane of the same 16 rows): lane l's 16 bytes land at lds + 16*l.  M0 carries the
// LDS byte address and is written in the statement that uses it (the compiler does not preserve M0 around asm; cdna guide
// 5.7); s_nop 0: SALU write of M0 -> LDS-DMA read of M0.  voffset >= the descriptor's extent writes zeros.
__device__ __forceinline__ void dma_pair(const __amdgpu_buffer_rsrc_t rsrc_hi, const __amdgpu_buffer_rsrc_t rsrc_lo, unsigned voffset,
                                         unsigned soff_hi, unsigned soff_lo, unsigned lds_hi, unsigned lds_lo) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, %5 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(rsrc_hi), "s"(rsrc_lo), "s"(soff_hi), "s"(soff_lo), "s"(lds_hi), "s"(lds_lo)
        : "memory");
}

// BM x BN workgroup tile, NWM x NWN waves of (BM/NWM) x (BN/NWN).  STAGES ring slots of one k32 step each.
// LDS: [STAGES][A hi BM x 64 B | A lo | B hi BN x 64 B | B lo].  The epilogue's transpose tile is the ring slot the last
// K step has just been computed from: the loader runs STAGES-1 steps ahead, so exactly that slot is free until the
// next tile's first barrier.
// Diagnostic build only (make timeline): start / end wall clock of every workgroup of a launch, as in conv_band_f16s3.hip.
#ifdef RTOD_TIMELINE
constexpr int RING_TL_BLOCKS = 1024;
__device__ unsigned long long g_ring_tl[RING_TL_BLOCKS * 2];
#endif

template <int BM, int BN, int NWM, int NWN, int STAGES, int MINW, int EPI>
__global__ __launch_bounds__(NWM * NWN * 64, MINW)
void conv_ring_f16s3_kernel(const ConvArgs a, const int grid_m, const int grid_n) {
    constexpr int WM = BM / NWM, WN = BN / NWN, NW = NWM * NWN, NT = NW * 64;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % NWM == 0 && BN % NWN == 0, "wave tile");
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int PANEL_A = BM * 64, PANEL_B = BN * 64, STAGE = 2 * PANEL_A + 2 * PANEL_B;
    // DMA work of one stage: row blocks of 16 rows (one 1 KiB piece per plane).  Every wave takes A_PER blocks of A and B_PER
    // of B (block = wave + j*NW); where the blocks do not divide evenly a wave re-issues another valid block (block index
    // modulo the count: the same bytes land on the same LDS addresses twice), so that every wave issues exactly LPW DMA
    // instructions per step and one vmcnt literal serves all.
    constexpr int RB_A = BM / 16, RB_B = BN / 16;
    constexpr int A_PER = (RB_A + NW - 1) / NW, B_PER = (RB_B + NW - 1) / NW, LPW = 2 * (A_PER + B_PER);
    static_assert((STAGES - 2) * LPW <= 24 && STAGES >= 3, "vmcnt literals / ring depth");
    static_assert(EPI != EPI_SPLIT_PW && EPI != EPI_SPLIT_RES_PW, "no fused pointwise epilogue in this kernel");
    static_assert(STAGE >= WM * BN * 4, "a ring slot must hold one epilogue pass (WM rows x BN floats)");
    // split-format outputs: transposed product + register epilogue (conv_f16s3_common.h); the head decode keeps the
    // pixel-major accumulator and the LDS-transposed epilogue (its rows are 255 contiguous floats)
    constexpr bool TRANSPOSED = EPI != EPI_DECODE;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef RTOD_TIMELINE
    const unsigned long long tl_start_ = __builtin_amdgcn_s_memrealtime();
#endif

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int M = a.B * a.Ho * a.Wo;
    const int n_tiles = grid_m * grid_n;
    const unsigned PS = (unsigned)a.in_ldc * 4u;                 // bytes per pixel (hi plane + lo plane)
    const unsigned lo_plane = (unsigned)a.in_ldc * 2u;
    const int nk = a.Kpad / HBK;
    const int hw = a.Ho * a.Wo;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)smem;                // LDS byte address of the ring

    // tile sequence of this workgroup: t = blockIdx.x, + gridDim.x, ...; XCD-aware bijective remap of the tile index
    // (workgroups b and b + 8 share an XCD / L2: XCD x takes a contiguous range of tiles, neighbours share A row panels)
    auto tile_of = [&](int t, int& bm, int& bn) __attribute__((always_inline)) {
        const int q = n_tiles >> 3, r = n_tiles & 7, xcd = t & 7;
        const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
        bm = u / grid_n; bn = u - bm * grid_n;
    };

    // ---- loader state: per row block of this wave (block id = wave + j*NW; < RB_A: an A block, else a B block).
    // Lane l of a piece -> row l>>2 of the block, 16-byte position l&3, which holds source chunk (l&3) ^ swizzle(row).
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((lrow >> 1) & 3);
    int ld_tile = blockIdx.x;                                    // tile the loader is in (may run ahead of the consumer's)
    int ld_kc = 0, ld_c0 = 0, ld_ky = 0, ld_kx = 0;             // K-chunk cursor inside that tile (wave-uniform)
    unsigned pa[A_PER]; int iy0[A_PER], ix0[A_PER];             // A blocks: byte offset of the receptive-field corner, its coordinates
    unsigned pw_[B_PER];                                         // B blocks: byte offset of the weight row
    int blk_a[A_PER], blk_b[B_PER];                              // wave-uniform block ids
#pragma unroll
    for (int j = 0; j < A_PER; ++j) blk_a[j] = (wave + j * NW) % RB_A;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) blk_b[j] = (wave + j * NW) % RB_B;
    auto loader_enter_tile = [&]() __attribute__((always_inline)) {
        int bm, bn;
        const bool live = ld_tile < n_tiles;
        tile_of(live ? ld_tile : 0, bm, bn);
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int m = bm * BM + blk_a[j] * 16 + lrow;
            const bool ok = live && m < M;
            const int mm = ok ? m : 0;
            const int b = mm / hw, r = mm - b * hw;
            const int oy = r / a.Wo, ox = r - oy * a.Wo;
            iy0[j] = ok ? oy * a.stride - a.pad : -(1 << 28);
            ix0[j] = ox * a.stride - a.pad;
            pa[j] = (unsigned)((b * a.Hi + iy0[j]) * a.Wi + ix0[j]) * PS + (unsigned)(a.in_coff + lchunk * 8) * 2u;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int rho = blk_b[j] * 16 + lrow;                               // panel row; transposed product: channel order of tr_chan_of_row
            const int n = bn * BN + (TRANSPOSED ? tr_chan_of_row(rho) : rho);
            pw_[j] = live ? (unsigned)(n * 32 + lchunk * 8) * 2u : OOB;          // K-chunk major planes: [chunk][Npad][32]
        }
        ld_kc = 0; ld_c0 = 0; ld_ky = 0; ld_kx = 0;
    };
    // issue this wave's pieces of the next stage into ring slot `slot`, advance the cursor (to the next tile at the end of K)
    auto loader_issue = [&](int slot) __attribute__((always_inline)) {
        const unsigned tap_off = (unsigned)(ld_ky * a.Wi + ld_kx) * PS + (unsigned)ld_c0 * 2u;
        const unsigned koff = (unsigned)ld_kc * (unsigned)a.Npad * (HBK * 2);
        const unsigned sbase = lds0 + (unsigned)slot * STAGE;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const bool ok = (unsigned)(iy0[j] + ld_ky) < (unsigned)a.Hi && (unsigned)(ix0[j] + ld_kx) < (unsigned)a.Wi;
            const unsigned vo = ok ? pa[j] + tap_off : OOB;
            const unsigned l = sbase + (unsigned)blk_a[j] * 1024u;
            dma_pair(rs_a, rs_a, vo, 0u, lo_plane, l, l + PANEL_A);
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const unsigned l = sbase + 2u * PANEL_A + (unsigned)blk_b[j] * 1024u;
            dma_pair(rs_wh, rs_wl, pw_[j], koff, koff, l, l + PANEL_B);           // hi and lo weight planes: two allocations
        }
        ++ld_kc;
        if (++ld_kx == a.kw) { ld_kx = 0; if (++ld_ky == a.kh) { ld_ky = 0; ld_c0 += HBK; } }
        if (ld_kc == nk) { ld_tile += gridDim.x; loader_enter_tile(); }
    };

    // ---- consumer state
    const int wm = wave / NWN, wn = wave - wm * NWN;
    const int lr = lane & 15, lh = lane >> 4;
    const int co = (lh ^ ((lr >> 1) & 3)) << 4;                  // WM, WN % 16 == 0: the row's swizzle is the lane's
    const int a_row = (wm * WM + lr) * 64 + co, b_row = 2 * PANEL_A + (wn * WN + lr) * 64 + co;

    loader_enter_tile();
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) loader_issue(s);        // prologue: STAGES-1 stages in flight
    int slot = 0;                                                // ring slot of the step about to be computed

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int bm, bn;
        tile_of(tile, bm, bn);
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        // bias / inv_scale of this wave's column groups, loaded and WAITED FOR here: the compiler's wait for a global load
        // is a vmcnt that also covers every DMA piece issued before it; at the start of a tile that only waits for stages
        // this tile needs next, inside the epilogue it would drain the ring that is prefetching the next tile
        float ebias[TN], einv[TN];
        if constexpr (!TRANSPOSED) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = bn * BN + wn * WN + j * 16 + lr;
                ebias[j] = n < a.Cout ? a.bias[n] : 0.f;
                einv[j] = n < a.Cout ? a.inv_scale[n] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(ebias[j]), "+v"(einv[j]));
        }

#pragma unroll 1
        for (int kc = 0; kc < nk; ++kc) {
            ring_wait_vmcnt<(STAGES - 2) * LPW>();               // this wave's pieces of the current step have landed (the younger STAGES-2 stages may stay in flight)
            __builtin_amdgcn_s_barrier();                        // ... and everybody's; the previous step's slot is free
            {
                const int prev = slot == 0 ? STAGES - 1 : slot - 1;
                loader_issue(prev);                              // the stage STAGES-1 steps ahead (possibly of the next tile)
            }
            const unsigned char* st = smem + slot * STAGE;
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(st + a_row + i * 16 * 64);
                al[i] = *reinterpret_cast<const f16x8*>(st + PANEL_A + a_row + i * 16 * 64);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(st + b_row + j * 16 * 64);
                bl[j] = *reinterpret_cast<const f16x8*>(st + PANEL_B + b_row + j * 16 * 64);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (TRANSPOSED) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
            slot = slot + 1 == STAGES ? 0 : slot + 1;
        }
        // epilogue: the slot just consumed is the only one the loader does not own (it is refilled after the next tile's
        // first barrier); the other STAGES-1 slots keep filling for the next tile meanwhile
#ifdef RTOD_DIAG
        if (a.dbg & 4) continue;                                 // timing experiment: no epilogue
#endif
        if constexpr (TRANSPOSED) {
            // straight from the accumulators: no LDS, no barrier — the waves drift apart here and re-align at the next step
            int mrow[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) { const int m = bm * BM + wm * WM + i * 16 + lr; mrow[i] = m < M ? m : -1; }
            conv_f16s3_epilogue_regs<WM, WN, EPI == EPI_SPLIT_RES>(a, acc, nullptr, mrow, bn * BN + wn * WN, tid, lh);
        } else {
            const int done = slot == 0 ? STAGES - 1 : slot - 1;
            __builtin_amdgcn_s_barrier();                        // every wave has read its last fragments from that slot
            conv_f16s3_epilogue<BM, BN, WM, WN, NT, EPI, STAGE, 1, true>(a, acc, smem + done * STAGE, bm, bn, tid, wm, wn, lr, lh, M, 0, ebias, einv);
            __builtin_amdgcn_s_barrier();                        // transpose reads done before the slot is handed back to the loader
        }
    }
    ring_wait_vmcnt<0>();                                        // trailing (out-of-range) pieces: nothing may be in flight at exit
#ifdef RTOD_TIMELINE
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < RING_TL_BLOCKS) { g_ring_tl[blockIdx.x * 2] = tl_start_; g_ring_tl[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// One list drives the mode table, the launch switch and the kernel names rocprofv3 prints:
//   X(mode, BM, BN, waves along M, waves along N, STAGES, MINW, workgroups per CU)
#define RTOD_RING_TILES(X) \
    X(0, 64, 128, 2, 2, 3, 2, 2) X(1, 128, 128, 4, 2, 4, 2, 1) X(2, 64, 64, 2, 2, 4, 2, 2) X(3, 128, 64, 4, 2, 3, 4, 2) \
    X(4, 32, 128, 1, 4, 4, 2, 2) X(5, 192, 128, 4, 2, 3, 2, 1) X(6, 96, 128, 2, 4, 5, 2, 1) X(7, 128, 128, 4, 2, 5, 2, 1)

#define RTOD_X_INFO(mode, bm, bn, nwm, nwn, st, minw, wpc) {bm, bn, "conv_ring_f16s3<" #bm "x" #bn "," #nwm "x" #nwn ",s" #st ">"},
static const ConvVariantInfo kRingModes[RING_MODES] = { RTOD_RING_TILES(RTOD_X_INFO) };
#undef RTOD_X_INFO
const ConvVariantInfo& conv_ring_mode_info(int mode) { return kRingModes[mode < 0 || mode >= RING_MODES ? 0 : mode]; }

int conv_ring_kernel_name(int mode, int epi, char* buf, size_t len) {
#define RTOD_X_NAME(m, bm, bn, nwm, nwn, st, minw, wpc) \
    if (mode == m) return snprintf(buf, len, "void rtod::conv_ring_f16s3_kernel<" #bm ", " #bn ", " #nwm ", " #nwn ", " #st ", " #minw ", %d>(rtod::ConvArgs, int, int)", epi);
    RTOD_RING_TILES(RTOD_X_NAME)
#undef RTOD_X_NAME
    return -1;
}

template <int BM, int BN, int NWM, int NWN, int STAGES, int MINW, int WPC>
static int launch_ring(const ConvArgs& a, hipStream_t s) {
    constexpr int NT = NWM * NWN * 64;
    const int M = a.B * a.Ho * a.Wo;
    const int gm = (M + BM - 1) / BM, gn = (a.Cout + BN - 1) / BN;
    const bool pw = a.pw_wh != nullptr;
    const int lds = STAGES * (2 * BM * 64 + 2 * BN * 64);
    if (lds > 160 * 1024) { set_error("conv_ring_f16s3: %d bytes of LDS", lds); return RTOD_E_ARG; }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return hip_fail(hipGetLastError(), "conv_ring_f16s3 device query");
    const int slots = cus * ((160 * 1024) / lds < WPC ? (160 * 1024) / lds : WPC);
    const int tiles = gm * gn;
    const int grid = tiles < slots ? tiles : slots;
    auto k_dec = conv_ring_f16s3_kernel<BM, BN, NWM, NWN, STAGES, MINW, EPI_DECODE>;
    auto k_res = conv_ring_f16s3_kernel<BM, BN, NWM, NWN, STAGES, MINW, EPI_SPLIT_RES>;
    auto k_plain = conv_ring_f16s3_kernel<BM, BN, NWM, NWN, STAGES, MINW, EPI_SPLIT>;
    static std::atomic<unsigned long long> attr_done{0};       // per instantiation, one bit per device: > 64 KiB of dynamic LDS needs the opt-in
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        const int mx = 160 * 1024;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dec), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_res), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_plain), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
            return hip_fail(hipGetLastError(), "conv_ring_f16s3 LDS attribute");
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    if (pw) { set_error("conv_ring_f16s3: fused pointwise epilogue not built for this kernel"); return RTOD_E_ARG; }
    if (a.dec.enabled) hipLaunchKernelGGL(k_dec, dim3(grid), dim3(NT), lds, s, a, gm, gn);
    else if (a.res) hipLaunchKernelGGL(k_res, dim3(grid), dim3(NT), lds, s, a, gm, gn);
    else hipLaunchKernelGGL(k_plain, dim3(grid), dim3(NT), lds, s, a, gm, gn);
#ifdef RTOD_TIMELINE
    {
        static int printed = 0;
        const int nb = grid < RING_TL_BLOCKS ? grid : RING_TL_BLOCKS;
        if (printed < 80 && hipDeviceSynchronize() == hipSuccess) {
            static unsigned long long h[RING_TL_BLOCKS * 2];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ring_tl), sizeof(unsigned long long) * 2 * nb) == hipSuccess) {
                unsigned long long t0 = ~0ull, t1 = 0; double d = 0, smax = 0, emin = 1e9;
                for (int b = 0; b < nb; ++b) { if (h[b * 2] < t0) t0 = h[b * 2]; if (h[b * 2 + 1] > t1) t1 = h[b * 2 + 1]; }
                for (int b = 0; b < nb; ++b) { d += (h[b * 2 + 1] - h[b * 2]) / 100.0; const double st = (h[b * 2] - t0) / 100.0, en = (h[b * 2 + 1] - t0) / 100.0; if (st > smax) smax = st; if (en < emin) emin = en; }
                fprintf(stderr, "[timeline] ring<%d,%d,%dx%d,s%d> k=%d Ho=%d Cin=%d Cout=%d tiles=%d wgs=%d | span %.1f us | wg mean %.1f us | last start %.1f | first end %.1f\n",
                        BM, BN, NWM, NWN, STAGES, a.kh, a.Ho, a.Cin, a.Cout, tiles, grid, (t1 - t0) / 100.0, d / nb, smax, emin);
                ++printed;
            }
        }
    }
#endif
    return hip_fail(hipGetLastError(), "conv_ring_f16s3 launch");
}

int launch_conv_ring_f16s3(const ConvArgs& a_in, int mode, hipStream_t s) {
    ConvArgs a = a_in;
    if (!a.in || !a.w_hi || !a.w_lo || !a.bias || !a.inv_scale || !a.out) { set_error("launch_conv_ring: null pointer"); return RTOD_E_ARG; }
    if (a.Cin % HBK || a.in_ldc % 8 || a.in_coff % 8 || a.Kpad % HBK || a.K != a.Kpad || a.K != a.kh * a.kw * a.Cin) {
        set_error("launch_conv_ring: needs Cin %% 32 == 0 and 8-channel aligned views (Cin=%d ldc=%ld coff=%d K=%d Kpad=%d)", a.Cin, (long)a.in_ldc, a.in_coff, a.K, a.Kpad);
        return RTOD_E_ARG;
    }
    if (a.B <= 0 || a.Ho <= 0 || a.Wo <= 0 || a.Cout <= 0) { set_error("launch_conv_ring: empty shape"); return RTOD_E_ARG; }
    if (a.in_bytes == 0 || a.in_bytes >= OOB || a.w_bytes == 0 || a.w_bytes >= OOB) { set_error("launch_conv_ring: buffer extents"); return RTOD_E_ARG; }
    if ((uint64_t)a.B * a.Hi * a.Wi * a.in_ldc * 4ull > (uint64_t)a.in_bytes) { set_error("launch_conv_ring: input view exceeds its buffer"); return RTOD_E_ARG; }
#ifdef RTOD_DIAG
    // diagnostic build only: zero-extent descriptors drop the loads of one operand (DMA pieces then write zeros), bit 4 the epilogue
    static const int dbg_zero = getenv("RTOD_DBG_ZERO") ? atoi(getenv("RTOD_DBG_ZERO")) : 0;
    if (dbg_zero & 1) a.in_bytes = 1;
    if (dbg_zero & 2) a.w_bytes = 1;
    a.dbg = dbg_zero;
#endif
    switch (mode) {
#define RTOD_X_CASE(m, bm, bn, nwm, nwn, st, minw, wpc) case m: return launch_ring<bm, bn, nwm, nwn, st, minw, wpc>(a, s);
        RTOD_RING_TILES(RTOD_X_CASE)
#undef RTOD_X_CASE
    }
    set_error("launch_conv_ring: mode %d unsupported", mode);
    return RTOD_E_ARG;
}

}  // namespace rtod
