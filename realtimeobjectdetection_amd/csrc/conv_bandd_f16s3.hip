// 3x3 / stride 1 / pad 1 convolution, LDS-resident input band + weight fragments straight from global memory
// (split-precision f16 MFMA), gfx950.  Round 4.
//
// Same contract, band geometry, K order and MFMA order as conv_band_f16s3.hip (reference: conv -> BN -> leaky,
// src/darknet.py:467-501, shortcut fused, 263-268): every mode of this file is one more tile of the band family and gives the
// same bits as the others (chunk outer, tap inner; per step al*bh, ah*bl, ah*bh), so the frame-independence rule of
// conv_band_f16s3.hip holds.
//
// What is different.  conv_band_f16s3.hip stages the weight panel of every (chunk, tap) step through LDS: global -> VGPR ->
// ds_write -> barrier -> ds_read, one workgroup barrier per 24 MFMAs of a wave, and a step's memory phase and MFMA phase do not
// overlap inside a workgroup (round 3's ablations: the bare MFMA loop is 53 % of the kernel).  But the packed weight planes are
// ALREADY in MFMA fragment order: [step][Npad][32] f16, so the B operand of a 16-column tile (lane -> row lr, 16-byte chunk
// lh) is 1 KiB of consecutive bytes.  Here every wave loads its own B fragments with `buffer_load_dwordx4` straight into
// registers, two steps ahead (three register sets), and owns ALL BM rows of a WN-column strip of the tile, so no two waves of
// a workgroup load the same weights.  What is left in LDS is the input band, written by LDS-DMA (no staging registers, no
// ds_write).  Consequences:
//   * no barrier, no LDS write and no B fragment read inside a channel chunk: a wave runs 9 taps x TM x TN x 3 MFMAs (432 for a
//     128x32 strip) between two barriers, its A fragments read two 16-row tiles ahead of their products;
//   * LDS traffic per MFMA drops from 512 B (2 + 4 tiles of 32x64 per 24 MFMAs, plus the staging writes) to 341 B, L1 traffic
//     stays what it was (every weight byte is loaded once per workgroup, as before);
//   * LDS per workgroup is the band alone (37 KB at 76x76): 128x128 tiles of FOUR waves fit three to a CU, which puts the 722
//     tiles of a 76x76x8 layer on 768 slots in ONE round (conv_band: 512 slots, 1.41 rounds, the second one half empty).
//
// Band image in LDS: 16-row blocks of 2 KiB, [16 rows x 64 B hi][16 rows x 64 B lo]; inside a 1 KiB piece the 16-byte chunk
// position of (row, chunk c) is c ^ ((row >> 1) & 3) as in conv_band_f16s3.hip (conflict-free ds_read_b128 at every shift;
// tools/lds_bank_sim.py: a block boundary is a multiple of 256 B, the bank pattern is that of the contiguous image).  One
// `buffer_load_dwordx4 ... lds` writes a whole piece (lane l -> 16 bytes at 16 l: row l >> 2, position l & 3; the swizzle is
// applied on the lane's SOURCE address).  The lo piece sits 1024 B behind the hi piece and the next 16-row tile 2048 B further:
// both are instruction offsets, so a unit (one 16-row tile of one tap) costs two vector instructions of address work.
// Out-of-image taps read a zero block, selected per lane from a 9-bit mask per tile, as in conv_band_f16s3.hip.
//
// vmcnt bookkeeping (in-order retirement; LDS-DMA counts too).  At the top of every chunk everything this wave has issued is
// waited for (vmcnt(0): its band pieces, the B sets of the chunk's first two steps), then the barrier, then the next chunk's
// band pieces are issued (double-buffered band) and step t issues the B set of step t + 2 and, from t = 2 on, waits with
// vmcnt(2 NB) for its own set (two younger sets stay in flight; that wait also covers the band pieces, issued two steps earlier).
// Single-buffered band (76x76: three workgroups per CU leave 53 KB each): barrier, DMA, vmcnt(0), barrier at every chunk top;
// the other two workgroups' waves on the SIMD cover the exposed load.
#include "conv_bandd_common.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

namespace rtod {

constexpr int BANDD_MAX_W = 94;           // tiles of the band layers (conv_band_supported)
constexpr int BANDD_WIDE_W = 160;         // the wide tile: 3x3 layers with 94 < W <= 160 (152x152 at 608, 104x104 at 416), one band buffer of up to 29 blocks

// Diagnostic build only (make timeline -> librtod_tl.so, -DRTOD_TIMELINE): per workgroup (wave 0) wall clock at start / end and shader
// cycles spent in: prologue, chunk tops (arrival -> barrier passed), chunk bodies, drain, epilogue.  The product library compiles none of it.
#ifdef RTOD_TIMELINE
constexpr int BD_TL_BLOCKS = 2048, BD_TL_N = 8;
__device__ unsigned long long g_bandd_tl[BD_TL_BLOCKS * BD_TL_N];
#define BD_STAMP(slot) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tl_[slot] += tn_ - tl_prev_; tl_prev_ = tn_; }
#else
#define BD_STAMP(slot)
#endif
__host__ __device__ constexpr int bandd_rows(int bm, int w) { return (bm + 2 * w + 2 + 15) / 16 * 16; }

// BM x BN workgroup tile; NWM x NWN waves per K group, each wave a (BM/NWM) x (BN/NWN) strip (NWM = 1: no weight byte is
// loaded twice in a workgroup).  DB: double-buffered band.  KG = 2: two wave groups on the even / odd channel chunks (own band
// buffers), summed in the epilogue — the split-K layers of conv_band_f16s3.hip (conv_band_layer_kg), same summation order.
template <int BM, int BN, int NWM, int NWN, int MINW, int EPI, int BUFM, int KG, int MAXW>
__global__ __launch_bounds__(NWM * NWN * 64 * KG, MINW)
void conv_bandd_f16s3_kernel(const ConvArgs a, const int grid_m, const int grid_n) {
    // BUFM: 0 one band buffer, replaced between two barriers at every chunk top (the load is exposed; other workgroups cover it);
    //       1 two band buffers, the next chunk's band loads under the current chunk.
    // (Round 4 also measured a ROLLING band — one buffer replaced in place in three parts as the taps release its rows, every load
    //  under two steps of MFMAs, three barriers per chunk: bit-identical, 2.6 % SLOWER than BUFM 0 at 76x76; a barrier costs the
    //  workgroup 1.2-1.8 k cycles of skew, more than the exposed load.  profiles/experiments/r04_bandd_roll_stagger_prio.patch)
    constexpr bool DB = BUFM == 1;
    static_assert(BUFM == 0 || BUFM == 1, "band buffering");
    constexpr int WM = BM / NWM, WN = BN / NWN, NW = NWM * NWN, NT = NW * 64;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % NWM == 0 && BN % NWN == 0, "wave tile");
    constexpr int TM = WM / 16, TN = WN / 16;
    static_assert(TN <= 2, "strip width: 16 or 32 columns (instruction offsets, tie)");
    constexpr int NB = 2 * TN;                                  // B loads of one step
    constexpr int UNITS = 9 * TM;                               // (tap, 16-row tile) units of one channel chunk
    constexpr int NBLK_MAX = bandd_rows(BM, MAXW) / 16;
    constexpr int PPW = (NBLK_MAX + NW - 1) / NW;               // band blocks per wave, at most
    constexpr int RG = BM * BN * 4 <= 32768 ? BM : (BM / 2) * BN * 4 <= 32768 ? BM / 2 : BM / 4;   // epilogue rows per pass (<= 32 KB of fp32)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef RTOD_TIMELINE
    const unsigned long long tl_start_ = __builtin_amdgcn_s_memrealtime();
    unsigned long long tl_prev_ = __builtin_amdgcn_s_memtime();
    unsigned long long tl_[5] = {0, 0, 0, 0, 0};               // prologue, chunk tops, chunk bodies, drain, epilogue
#endif
    const int W = a.Wi, H = a.Hi;
    const int NBLK = bandd_rows(BM, W) / 16;
    const int BUF = (NBLK + 1) * 2048;                          // one band buffer: NBLK blocks + the zero block
    const int zero_off = NBLK * 2048;
    const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);
    const int gbase = kg * (DB ? 2 : 1) * BUF;                  // this K group's band buffers

    const int nwg = grid_m * grid_n;
    int bid = blockIdx.x;
    if (!a.xcd_by_n) {                                          // XCD x (= blockIdx % 8) takes a contiguous range of tiles
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int bm = bid / grid_n, bn = bid - bm * grid_n;

    const int tid = KG == 1 ? (int)threadIdx.x : (int)threadIdx.x - kg * NT;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = NWM == 1 ? 0 : wave / NWN, wn = NWM == 1 ? wave : wave - wm * NWN;     // NWM == 1: wm is a compile-time 0 (epilogue passes)
    const int lr = lane & 15, lh = lane >> 4;
    const int M = a.B * H * W;
    const int m0 = bm * BM;
    const int NBv = BM + 2 * W + 2;                             // band rows that can hold a pixel
    const unsigned PS = (unsigned)a.in_ldc * 4u;
    const unsigned lo_plane = (unsigned)a.in_ldc * 2u;

    // (Round 4 also measured per-GROUP chunk-top barriers for the KG == 2 tiles — an arrival counter in LDS per K group instead of the
    //  workgroup's s_barrier, so that one group's wait would be the other's MFMA time: bit-identical, 1.5 % SLOWER on the 19x19 layers in
    //  an in-process A/B (profiles/experiments/r04_epilogue_groupbarrier_ab.log).  The 39 k cycles a wave spends at the chunk tops of a
    //  19x19 tile are mostly its SIMD partner's MFMA time: inside the loop the pipe is already ~80 % busy.)
    // zero blocks (hi row 0 and lo row 0 of the block behind the band), every buffer of every group
    if (threadIdx.x < 8 * (DB ? 2 : 1) * KG) {
        const int b = threadIdx.x >> 3, k = threadIdx.x & 7;
        *reinterpret_cast<u32x4*>(smem + b * BUF + zero_off + (k >> 2) * 1024 + (k & 3) * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- band DMA: block = wave + k NW; lane -> row lane >> 2 of the block, position lane & 3 <- source chunk (lane & 3) ^ swizzle(row)
    // (per-block source offsets are recomputed at every chunk top from two registers: held in registers across the main loop they
    //  were the first thing the allocator spilled)
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)smem;
    // Every wave issues the SAME number of pieces, n_dma = ceil(NBLK / NW) pairs (the counted waits of the steps behind a DMA
    // need it): a wave whose block index runs past the band writes zeros (out-of-range source) into the zero block instead.
    const int n_dma = (NBLK + NW - 1) / NW;
    static_assert(PPW <= 5 || !DB, "bandd_wait_vmcnt_plus covers 5 pairs");
    // blocks [blk_lo, blk_hi) of channel chunk cc -> band buffer buf, n pairs per wave (n = ceil((blk_hi - blk_lo) / NW), wave-uniform)
    auto dma_blocks = [&](int cc, int buf, int blk_lo, int blk_hi, int n) __attribute__((always_inline)) {
        const unsigned soff = (unsigned)(cc * KG + kg) * 64u;
        const unsigned base = lds0 + (unsigned)(gbase + buf * BUF);
        int l = (int)threadIdx.x;
        asm volatile("" : "+v"(l));                             // not loop-invariant: see above (everything below is recomputed per call)
        const int dma_lrow = (l & 63) >> 2;
        const unsigned dma_cpart = (unsigned)(a.in_coff + ((l & 3) ^ ((dma_lrow >> 1) & 3)) * 8) * 2u;
        const int r0 = (blk_lo + wave) * 16 + dma_lrow;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int blk = blk_lo + wave + k * NW;
            const int r = r0 + k * NW * 16;
            const int q = m0 - W - 1 + r;
            const unsigned vo = (blk < blk_hi && r < NBv && (unsigned)q < (unsigned)M) ? (unsigned)q * PS + dma_cpart : OOB;
            if (k < n) bandd_dma_pair(rs_a, vo, soff, lo_plane + soff, base + (unsigned)(blk < blk_hi ? blk * 2048 : zero_off));
        }
    };
    auto dma_band = [&](int cc, int buf) __attribute__((always_inline)) { dma_blocks(cc, buf, 0, NBLK, n_dma); };
    // ---- B fragments: lane (lr, lh) <- weight row n0 + 16 j + lr, 16-byte chunk lh of the step's panel; 3 register sets
    const int n0 = bn * BN + wn * WN;
    const unsigned bvoff = n0 + lr < a.Npad ? (unsigned)((n0 + lr) * 32 + lh * 8) * 2u : OOB;     // Npad % 128 == 0: a strip is inside or outside
    const unsigned wchunk = (unsigned)a.Npad * (HBK * 2);       // bytes of one (chunk, tap) panel of a weight plane
    const int n_cc = a.Cin / 32 / KG;                           // channel chunks of this group: kg, kg + KG, ...
    const int nsteps = 9 * n_cc;
    u32x4 Bq[3][TN][2];
    int ld_step = 0, ld_cc = 0, ld_tap = 0;                     // step whose B set is loaded next: local index, chunk, tap
    auto load_b = [&](u32x4 (&q)[TN][2]) __attribute__((always_inline)) {
        const unsigned vo = ld_step < nsteps ? bvoff : OOB;
        const unsigned koff = (unsigned)((ld_cc * KG + kg) * 9 + ld_tap) * wchunk;
        q[0][0] = bandd_load_b<0>(rs_wh, vo, koff);
        q[0][1] = bandd_load_b<0>(rs_wl, vo, koff);
        if constexpr (TN == 2) {
            q[1][0] = bandd_load_b<1024>(rs_wh, vo, koff);
            q[1][1] = bandd_load_b<1024>(rs_wl, vo, koff);
        }
        ++ld_step;
        if (++ld_tap == 9) { ld_tap = 0; ++ld_cc; }
    };

    // ---- per-lane validity of the 9 taps for the TM 16-row tiles (9 bits per tile, three tiles per register)
    constexpr int NVM = (TM + 2) / 3;
    unsigned vmw[NVM];
    {
#pragma unroll
        for (int w = 0; w < NVM; ++w) vmw[w] = 0u;
        const int hw = H * W;
        int m = m0 + wm * WM + lr;
        const int mm = m < M ? m : 0;
        int b = mm / hw;
        int r = mm - b * hw;
        int oy = r / W, ox = r - oy * W;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            unsigned vm = 0;
            if (m < M) {
                const unsigned xm = (ox >= 1 ? 1u : 0u) | 2u | (ox + 1 < W ? 4u : 0u);
                vm = (oy >= 1 ? xm : 0u) | (xm << 3) | (oy + 1 < H ? xm << 6 : 0u);
            }
            vmw[i / 3] |= vm << ((i % 3) * 9);
            m += 16; ox += 16;
            if (W >= 16) {                                      // (uniform) one wrap at most: two selects instead of two divergent loops per tile
                const bool wx = ox >= W; ox = wx ? ox - W : ox; oy += wx ? 1 : 0;
                const bool wy = oy >= H; oy = wy ? oy - H : oy;
            } else {
                while (ox >= W) { ox -= W; ++oy; }
                while (oy >= H) { oy -= H; ++b; }
            }
        }
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    // ---- A fragments.  Unit (t, i): rows prow0 + 16 i + shift(t) of the band.  x0(t): LDS offset of tile 0's row of this lane.
    const int prow0 = wm * WM + lr;
    const int lh4 = lh << 4;
    f16x8 Ah[3], Al[3];
    // prow_c / vmw are laundered through an empty asm at every chunk top — left loop-invariant, the 9 TM unit addresses are
    // hoisted out of the chunk loop and spilled (72 scratch reloads, each behind a vmcnt(0), in the main loop)
    typedef const f16x8 __attribute__((address_space(3))) lds_f16x8;
    int prow_c = prow0;
    auto read_unit = [&](int t, int i, int bufoff, f16x8& h, f16x8& l) __attribute__((always_inline)) {
        const int row = prow_c + (t / 3) * W + (t % 3);
        const int x0 = (((row & ~15) << 7) | ((row & 15) << 6) | (((row << 3) ^ lh4) & 0x30)) + bufoff;
        // o = tap valid ? x0 : zero block.  Written as a ternary it becomes a divergent branch per unit, as (x0 & sel) | (z & ~sel) five
        // vector instructions; bit-field extract + v_bfi_b32 is two
        const int sel = __builtin_amdgcn_sbfe((int)vmw[i / 3], (i % 3) * 9 + t, 1);     // 0 or -1
        const int z = bufoff + zero_off - 2048 * i;
        int o;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(o) : "v"(sel), "v"(x0), "s"(z));
        h = *reinterpret_cast<lds_f16x8*>((unsigned)(o + 2048 * i));
        l = *reinterpret_cast<lds_f16x8*>((unsigned)(o + 2048 * i + 1024));
    };

    // ---- prologue: band chunk 0, B sets of steps 0 and 1
    dma_band(0, 0);
    load_b(Bq[0]);
    load_b(Bq[1]);
    if constexpr (DB) {                                         // band(0) landed (the chunk tops of the double-buffered loop wait for nothing)
        bandd_wait_vmcnt<0>();
        bandd_tie<TN>(Bq[0]); bandd_tie<TN>(Bq[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                            // zero blocks written

    // (Round 4 measured a prefetch of the tile's shortcut operand into L2 here — two dword touches per thread at steps 3 and 5 of the last
    //  channel chunk, counted into the waits of steps 3-7: bit-identical, no gain (band76 +0.4 %, band38 -0.3 %, band19 +0.5 %).  The
    //  epilogue's cost is vector issue — three waves per SIMD in the same phase — not the shortcut's latency: stamps and a timing
    //  ablation in profiles/experiments/r04_bandd_epilogue_*.log, patch r04_bandd_shortcut_prefetch.patch.)
    BD_STAMP(0)
    __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
    for (int cc = 0; cc < n_cc; ++cc) {
        const int buf = DB ? (cc & 1) : 0;
        const int bufoff = (int)lds0 + gbase + buf * BUF;
        asm volatile("" : "+v"(prow_c));
#pragma unroll
        for (int w = 0; w < NVM; ++w) asm volatile("" : "+v"(vmw[w]));
        // ---- chunk top: everything issued so far has landed; band(cc) complete in every wave's view
        if constexpr (!DB) {
            if (cc > 0) {
                __builtin_amdgcn_s_barrier();                   // every wave has read its last fragments of chunk cc - 1
                dma_band(cc, 0);
            }
        }
        // DB: this wave's pieces of band(cc) were issued at the previous chunk top and are older than the B set step 2 of that
        // chunk waited for: landed.  Nothing is waited for here: the B sets of the next two steps stay in flight across the barrier.
        int dma_behind = 0;                                     // band pairs issued between B(s + 1) and B(s + 2) of the current phase's first step
        if constexpr (!DB) {
            bandd_wait_vmcnt<0>();
            bandd_tie<TN>(Bq[0]); bandd_tie<TN>(Bq[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (DB) { if (cc + 1 < n_cc) { dma_band(cc + 1, buf ^ 1); dma_behind = n_dma; } }
        BD_STAMP(1)
        read_unit(0, 0, bufoff, Ah[0], Al[0]);
        if constexpr (UNITS > 1) read_unit(1 / TM, 1 % TM, bufoff, Ah[1], Al[1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int t = u / TM, i = u % TM;
            if (i == 0) {                                       // step head: B set of step t + 2, wait for this step's
                load_b(Bq[(t + 2) % 3]);
                if constexpr (DB) {                             // younger than this step's set: the next two sets and, for t < 2, the band pieces
                    if (t >= 2) bandd_wait_vmcnt<2 * NB>(); else bandd_wait_vmcnt_plus<2 * NB>(dma_behind);
                    bandd_tie<TN>(Bq[t % 3]);
                } else if (t >= 2) { bandd_wait_vmcnt<2 * NB>(); bandd_tie<TN>(Bq[t % 3]); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (u + 2 < UNITS) read_unit((u + 2) / TM, (u + 2) % TM, bufoff, Ah[(u + 2) % 3], Al[(u + 2) % 3]);
            const f16x8 ah = Ah[u % 3], al = Al[u % 3];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f16x8 bh = __builtin_bit_cast(f16x8, Bq[t % 3][j][0]), bl = __builtin_bit_cast(f16x8, Bq[t % 3][j][1]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        BD_STAMP(2)
    }
    // drain: the two trailing (out-of-range) B sets' registers stay allocated until they have landed
    bandd_wait_vmcnt<0>();
    bandd_tie<TN>(Bq[0]); bandd_tie<TN>(Bq[1]); bandd_tie<TN>(Bq[2]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    BD_STAMP(3)

    bandd_epilogue<BM, BN, WM, WN, NT * KG, RG, EPI == EPI_SPLIT_RES, KG>(a, acc, smem, bm, bn, (int)threadIdx.x, wm, wn, lr, lh, M, kg);
#ifdef RTOD_TIMELINE
    BD_STAMP(4)
    if (threadIdx.x == 0 && blockIdx.x < BD_TL_BLOCKS) {
        unsigned long long* o = g_bandd_tl + blockIdx.x * BD_TL_N;
        o[0] = tl_start_; o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);   // HW_ID, XCC_ID
        for (int i = 0; i < 5; ++i) o[3 + i] = tl_[i];
    }
#endif
}

template <int BM, int BN, int NWM, int NWN, int MINW, int BUFM, int KG, int MAXW>
static int launch_bandd(const ConvArgs& a, hipStream_t s) {
    constexpr bool DB = BUFM == 1;
    constexpr int NT = NWM * NWN * 64 * KG;
    constexpr int RG = BM * BN * 4 <= 32768 ? BM : (BM / 2) * BN * 4 <= 32768 ? BM / 2 : BM / 4;
    const int M = a.B * a.Ho * a.Wo;
    const int gm = (M + BM - 1) / BM, gn = (a.Cout + BN - 1) / BN;
    if (a.Cin % (32 * KG)) { set_error("launch_conv_bandd: Cin=%d not a multiple of %d", a.Cin, 32 * KG); return RTOD_E_ARG; }
    ConvArgs ax = a;
    ax.xcd_by_n = (gn % 8 == 0 && (int64_t)a.Cout * a.K > (int64_t)M * a.Cin) ? 1 : 0;
    const int main_bytes = KG * (DB ? 2 : 1) * (bandd_rows(BM, a.Wi) / 16 + 1) * 2048;
    const int epi_bytes = RG * BN * 4;
    const int lds = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    if (lds > 160 * 1024) { set_error("launch_conv_bandd: %d bytes of LDS", lds); return RTOD_E_ARG; }
    auto k_res = conv_bandd_f16s3_kernel<BM, BN, NWM, NWN, MINW, EPI_SPLIT_RES, BUFM, KG, MAXW>;
    auto k_plain = conv_bandd_f16s3_kernel<BM, BN, NWM, NWN, MINW, EPI_SPLIT, BUFM, KG, MAXW>;
    static std::atomic<unsigned long long> attr_done{0};       // per instantiation, one bit per device; > 64 KiB of dynamic LDS needs the opt-in
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError(), "conv_bandd_f16s3 hipGetDevice");
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        const int cap = KG * (DB ? 2 : 1) * (bandd_rows(BM, MAXW) / 16 + 1) * 2048;
        const int mx = std::min(cap > epi_bytes ? cap : epi_bytes, 160 * 1024);    // (a launch that needs more is refused above)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_res), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_plain), hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
            return hip_fail(hipGetLastError(), "conv_bandd_f16s3 LDS attribute");
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    if (a.res) hipLaunchKernelGGL(k_res, dim3(gm * gn), dim3(NT), lds, s, ax, gm, gn);
    else hipLaunchKernelGGL(k_plain, dim3(gm * gn), dim3(NT), lds, s, ax, gm, gn);
#ifdef RTOD_TIMELINE
    {   // diagnostic build: synchronises, prints the launch's workgroup timeline
        static int printed = 0;
        const int nb = gm * gn < BD_TL_BLOCKS ? gm * gn : BD_TL_BLOCKS;
        if (printed < 200 && hipDeviceSynchronize() == hipSuccess) {
            static unsigned long long h[BD_TL_BLOCKS * BD_TL_N];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bandd_tl), sizeof(unsigned long long) * BD_TL_N * nb) == hipSuccess) {
                unsigned long long t0 = ~0ull, t1 = 0;
                double ph[5] = {0, 0, 0, 0, 0}, rt = 0, smax = 0, emin = 1e9;
                for (int b = 0; b < nb; ++b) { if (h[b * BD_TL_N] < t0) t0 = h[b * BD_TL_N]; if (h[b * BD_TL_N + 1] > t1) t1 = h[b * BD_TL_N + 1]; }
                for (int b = 0; b < nb; ++b) {
                    for (int i = 0; i < 5; ++i) ph[i] += (double)h[b * BD_TL_N + 3 + i];
                    rt += (double)(h[b * BD_TL_N + 1] - h[b * BD_TL_N]);
                    const double st = (h[b * BD_TL_N] - t0) / 100.0, en = (h[b * BD_TL_N + 1] - t0) / 100.0;
                    if (st > smax) smax = st; if (en < emin) emin = en;
                }
                const double cyc = ph[0] + ph[1] + ph[2] + ph[3] + ph[4];
                double ep[5] = {0, 0, 0, 0, 0};
                {
                    static unsigned long long he[BD_EPI_BLOCKS * 5];
                    const int ne = nb < BD_EPI_BLOCKS ? nb : BD_EPI_BLOCKS;
                    if (hipMemcpyFromSymbol(he, HIP_SYMBOL(g_bandd_epi), sizeof(unsigned long long) * 5 * ne) == hipSuccess)
                        for (int b = 0; b < ne; ++b) for (int i = 0; i < 5; ++i) ep[i] += (double)he[b * 5 + i] / ne / 1e3;
                }
                fprintf(stderr, "[timeline]   epilogue kcycles/wg (thread 0): loads + first barrier %.1f, transpose writes %.1f, barrier %.1f, reads + shortcut + split + stores %.1f, trailing barrier %.1f\n", ep[0], ep[1], ep[2], ep[3], ep[4]);
                fprintf(stderr, "[timeline] bandd<%d,%d,%dx%d,db%d,k%d> W=%d Cin=%d Cout=%d res=%d wgs=%d lds=%d | span %.1f us | wg mean %.1f us, last start %.1f, first end %.1f | kcycles/wg: prologue %.1f tops %.1f bodies %.1f drain %.1f epilogue %.1f | clock %.0f MHz\n",
                        BM, BN, NWM, NWN, BUFM, KG, a.Wi, a.Cin, a.Cout, a.res ? 1 : 0, gm * gn, lds, (t1 - t0) / 100.0, rt / nb / 100.0, smax, emin,
                        ph[0] / nb / 1e3, ph[1] / nb / 1e3, ph[2] / nb / 1e3, ph[3] / nb / 1e3, ph[4] / nb / 1e3, rt > 0 ? cyc / rt * 100.0 : 0.0);
                if (printed % 11 == 3) {                        // every now and then: the distribution behind the means
                    std::vector<double> dur(nb), en(nb);
                    std::map<unsigned long long, int> per_cu;
                    for (int b = 0; b < nb; ++b) {
                        dur[b] = (h[b * BD_TL_N + 1] - h[b * BD_TL_N]) / 100.0; en[b] = (h[b * BD_TL_N + 1] - t0) / 100.0;
                        const unsigned long long w = h[b * BD_TL_N + 2];
                        ++per_cu[((w >> 32) << 16) | (((w >> 13) & 7) << 8) | ((w >> 12) & 1) << 4 | ((w >> 8) & 15)];     // xcc, se, sh, cu
                    }
                    std::sort(dur.begin(), dur.end()); std::sort(en.begin(), en.end());
                    int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (auto& kv : per_cu) ++hist[kv.second < 7 ? kv.second : 7];
                    fprintf(stderr, "[timeline]   wg duration us min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | end us p10 %.1f p50 %.1f p90 %.1f max %.1f | CUs seen %d, with 1/2/3/4/5 wgs: %d %d %d %d %d\n",
                            dur[0], dur[nb / 10], dur[nb / 2], dur[nb * 9 / 10], dur[nb - 1], en[nb / 10], en[nb / 2], en[nb * 9 / 10], en[nb - 1],
                            (int)per_cu.size(), hist[1], hist[2], hist[3], hist[4], hist[5]);
                    // per workgroup slot (TG_ID): mean start, mean duration
                    double ss[16] = {0}, sd[16] = {0}; int sn[16] = {0};
                    for (int b = 0; b < nb; ++b) { const int tg = (int)((h[b * BD_TL_N + 2] >> 16) & 15); ++sn[tg]; ss[tg] += (h[b * BD_TL_N] - t0) / 100.0; sd[tg] += (h[b * BD_TL_N + 1] - h[b * BD_TL_N]) / 100.0; }
                    for (int tg = 0; tg < 16; ++tg) if (sn[tg]) fprintf(stderr, "[timeline]   slot %d: %d wgs, mean start %.1f us, mean duration %.1f us\n", tg, sn[tg], ss[tg] / sn[tg], sd[tg] / sn[tg]);
                }
                ++printed;
            }
        }
    }
#endif
    return hip_fail(hipGetLastError(), "conv_bandd_f16s3 launch");
}

// One list drives the mode table, the launch switch and the kernel names rocprofv3 prints:
//   X(index, BM, BN, waves along M, waves along N, MINW, double-buffered band, K groups, widest image, name suffix)
#define RTOD_BANDD_TILES(X) \
    X(0, 128, 128, 1, 4, 3, 0, 1, 94, "") X(1, 128, 128, 1, 4, 3, 1, 1, 94, ",db") X(2, 64, 128, 1, 4, 4, 1, 1, 94, ",db") \
    X(3, 64, 128, 1, 4, 4, 1, 2, 94, ",db,k2") X(4, 128, 128, 1, 4, 2, 1, 2, 94, ",db,k2") \
    X(5, 96, 128, 1, 4, 3, 1, 1, 94, ",db") X(6, 96, 128, 1, 4, 3, 1, 2, 94, ",db,k2") \
    X(7, 128, 128, 1, 4, 2, 0, 1, 160, ",w160") X(8, 48, 128, 1, 4, 3, 1, 2, 94, ",db,k2")
// (measured and dropped: 128x64 tiles of 128x16 strips, double-buffered, with and without split-K — 3-9 % behind the 32-column strips on every grid)

#define RTOD_X_INFO(idx, bm, bn, nwm, nwn, minw, db, kg, maxw, sfx) {bm, bn, "conv_bandd_f16s3<" #bm "x" #bn "," #nwm "x" #nwn sfx ">"},
static const ConvVariantInfo kBanddModes[BANDD_MODES] = { RTOD_BANDD_TILES(RTOD_X_INFO) };
#undef RTOD_X_INFO
const ConvVariantInfo& conv_bandd_mode_info(int idx) { return kBanddModes[idx < 0 || idx >= BANDD_MODES ? 0 : idx]; }
bool conv_bandd_wide_supported(int ksize, int stride, int pad, int cin, int w_in) {
    return ksize == 3 && stride == 1 && pad == 1 && cin % 32 == 0 && w_in > BANDD_MAX_W && w_in <= BANDD_WIDE_W;
}
int conv_bandd_mode_kg(int idx) {
#define RTOD_X_KG(i, bm, bn, nwm, nwn, minw, db, kg, maxw, sfx) if (idx == i) return kg;
    RTOD_BANDD_TILES(RTOD_X_KG)
#undef RTOD_X_KG
    return 0;
}
int conv_bandd_kernel_name(int idx, int epi, char* buf, size_t len) {
#define RTOD_X_NAME(i, bm, bn, nwm, nwn, minw, db, kg, maxw, sfx) \
    if (idx == i) return snprintf(buf, len, "void rtod::conv_bandd_f16s3_kernel<" #bm ", " #bn ", " #nwm ", " #nwn ", " #minw ", %d, " #db ", " #kg ", " #maxw ">(rtod::ConvArgs, int, int)", epi);
    RTOD_BANDD_TILES(RTOD_X_NAME)
#undef RTOD_X_NAME
    return -1;
}

int launch_conv_bandd_f16s3(const ConvArgs& a, int idx, hipStream_t s) {
    if (!a.in || !a.w_hi || !a.w_lo || !a.bias || !a.inv_scale || !a.out) { set_error("launch_conv_bandd: null pointer"); return RTOD_E_ARG; }
    if (a.kh != 3 || a.stride != 1 || a.pad != 1 || a.Cin % 32 || a.Wi > BANDD_WIDE_W || a.kw != 3 || a.Ho != a.Hi || a.Wo != a.Wi || a.dec.enabled || a.pw_wh) {
        set_error("launch_conv_bandd: unsupported shape (k=%d s=%d pad=%d Cin=%d W=%d)", a.kh, a.stride, a.pad, a.Cin, a.Wi); return RTOD_E_ARG;
    }
    if (a.in_ldc % 8 || a.in_coff % 8 || a.K != a.Kpad || a.K != 9 * a.Cin || a.Npad % 128) { set_error("launch_conv_bandd: bad view / K"); return RTOD_E_ARG; }
    if (a.in_bytes == 0 || a.in_bytes >= OOB || a.w_bytes == 0 || a.w_bytes >= OOB) { set_error("launch_conv_bandd: buffer extents"); return RTOD_E_ARG; }
    if ((uint64_t)a.B * a.Hi * a.Wi * a.in_ldc * 4ull > (uint64_t)a.in_bytes) { set_error("launch_conv_bandd: input view exceeds its buffer"); return RTOD_E_ARG; }
    switch (idx) {
#define RTOD_X_CASE(i, bm, bn, nwm, nwn, minw, db, kg, maxw, sfx) case i: if (a.Wi > maxw) break; return launch_bandd<bm, bn, nwm, nwn, minw, db, kg, maxw>(a, s);
        RTOD_BANDD_TILES(RTOD_X_CASE)
#undef RTOD_X_CASE
    }
    set_error("launch_conv_bandd: mode %d unsupported (image width %d)", idx, a.Wi);
    return RTOD_E_ARG;
}

}  // namespace rtod
