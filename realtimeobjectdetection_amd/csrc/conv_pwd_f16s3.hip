// 1x1 / stride 1 convolution (split-precision f16 MFMA), gfx950: activation slabs by LDS-DMA in full 128-byte lines, weight
// fragments straight from global memory, one workgroup barrier per 64 channels.  Round 4.
//
// Reference: conv -> BN -> leaky for the 1x1 layers (src/darknet.py:467-501, `nn.Conv2d(..., 1, 1, 0)` at 488-490).  Same K order
// (32-channel chunks ascending), same MFMA order per chunk (al*bh, ah*bl, ah*bh) and same epilogue arithmetic as the generic
// and ring tiles (conv_igemm_f16s3.hip, conv_ring_f16s3.hip): bit-identical, so these tiles are further autotune candidates
// of every plain 1x1 layer with Cin % 64 == 0.
//
// Why.  The ring / generic tiles spend ~1 500 cycles per k32 step on 384 of MFMA: both operands go through LDS, one workgroup
// barrier per step, 16-row x 64-byte LDS-DMA pieces (half-line requests: twice the address work per byte).  Here (the recipe that
// took 12-20 % out of the 3x3 band layers, conv_bandd_f16s3.hip):
//   * a wave owns ALL BM rows of a 32-column strip: its B fragments are 1 KiB runs of the packed weight planes
//     ([chunk][Npad][32] f16), loaded with `buffer_load_dwordx4` into registers two steps ahead — no LDS write, no B fragment read,
//     no weight byte loaded twice in a workgroup;
//   * the A operand is staged in SLABS of 64 channels: one LDS-DMA piece = 8 pixels x 128 contiguous bytes (the 64 hi halves, or
//     the 64 lo halves, of a pixel's slab — a full cache line per row), NST slabs in a ring, ONE barrier per slab (two k32 steps);
//   * LDS per workgroup is the ring alone (BM x 256 B per slab): 64-row tiles of four waves fit three to a CU.
//
// Slab image in LDS: 16-row blocks of 4 KiB = [hi rows 0-7][hi rows 8-15][lo rows 0-7][lo rows 8-15], each piece 8 rows x 128 B.
// Inside a piece the 16-byte slot of (row r, channel group c of 8) is c ^ r (c = 0..7 over the slab's 64 channels): the four
// 16-lane groups of a ds_read_b128 (tools/lds_bank_sim.py) then touch every bank exactly once per k32 half.  A DMA instruction
// writes lane l's 16 bytes at piece + 16 l, so lane l = (row l >> 3, slot l & 7) fetches channel group (l & 7) ^ (l >> 3): the
// swizzle is applied on the SOURCE address and every row is still one contiguous 128-byte request.
//
// vmcnt bookkeeping (in-order retirement; LDS-DMA counts too).  Issue order: prologue = slabs 0 .. NST-2, B(0), B(1); iteration j =
// [barrier] slab j+NST-1, then for q = 0, 1: B(2j+q+2), wait for B(2j+q).  Every wave issues the same number of instructions per
// slab (pieces past the end of K have an out-of-range source and write zeros into a slot nobody reads), so the B wait is always
// vmcnt(8 + 2 NPP), and it implies that this wave's pieces of slab j+1 have landed by the end of iteration j.
#include "conv_bandd_common.h"
#include <atomic>

namespace rtod {

// hi and lo piece of one 8-row group: lane l's 16 bytes land at lds + 16 l; the lo piece sits 2 KiB behind the hi piece
__device__ __forceinline__ void pwd_dma_pair(const __amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soff_hi, unsigned soff_lo, unsigned lds_hi) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "s_add_u32 m0, %5, 0x800\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(rsrc), "s"(soff_hi), "s"(soff_lo), "s"(lds_hi)
        : "memory", "scc");
}

template <int N> __device__ __forceinline__ void pwd_wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt literal");
#define RTOD_VMCNT_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RTOD_VMCNT_CASE(8) RTOD_VMCNT_CASE(10) RTOD_VMCNT_CASE(12) RTOD_VMCNT_CASE(14) RTOD_VMCNT_CASE(16) RTOD_VMCNT_CASE(18) RTOD_VMCNT_CASE(20)
    RTOD_VMCNT_CASE(22) RTOD_VMCNT_CASE(24) RTOD_VMCNT_CASE(26) RTOD_VMCNT_CASE(28) RTOD_VMCNT_CASE(30) RTOD_VMCNT_CASE(32) RTOD_VMCNT_CASE(36)
    RTOD_VMCNT_CASE(40) RTOD_VMCNT_CASE(44) RTOD_VMCNT_CASE(48)
    else static_assert(N < 0, "add the vmcnt literal");
#undef RTOD_VMCNT_CASE
}

// BM x (32 NW) workgroup tile, NW waves, each a BM x 32 strip; NST slabs of 64 channels in the LDS ring.
template <int BM, int NW, int NST, int MINW, int EPI>
__global__ __launch_bounds__(NW * 64, MINW)
void conv_pwd_f16s3_kernel(const ConvArgs a, const int grid_m, const int grid_n) {
    constexpr int BN = NW * 32, NT = NW * 64, TM = BM / 16, TN = 2;
    constexpr int SLAB = BM * 256;                              // bytes of one slab: BM rows x 64 channels x (hi + lo)
    constexpr int NP = BM / 8;                                  // 8-row groups (hi + lo DMA pairs) of a slab
    constexpr int NPP = (NP + NW - 1) / NW;                     // ... per wave: every wave issues the same number (counted waits); a group past
                                                                // the slab has an out-of-range source and lands in a 4 KiB dump behind the ring
    static_assert(BM % 16 == 0 && NPP >= 1 && NPP <= 8, "slab pieces per wave");
    static_assert(NST >= 2 && NST <= 4, "ring depth");
    constexpr int RG = BM * BN * 4 <= 32768 ? BM : (BM / 2) * BN * 4 <= 32768 ? BM / 2 : BM / 4;   // epilogue rows per pass (<= 32 KB of fp32)
    constexpr int WAIT_B = 8 + 2 * NPP;                         // younger than the awaited B set: one B set, one slab, one B set
    constexpr int WAIT_SLAB = 8 + (NST - 2) * (2 * NPP + 8);    // younger than slab j at the top of iteration j (j >= NST - 1)
    constexpr int WAIT_SLAB_PRO = 8 + (NST - 2) * 2 * NPP;      // ... for the slabs issued by the prologue (a lower bound of what is younger)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nwg = grid_m * grid_n;
    int bid = blockIdx.x;
    if (!a.xcd_by_n) {                                          // XCD x (= blockIdx % 8) takes a contiguous range of tiles
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int bm = bid / grid_n, bn = bid - bm * grid_n;
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 15, lh = lane >> 4;
    const int M = a.B * a.Hi * a.Wi;
    const int m0 = bm * BM;
    const unsigned PS = (unsigned)a.in_ldc * 4u;                // bytes of one pixel (hi plane + lo plane)
    const unsigned lo_plane = (unsigned)a.in_ldc * 2u;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int NS = a.Cin / 64;                                  // slabs

    // ---- slab DMA: 8-row group p = wave + k NW -> block p >> 1, half p & 1; lane -> row lane >> 3, slot lane & 7 <- channel group slot ^ row
    unsigned dma_vo[NPP];
#pragma unroll
    for (int k = 0; k < NPP; ++k) {
        const int p = wave + k * NW;
        const int r = lane >> 3;
        const int m = m0 + p * 8 + r;
        dma_vo[k] = (p < NP && m < M) ? (unsigned)m * PS + (unsigned)(a.in_coff + (((lane & 7) ^ r) << 3)) * 2u : OOB;
    }
    auto dma_slab = [&](int js) __attribute__((always_inline)) {
        const bool live = js < NS;
        const unsigned soff = (unsigned)js * 128u;
        const unsigned base = lds0 + (unsigned)((js % NST) * SLAB);
#pragma unroll
        for (int k = 0; k < NPP; ++k) {
            const int p = wave + k * NW;
            pwd_dma_pair(rs_a, live ? dma_vo[k] : OOB, soff, lo_plane + soff,
                         p < NP ? base + (unsigned)((p >> 1) * 4096 + (p & 1) * 1024) : lds0 + (unsigned)(NST * SLAB));
        }
    };
    // ---- B fragments: lane (lr, lh) <- weight row n0 + 16 j + lr, 16-byte chunk lh of the step's panel; four register sets (t % 4)
    const int wn = wave;
    const int n0 = bn * BN + wn * 32;
    const unsigned bvoff = n0 + lr < a.Npad ? (unsigned)((n0 + lr) * 32 + lh * 8) * 2u : OOB;     // Npad % 128 == 0 and BN % 128 == 0: a strip is inside or outside
    const unsigned wchunk = (unsigned)a.Npad * (HBK * 2);       // bytes of one k32 panel of a weight plane
    const int nsteps = 2 * NS;
    u32x4 Bq[4][TN][2];
    auto load_b = [&](int t, u32x4 (&q)[TN][2]) __attribute__((always_inline)) {
        const unsigned vo = t < nsteps ? bvoff : OOB;
        const unsigned koff = (unsigned)t * wchunk;
        q[0][0] = bandd_load_b<0>(rs_wh, vo, koff);
        q[0][1] = bandd_load_b<0>(rs_wl, vo, koff);
        q[1][0] = bandd_load_b<1024>(rs_wh, vo, koff);
        q[1][1] = bandd_load_b<1024>(rs_wl, vo, koff);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    // ---- A fragments: unit (q, i) = k32 half q of the slab, 16-row tile i
    typedef const f16x8 __attribute__((address_space(3))) lds_f16x8;
    const int rbase = (lr >> 3) * 1024 + (lr & 7) * 128;
    int xq0 = rbase + ((lh ^ (lr & 7)) << 4);
    int xq1 = rbase + (((4 + lh) ^ (lr & 7)) << 4);
    f16x8 Ah[3], Al[3];
    auto read_unit = [&](int q, int i, int slotoff, f16x8& h, f16x8& l) __attribute__((always_inline)) {
        const int o = (q ? xq1 : xq0) + slotoff;
        h = *reinterpret_cast<lds_f16x8*>((unsigned)(o + 4096 * i));
        l = *reinterpret_cast<lds_f16x8*>((unsigned)(o + 4096 * i + 2048));
    };

    // ---- prologue: slabs 0 .. NST-2, B sets of steps 0 and 1
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) dma_slab(s);
    load_b(0, Bq[0]);
    load_b(1, Bq[1]);

    __builtin_amdgcn_s_setprio(2);
    // one iteration = one slab = two k32 steps; `par` = j & 1 selects the B register sets (t % 4 = 2 par + q)
    auto slab_body = [&](int j, auto par_c) __attribute__((always_inline)) {
        constexpr int par = decltype(par_c)::value;
        asm volatile("" : "+v"(xq0), "+v"(xq1));                // not loop-invariant: the unit addresses are recomputed per slab, not held in registers
        if (j < NST - 1) pwd_wait_vmcnt<WAIT_SLAB_PRO>(); else pwd_wait_vmcnt<WAIT_SLAB>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                           // slab j complete in every wave's view; slab j - 1 released
        dma_slab(j + NST - 1);
        const int slotoff = (int)lds0 + (j % NST) * SLAB;
        read_unit(0, 0, slotoff, Ah[0], Al[0]);
        read_unit(1 / TM, 1 % TM, slotoff, Ah[1], Al[1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2 * TM; ++u) {
            const int q = u / TM, i = u % TM;
            if (i == 0) {                                       // step head: B set of step t + 2, wait for this step's
                if (q == 0) load_b(2 * j + 2, Bq[(2 * par + 2) % 4]); else load_b(2 * j + 3, Bq[(2 * par + 3) % 4]);
                pwd_wait_vmcnt<WAIT_B>();
                if (q == 0) bandd_tie<TN>(Bq[2 * par]); else bandd_tie<TN>(Bq[2 * par + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (u + 2 < 2 * TM) read_unit((u + 2) / TM, (u + 2) % TM, slotoff, Ah[(u + 2) % 3], Al[(u + 2) % 3]);
            const f16x8 ah = Ah[u % 3], al = Al[u % 3];
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const f16x8 bh = __builtin_bit_cast(f16x8, Bq[2 * par + q][jn][0]), bl = __builtin_bit_cast(f16x8, Bq[2 * par + q][jn][1]);
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[i][jn], 0, 0, 0);
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[i][jn], 0, 0, 0);
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i][jn], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll 1
    for (int j = 0; j < NS; j += 2) {
        slab_body(j, std::integral_constant<int, 0>{});
        if (j + 1 < NS) slab_body(j + 1, std::integral_constant<int, 1>{});
    }
    // drain: trailing (out-of-range) B sets and slab pieces have landed before the ring becomes the transpose tile
    pwd_wait_vmcnt<0>();
    bandd_tie<TN>(Bq[0]); bandd_tie<TN>(Bq[1]); bandd_tie<TN>(Bq[2]); bandd_tie<TN>(Bq[3]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);

    bandd_epilogue<BM, BN, BM, 32, NT, RG, EPI == EPI_SPLIT_RES, 1>(a, acc, smem, bm, bn, tid, 0, wn, lr, lh, M, 0);
}

template <int BM, int NW, int NST, int MINW>
static int launch_pwd(const ConvArgs& a, hipStream_t s) {
    constexpr int BN = NW * 32, NT = NW * 64;
    constexpr int RG = BM * BN * 4 <= 32768 ? BM : (BM / 2) * BN * 4 <= 32768 ? BM / 2 : BM / 4;
    const int M = a.B * a.Ho * a.Wo;
    const int gm = (M + BM - 1) / BM, gn = (a.Cout + BN - 1) / BN;
    ConvArgs ax = a;
    ax.xcd_by_n = (gn % 8 == 0 && (int64_t)a.Cout * a.K > (int64_t)M * a.Cin) ? 1 : 0;
    constexpr int ring_bytes = NST * BM * 256 + ((BM / 8) % NW ? 4096 : 0), epi_bytes = RG * BN * 4;
    constexpr int lds = ring_bytes > epi_bytes ? ring_bytes : epi_bytes;
    static_assert(lds <= 160 * 1024, "LDS");
    auto k_res = conv_pwd_f16s3_kernel<BM, NW, NST, MINW, EPI_SPLIT_RES>;
    auto k_plain = conv_pwd_f16s3_kernel<BM, NW, NST, MINW, EPI_SPLIT>;
    if constexpr (lds > 64 * 1024) {                            // > 64 KiB of dynamic LDS needs the opt-in, once per device
        static std::atomic<unsigned long long> attr_done{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError(), "conv_pwd_f16s3 hipGetDevice");
        if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_res), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(k_plain), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return hip_fail(hipGetLastError(), "conv_pwd_f16s3 LDS attribute");
            attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
    }
    if (a.res) hipLaunchKernelGGL(k_res, dim3(gm * gn), dim3(NT), lds, s, ax, gm, gn);
    else hipLaunchKernelGGL(k_plain, dim3(gm * gn), dim3(NT), lds, s, ax, gm, gn);
    return hip_fail(hipGetLastError(), "conv_pwd_f16s3 launch");
}

// One list drives the mode table, the launch switch and the kernel names rocprofv3 prints:
//   X(index, BM, waves (BN = 32 waves), ring slabs, MINW)
#define RTOD_PWD_TILES(X) \
    X(0, 64, 4, 3, 3) X(1, 32, 4, 3, 3) X(2, 128, 4, 3, 1) X(3, 64, 8, 3, 1) X(4, 64, 4, 2, 3) X(5, 32, 4, 4, 3) X(6, 96, 4, 2, 3) X(7, 96, 4, 3, 2) X(8, 48, 4, 3, 3) X(9, 48, 4, 4, 3) X(10, 64, 2, 2, 4)

#define RTOD_X_INFO(idx, bm, nw, nst, minw) {bm, nw * 32, "conv_pwd_f16s3<" #bm "x" #nw "w,r" #nst ">"},
static const ConvVariantInfo kPwdModes[PWD_MODES] = { RTOD_PWD_TILES(RTOD_X_INFO) };
#undef RTOD_X_INFO
const ConvVariantInfo& conv_pwd_mode_info(int idx) { return kPwdModes[idx < 0 || idx >= PWD_MODES ? 0 : idx]; }

bool conv_pwd_supported(int ksize, int stride, int pad, int cin) { return ksize == 1 && stride == 1 && pad == 0 && cin >= 64 && cin % 64 == 0; }

int conv_pwd_kernel_name(int idx, int epi, char* buf, size_t len) {
#define RTOD_X_NAME(i, bm, nw, nst, minw) \
    if (idx == i) return snprintf(buf, len, "void rtod::conv_pwd_f16s3_kernel<" #bm ", " #nw ", " #nst ", " #minw ", %d>(rtod::ConvArgs, int, int)", epi);
    RTOD_PWD_TILES(RTOD_X_NAME)
#undef RTOD_X_NAME
    return -1;
}

int launch_conv_pwd_f16s3(const ConvArgs& a, int idx, hipStream_t s) {
    if (!a.in || !a.w_hi || !a.w_lo || !a.bias || !a.inv_scale || !a.out) { set_error("launch_conv_pwd: null pointer"); return RTOD_E_ARG; }
    if (!conv_pwd_supported(a.kh, a.stride, a.pad, a.Cin) || a.kw != 1 || a.Ho != a.Hi || a.Wo != a.Wi || a.dec.enabled || a.pw_wh) {
        set_error("launch_conv_pwd: unsupported shape (k=%d s=%d pad=%d Cin=%d)", a.kh, a.stride, a.pad, a.Cin); return RTOD_E_ARG;
    }
    if (a.in_ldc % 8 || a.in_coff % 8 || a.K != a.Kpad || a.K != a.Cin || a.Npad % 128) { set_error("launch_conv_pwd: bad view / K"); return RTOD_E_ARG; }
    if (a.in_bytes == 0 || a.in_bytes >= OOB || a.w_bytes == 0 || a.w_bytes >= OOB) { set_error("launch_conv_pwd: buffer extents"); return RTOD_E_ARG; }
    if ((uint64_t)a.B * a.Hi * a.Wi * a.in_ldc * 4ull > (uint64_t)a.in_bytes) { set_error("launch_conv_pwd: input view exceeds its buffer"); return RTOD_E_ARG; }
    switch (idx) {
#define RTOD_X_CASE(i, bm, nw, nst, minw) case i: return launch_pwd<bm, nw, nst, minw>(a, s);
        RTOD_PWD_TILES(RTOD_X_CASE)
#undef RTOD_X_CASE
    }
    set_error("launch_conv_pwd: mode %d unsupported", idx);
    return RTOD_E_ARG;
}

}  // namespace rtod
