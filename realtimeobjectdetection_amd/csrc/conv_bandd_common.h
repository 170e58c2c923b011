// Pieces shared by the kernels that load their weight fragments straight from global memory (conv_bandd_f16s3.hip: 3x3 band;
// conv_pwd_f16s3.hip: 1x1): counted vmcnt waits, the raw fragment load, register ties, and the epilogue in row passes.
#pragma once
#include "conv_f16s3_common.h"

namespace rtod {

#ifdef RTOD_TIMELINE
// diagnostic build: shader cycles of the epilogue's phases per workgroup (thread 0): [0] loads issued + first barrier, [1] transpose writes,
// [2] barrier, [3] tile reads + shortcut + split + stores, [4] trailing barrier
constexpr int BD_EPI_BLOCKS = 2048;
static __device__ unsigned long long g_bandd_epi[BD_EPI_BLOCKS * 5];
#define BD_ESTAMP(slot) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); et_[slot] += tn_ - eprev_; eprev_ = tn_; }
#else
#define BD_ESTAMP(slot)
#endif

template <int N> __device__ __forceinline__ void bandd_wait_vmcnt() {
    static_assert(N >= 0 && N <= 16, "vmcnt literal");
#define RTOD_VMCNT_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RTOD_VMCNT_CASE(2) RTOD_VMCNT_CASE(4) RTOD_VMCNT_CASE(5) RTOD_VMCNT_CASE(6) RTOD_VMCNT_CASE(8) RTOD_VMCNT_CASE(9) RTOD_VMCNT_CASE(10) RTOD_VMCNT_CASE(12) RTOD_VMCNT_CASE(16)
#undef RTOD_VMCNT_CASE
}

// the same with a count that is a constant only after unrolling (a loop variable of a fully unrolled loop): the switch folds
__device__ __forceinline__ void bandd_wait_vmcnt_folded(int n) {
    switch (n) {
#define RTOD_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        RTOD_VMCNT_CASE(4) RTOD_VMCNT_CASE(5) RTOD_VMCNT_CASE(6) RTOD_VMCNT_CASE(8) RTOD_VMCNT_CASE(9) RTOD_VMCNT_CASE(10)
#undef RTOD_VMCNT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;          // (never taken: stricter, still correct)
    }
}

// wait for all but the NBASE + 2 n youngest operations, n (0 ... 5) wave-uniform: the band pieces issued after the awaited B set
template <int NBASE> __device__ __forceinline__ void bandd_wait_vmcnt_plus(int n) {
    static_assert(NBASE == 4 || NBASE == 8, "two B sets of 2 or 4 loads");
    asm volatile("" : "+s"(n));                                 // opaque: left visible, the loop-invariant n unswitches the whole chunk loop six ways
    if constexpr (NBASE == 8) {
        if (n == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    } else {
        if (n == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    }
}

// raw buffer load with an instruction offset (the 16-column tile of the strip: j KiB), hidden from the compiler's waitcnt pass
template <int OFF> __device__ __forceinline__ u32x4 bandd_load_b(const __amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voffset), "s"(rsrc), "s"(soffset), "n"(OFF) : "memory");
    return v;
}

// hi and lo piece of one 16-row band block: lane l's 16 bytes land at lds + 16 l (M0 written in the statement that uses it)
__device__ __forceinline__ void bandd_dma_pair(const __amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soff_hi, unsigned soff_lo, unsigned lds_hi) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "s_add_u32 m0, %5, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffset), "s"(rsrc), "s"(soff_hi), "s"(soff_lo), "s"(lds_hi)
        : "memory", "scc");
}

template <int TN> __device__ __forceinline__ void bandd_tie(u32x4 (&q)[TN][2]) {
    static_assert(TN >= 1 && TN <= 2, "strip width");
    if constexpr (TN == 1) asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]) :: "memory");
    else asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[1][0]), "+v"(q[1][1]) :: "memory");
}

// ---- epilogue: scale / bias / activation, LDS transpose in passes of RG rows, split-format store (+ residual).
// Arithmetic and expression shapes are those of conv_f16s3_epilogue (conv_f16s3_common.h): the same bits.  Unlike that
// function a pass may cover a PART of a wave's rows (a wave owns all BM rows of its strip; the whole tile would need 64 KB).
template <int BM, int BN, int WM, int WN, int NT, int RG, bool RES, int KG>
__device__ __forceinline__ void bandd_epilogue(const ConvArgs& a, f32x4 (&acc)[WM / 16][WN / 16], unsigned char* smem, int bm, int bn, int tid,
                                               int wm, int wn, int lr, int lh, int M, int kg) {
    constexpr int TM = WM / 16, TN = WN / 16, MT = 16, NE = 4, TS = BN;
    static_assert(BM % RG == 0 && RG % 16 == 0, "epilogue pass");
#ifdef RTOD_TIMELINE
    unsigned long long et_[5] = {0, 0, 0, 0, 0};
    unsigned long long eprev_ = __builtin_amdgcn_s_memtime();
#endif
    float* T = reinterpret_cast<float*>(smem);
    float amax = 0.f;
    const float escale = SPLIT_SCALE;
    constexpr int GPR = BN / 8;
    constexpr int NG = (RG * GPR + NT - 1) / NT;
    // every global load of the epilogue up front, for ALL passes: bias / scale of the wave's column tiles, then the residual
    // operands (vmcnt retires in order: the transpose only has to wait for the first).  Loaded per pass, every pass paid the
    // memory latency again, between two barriers.
    constexpr int NP = BM / RG;
    float bias_j[TN], inv_j[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = bn * BN + wn * WN + j * MT + lr;
        const int nc = n < a.Cout ? n : 0;
        const float b = a.bias[nc], iv = a.inv_scale[nc];
        bias_j[j] = n < a.Cout ? b : 0.f; inv_j[j] = n < a.Cout ? iv : 0.f;
    }
    f16x8 rq_h[RES ? NP : 1][RES ? NG : 1], rq_l[RES ? NP : 1][RES ? NG : 1];
    // Item (pass p, i) of a thread is row p RG + tid / GPR + i (NT / GPR), 8 channels at (tid % GPR) 8 when NT is a multiple of GPR
    // (every tile of this family): row and channel split ONCE, addresses advance by a constant — the per-item 64-bit multiplies of
    // `m * 2 * ldc` (quarter-rate v_mul_lo_u32) were ~5 % of the epilogue's vector time.
    static_assert(NT % GPR == 0, "epilogue items: constant row step");
    constexpr int RSTEP = NT / GPR;
    const int er = tid / GPR, ec8 = (tid - er * GPR) * 8;
    const bool ecol = bn * BN + ec8 < a.Cout;
    if constexpr (RES) {
        const _Float16* rh0 = reinterpret_cast<const _Float16*>(a.res) + a.res_coff + bn * BN;
        const int64_t rstep = (int64_t)RSTEP * 2 * a.res_ldc;
        const _Float16* const q0 = rh0 + (int64_t)(bm * BM + er) * 2 * a.res_ldc + ec8;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const _Float16* q = q0 + (int64_t)(p * RG) * 2 * a.res_ldc;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int m = bm * BM + p * RG + er + i * RSTEP;
                // rows past M / channels past Cout are never stored: their operand only has to come from a valid address
                const _Float16* qq = (er + i * RSTEP < RG && m < M && ecol) ? q : rh0;      // (a pass's last item may be partial: RG GPR items over NT threads)
                rq_h[p][i] = *reinterpret_cast<const f16x8*>(qq);
                rq_l[p][i] = *reinterpret_cast<const f16x8*>(qq + a.res_ldc);
                q += rstep;
            }
        }
    }
    __syncthreads();                                            // every wave has read its last fragments: the band becomes the transpose tile
    BD_ESTAMP(0)
#pragma unroll
    for (int rg = 0; rg < BM; rg += RG) {
        if constexpr (KG == 2) {                                         // K group 1 deposits its raw sums, group 0 adds its own
            if (kg == 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r0 = wm * WM + i * MT;
                    if (r0 >= rg && r0 < rg + RG) {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int e = 0; e < NE; ++e) T[(r0 - rg + e + 4 * lh) * TS + wn * WN + j * MT + lr] = acc[i][j][e];
                    }
                }
            }
            __syncthreads();
        }
        if (KG == 1 || kg == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = wn * WN + j * MT + lr;
                const float bias = bias_j[j] * escale, inv = inv_j[j] * escale;
                auto col = [&](auto act) {                                   // 0 linear, 1 leaky, 2 SiLU
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int r0 = wm * WM + i * MT;
                        if (r0 >= rg && r0 < rg + RG) {                       // compile-time for one wave along M
#pragma unroll
                            for (int e = 0; e < NE; ++e) {
                                const int rl = r0 - rg + e + 4 * lh;
                                float s = acc[i][j][e];
                                if constexpr (KG == 2) s += T[rl * TS + nl];
                                float v = s * inv + bias;
                                if constexpr (decltype(act)::value == 2) v = silu_scaled(v, 1.0f / escale);
                                else if constexpr (decltype(act)::value == 1) v = __builtin_fmaxf(v, v * 0.1f);
                                T[rl * TS + nl] = v;
                            }
                        }
                    }
                };
                if (a.leaky == 2) col(std::integral_constant<int, 2>{});
                else if (a.leaky) col(std::integral_constant<int, 1>{});
                else col(std::integral_constant<int, 0>{});
            }
        }
        BD_ESTAMP(1)
        __syncthreads();
        BD_ESTAMP(2)
        _Float16* oq = reinterpret_cast<_Float16*>(a.out) + a.out_coff + bn * BN + (int64_t)(bm * BM + rg + er) * 2 * a.out_ldc + ec8;
        const int64_t ostep = (int64_t)RSTEP * 2 * a.out_ldc;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi, oq += ostep) {
            const int r = er + gi * RSTEP;
            const int m = bm * BM + rg + r;
            if (r >= RG || m >= M || !ecol) continue;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(T + r * TS + ec8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(T + r * TS + ec8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if constexpr (RES) {
                const f16x8 qh = rq_h[rg / RG][gi], ql = rq_l[rg / RG][gi];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)qh[e] + (float)ql[e];
            }
            f16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) { _Float16 h, l; split_f16(v[e], h, l, amax); ph[e] = h; pl[e] = l; }
            store_act16(oq, ph, false);
            store_act16(oq + a.out_ldc, pl, false);
        }
        BD_ESTAMP(3)
        if (rg + RG < BM) __syncthreads();
        BD_ESTAMP(4)
    }
#ifdef RTOD_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < BD_EPI_BLOCKS) for (int i = 0; i < 5; ++i) g_bandd_epi[blockIdx.x * 5 + i] = et_[i];
#endif
    split_overflow_report(a.ovf, amax);
}

}  // namespace rtod
