// 1x1 (pointwise) convolution as a barrier-free streaming kernel (split-precision f16 MFMA), gfx950.
//
// Reference semantics: conv(1x1, stride 1) -> BN(eval, folded) -> leaky / linear (src/darknet.py:467-501); same split
// activation / weight formats and the same three-product arithmetic as conv_igemm_f16s3.hip.
//
// Why its own kernel.  YOLOv3's 36 stand-alone 1x1 layers are each ~3 GFLOP over 20-70 MB: bandwidth-class work.  On the
// LDS-tiled kernels (register-staged or LDS-DMA ring alike) they sat at ~21 us — 14-17 % of the MFMA roof, 10-40 % of
// HBM: with K = 256 ... 1024 a tile is 8-32 k32 steps, every step pays barrier + LDS-read latency for only 12-24 MFMAs
// per wave, and the small layers (19x19: 2 888 pixels) cannot fill the chip with tiles at all.
//
// Structure (no barrier after the prologue, every wave independent):
//   * a workgroup = 12 waves = 192 consecutive pixels x BN output channels (BN = 32 / 64 / 128 chosen so that the layer
//     has <= 256 workgroups: one per CU); its weight slice [BN][K] (hi + lo planes, <= 128 KB) is DMA-ed into LDS once;
//   * a wave owns 16 pixels.  Activations go global -> VGPR directly in MFMA operand layout (lane = pixel x 8-channel
//     group: 16 B loads, 64 B contiguous per pixel and instruction), a 4-deep ring of 64-channel segments (64 VGPRs,
//     16 KB in flight per wave, 192 KB per CU) issued ahead with compiler-counted waits; weights come from LDS
//     (2 ds_read_b128 per 3 MFMAs);
//   * the product is computed transposed (out^T = W * act^T): the accumulator then holds, per lane, 4 consecutive output
//     channels of ONE pixel; with the weight rows of a 32-channel group interleaved over two MFMA tiles a lane ends up
//     with 8 consecutive channels, i.e. one 16-byte store per plane straight from registers — no LDS transpose, no
//     epilogue barrier.
// Per output the K products are summed chunk by chunk in ascending order, al*wh, ah*wl, ah*wh per chunk, whatever BN is:
// every tile choice gives the same bits.  (They differ in the last bits from the LDS-tiled kernels' — swapped MFMA
// operands — so a layer that this kernel supports always runs on it: frame independence.)
#include "conv_f16s3_common.h"
#include <atomic>
#include <cstdio>

namespace rtod {

constexpr int PW_WAVES = 12, PW_NT = PW_WAVES * 64;         // 192 pixels per workgroup
constexpr int PW_DEPTH = 4;                                 // 64-channel activation segments in flight per wave
constexpr int PW_TABLE = 2 * 128 * 4;                       // inv_scale / bias table (BN <= 128 floats each)

// panel row rho <-> output channel of the slice: tr_chan_of_row (conv_f16s3_common.h)
__device__ __forceinline__ int pw_chan_of_row(int rho) { return tr_chan_of_row(rho); }

__device__ __forceinline__ void pw_dma_pair(const __amdgpu_buffer_rsrc_t rsrc_hi, const __amdgpu_buffer_rsrc_t rsrc_lo, unsigned voffset,
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

// BN = 16 * NTL output channels per workgroup (NTL = 2, 4, 8 MFMA tiles).
// LDS: [K/32 panels][hi BN x 64 B | lo BN x 64 B] (rows permuted by pw_chan_of_row, 16-byte chunks XOR-swizzled by
// (row >> 1) & 3 like every other split kernel), then inv_scale[BN], bias[BN].
// NSEG = K / 64 is a template parameter: the segment loop is fully unrolled so that the compiler's vmcnt bookkeeping of the
// activation ring is exact (in a rolled loop it merges the states at the back edge and waits for nearly everything).
template <int NTL, int NSEG>
__global__ __launch_bounds__(PW_NT, 3)
void conv_pw_f16s3_kernel(const ConvArgs a, const int gm, const int n_strips) {
    constexpr int BN = 16 * NTL;
    constexpr int nseg = NSEG, nchunk = 2 * NSEG;
    constexpr int PANEL = BN * 64;                              // one plane of one k32 chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 15, lh = lane >> 4;
    const int M = a.B * a.Ho * a.Wo;
    const int bn = blockIdx.x / gm, wg_m = blockIdx.x - bn * gm;
    const int n0 = bn * BN;
    float* const tab = reinterpret_cast<float*>(smem + nchunk * 2 * PANEL);     // [0, BN): inv_scale * 8, [BN, 2 BN): bias * 8

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wh = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hi, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_lo, 0, a.w_bytes, 0x00020000);
    const unsigned PS = (unsigned)a.in_ldc * 4u;                 // bytes per pixel (hi plane + lo plane)
    const unsigned lo_plane = (unsigned)a.in_ldc * 2u;

    // ---- activation segments of a strip: 2 chunks x (hi, lo) per 64-channel segment, lane = (pixel lr, channel group lh)
    u32x4 ring[PW_DEPTH][4];
    unsigned pix_off = OOB;
    auto issue = [&](int seg, u32x4 (&slot)[4]) __attribute__((always_inline)) {
        const unsigned vo = pix_off == OOB ? OOB : pix_off + (unsigned)seg * 128u;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            slot[2 * q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo + q * 64u, 0, 0));
            slot[2 * q + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo + q * 64u, lo_plane, 0));
        }
    };
    auto strip_begin = [&](int strip) __attribute__((always_inline)) {
        const int m = strip * 16 + lr;
        pix_off = (strip < n_strips && m < M) ? (unsigned)m * PS + (unsigned)(a.in_coff + lh * 8) * 2u : OOB;
#pragma unroll
        for (int d = 0; d < PW_DEPTH; ++d)
            if (d < nseg) issue(d, ring[d]);                    // uniform
    };

    int strip = wg_m * PW_WAVES + wave;
    strip_begin(strip);                                          // the first strip's loads fly while the weights arrive

    // ---- prologue: weight slice -> LDS (all waves), scale / bias table
    {
        const int lrow = lane >> 2;
        const int n_rb = NTL;                                    // 16-row blocks per panel plane
        const int pieces = nchunk * n_rb;                        // (chunk, row block) pairs: one hi + one lo DMA each
        const unsigned lds0 = (unsigned)(size_t)smem;
        for (int p = wave; p < pieces; p += PW_WAVES) {          // wave-uniform
            const int kc = p / n_rb, rb = p - kc * n_rb;
            const int rho = rb * 16 + lrow;
            const int c = (lane & 3) ^ ((rho >> 1) & 3);
            const int n = n0 + pw_chan_of_row(rho);
            const unsigned vo = (unsigned)(n * 32 + c * 8) * 2u;         // K-chunk major planes: [chunk][Npad][32]
            const unsigned l = lds0 + (unsigned)(kc * 2 * PANEL + rb * 1024);
            pw_dma_pair(rs_wh, rs_wl, vo, (unsigned)kc * (unsigned)a.Npad * 64u, l, l + PANEL);
        }
        for (int i = tid; i < BN; i += PW_NT) {
            const int n = n0 + i;
            tab[i] = (n < a.Cout ? a.inv_scale[n] : 0.f) * SPLIT_SCALE;
            tab[BN + i] = (n < a.Cout ? a.bias[n] : 0.f) * SPLIT_SCALE;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's DMA pieces (and its first activation loads) landed
        __syncthreads();                                         // everybody's
    }

    const int w_lane = lr * 64 + ((lh ^ ((lr >> 1) & 3)) << 4);  // fragment address inside a 16-row block of a panel plane
    float amax = 0.f;
    _Float16* const oh = reinterpret_cast<_Float16*>(a.out) + a.out_coff + n0;

    for (; strip < n_strips; strip += gm * PW_WAVES) {
        f32x4 acc[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto consume = [&](const u32x4 (&slot)[4], int seg) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f16x8 xh = __builtin_bit_cast(f16x8, slot[2 * q]);
                const f16x8 xl = __builtin_bit_cast(f16x8, slot[2 * q + 1]);
                const unsigned char* pan = smem + (seg * 2 + q) * 2 * PANEL + w_lane;
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(pan + t * 1024);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(pan + PANEL + t * 1024);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[t], 0, 0, 0);
                }
            }
        };
#pragma unroll
        for (int seg = 0; seg < nseg; ++seg) {
            consume(ring[seg % PW_DEPTH], seg);
            if (seg + PW_DEPTH < nseg) issue(seg + PW_DEPTH, ring[seg % PW_DEPTH]);
        }
        // next strip's first segments fly during this strip's epilogue
        const int m = strip * 16 + lr;
        const int next = strip + gm * PW_WAVES;
        if (next < n_strips) strip_begin(next);                  // uniform

        // ---- epilogue straight from the accumulators: lane = pixel lr, channels 32P + 8*lh + {0..7} of tile pair P
#pragma unroll
        for (int P = 0; P < NTL / 2; ++P) {
            const int c0 = 32 * P + 8 * lh;
            const f32x4 i0 = *reinterpret_cast<const f32x4*>(tab + c0), i1 = *reinterpret_cast<const f32x4*>(tab + c0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(tab + BN + c0), b1 = *reinterpret_cast<const f32x4*>(tab + BN + c0 + 4);
            f16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = e < 4 ? acc[2 * P][e] : acc[2 * P + 1][e - 4];
                float v = s * (e < 4 ? i0[e] : i1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);      // (acc*inv + bias)*8 exactly (power of two)
                if (a.leaky) v = v > 0.f ? v : v * 0.1f;
                _Float16 h, l;
                split_f16(v, h, l, amax);
                ph[e] = h; pl[e] = l;
            }
            if (m < M && n0 + c0 < a.Cout) {
                _Float16* q = oh + (int64_t)m * 2 * a.out_ldc + c0;
                store_act16(q, ph, false);
                store_act16(q + a.out_ldc, pl, false);
            }
        }
    }
    split_overflow_report(a.ovf, amax);
}

static bool pw_nseg_built(int ntl, int nseg) {
    if (ntl == 2) return nseg == 2 || nseg == 4 || nseg == 6 || nseg == 8 || nseg == 12 || nseg == 16;
    if (ntl == 4) return nseg == 2 || nseg == 4 || nseg == 6 || nseg == 8;
    return nseg == 2 || nseg == 4;
}
bool conv_pw_supported(int ksize, int stride, int pad, int cin, int cout) {
    return ksize == 1 && stride == 1 && pad == 0 && cin % 64 == 0 && pw_nseg_built(2, cin / 64) && cout % 32 == 0;
}
bool conv_pw_mode_valid(int mode, int cin, int cout) {
    if (mode < 0 || mode >= PW_MODES) return false;
    const int bn = 32 << mode;
    return cout % bn == 0 && cin % 64 == 0 && pw_nseg_built(bn / 16, cin / 64) && (int64_t)cin * bn * 4 + PW_TABLE <= 160 * 1024;
}

static const ConvVariantInfo kPwModes[PW_MODES] = {
    {192, 32, "conv_pw_f16s3<192x32>"}, {192, 64, "conv_pw_f16s3<192x64>"}, {192, 128, "conv_pw_f16s3<192x128>"}};
const ConvVariantInfo& conv_pw_mode_info(int mode) { return kPwModes[mode < 0 || mode >= PW_MODES ? 0 : mode]; }
int conv_pw_kernel_name(int mode, int cin, char* buf, size_t len) {
    return snprintf(buf, len, "void rtod::conv_pw_f16s3_kernel<%d, %d>(rtod::ConvArgs, int, int)", 2 << mode, cin / 64);
}

template <int NTL, int NSEG>
static int launch_pw(const ConvArgs& a, hipStream_t s) {
    constexpr int BN = 16 * NTL;
    const int M = a.B * a.Ho * a.Wo;
    const int n_strips = (M + 15) / 16;
    const int slices = a.Cout / BN;
    const int lds = a.Kpad * BN * 4 + PW_TABLE;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return hip_fail(hipGetLastError(), "conv_pw_f16s3 device query");
    int gm = (n_strips + PW_WAVES - 1) / PW_WAVES;               // workgroups per slice if every wave took one strip
    const int cap = cus / slices > 0 ? cus / slices : 1;         // one workgroup per CU (12 waves at up to 168 VGPRs)
    if (gm > cap) gm = cap;                                      // persistent: waves loop over strips
    auto k = conv_pw_f16s3_kernel<NTL, NSEG>;
    static std::atomic<unsigned long long> attr_done{0};
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return hip_fail(hipGetLastError(), "conv_pw_f16s3 LDS attribute");
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(gm * slices), dim3(PW_NT), lds, s, a, gm, n_strips);
    return hip_fail(hipGetLastError(), "conv_pw_f16s3 launch");
}

int launch_conv_pw_f16s3(const ConvArgs& a, int mode, hipStream_t s) {
    if (!a.in || !a.w_hi || !a.w_lo || !a.bias || !a.inv_scale || !a.out) { set_error("launch_conv_pw: null pointer"); return RTOD_E_ARG; }
    if (!conv_pw_supported(a.kh, a.stride, a.pad, a.Cin, a.Cout) || a.kw != 1 || a.K != a.Cin || a.Kpad != a.K || a.dec.enabled || a.res || a.pw_wh) {
        set_error("launch_conv_pw: unsupported layer (k=%d s=%d Cin=%d Cout=%d)", a.kh, a.stride, a.Cin, a.Cout); return RTOD_E_ARG;
    }
    if (!conv_pw_mode_valid(mode, a.Cin, a.Cout)) { set_error("launch_conv_pw: mode %d invalid for Cin=%d Cout=%d", mode, a.Cin, a.Cout); return RTOD_E_ARG; }
    if (a.in_ldc % 8 || a.in_coff % 8 || a.out_ldc % 8 || a.out_coff % 8 || a.Ho != a.Hi || a.Wo != a.Wi) { set_error("launch_conv_pw: bad views"); return RTOD_E_ARG; }
    if (a.in_bytes == 0 || a.in_bytes >= OOB || a.w_bytes == 0 || a.w_bytes >= OOB) { set_error("launch_conv_pw: buffer extents"); return RTOD_E_ARG; }
    if ((uint64_t)a.B * a.Hi * a.Wi * a.in_ldc * 4ull > (uint64_t)a.in_bytes) { set_error("launch_conv_pw: input view exceeds its buffer"); return RTOD_E_ARG; }
    const int nseg = a.Kpad / 64;
#define RTOD_PW_CASE(ntl, ns) if (mode == (ntl == 2 ? 0 : ntl == 4 ? 1 : 2) && nseg == ns) return launch_pw<ntl, ns>(a, s);
    RTOD_PW_CASE(2, 2) RTOD_PW_CASE(2, 4) RTOD_PW_CASE(2, 6) RTOD_PW_CASE(2, 8) RTOD_PW_CASE(2, 12) RTOD_PW_CASE(2, 16)
    RTOD_PW_CASE(4, 2) RTOD_PW_CASE(4, 4) RTOD_PW_CASE(4, 6) RTOD_PW_CASE(4, 8)
    RTOD_PW_CASE(8, 2) RTOD_PW_CASE(8, 4)
#undef RTOD_PW_CASE
    set_error("launch_conv_pw: no instantiation for mode %d, K = %d", mode, a.Kpad);
    return RTOD_E_ARG;
}

}  // namespace rtod
