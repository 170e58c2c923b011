// Internal declarations shared by the kernels, the plan and the C ABI (not installed).
#pragma once
#include <hip/hip_runtime.h>
// Diagnostic builds (never the product library): -DRTOD_STAMPS = in-kernel s_memtime phase attribution (synchronises after
// every launch), -DRTOD_DIAG = the RTOD_DBG_ZERO timing knobs alone (zero-record descriptors, skipped epilogue); stamps imply diag.
#if defined(RTOD_STAMPS) && !defined(RTOD_DIAG)
#define RTOD_DIAG 1
#endif
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rtod.h"

namespace rtod {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define RTOD_HIP(expr)                                              \
    do {                                                            \
        hipError_t _e = (expr);                                     \
        if (_e != hipSuccess) return ::rtod::hip_fail(_e, #expr);   \
    } while (0)

// NHWC view of an activation: element (b,y,x,c) at base[((b*H + y)*W + x)*ldc + coff + c]
struct View {
    int split = 0;     // 0: fp32 NHWC; 1: split f16 hi/lo planes per pixel (conv_igemm_f16s3.hip), same bytes
    float* base = nullptr;
    int64_t ldc = 0;   // floats between consecutive pixels (>= C: concat buffers are wider)
    int coff = 0;      // first channel of this view inside the pixel
    int C = 0, H = 0, W = 0;
};

// Fused YOLO head decode (reference: src/util.py:193-237) applied in a conv epilogue or by the
// stand-alone kernel.
struct DecodeArgs {
    int enabled = 0;
    int G = 0;               // grid size
    int attrs = 0;           // 5 + classes
    int n_anchors = 0;
    int train = 0;           // TRAIN=True: sigmoids only
    int v5 = 0;              // cfg extension `[yolo] decode=v5` (YOLOv5-style heads): xy = (2 s - 0.5 + g) * stride, wh = (2 s)^2 * anchor
                             // (aw / ah then hold the anchors in PIXELS), everything else sigmoid; not reference behaviour
    float stride = 0.f;      // inp_dim // G
    float aw[4] = {0, 0, 0, 0};   // fp32(anchor_w / stride)
    float ah[4] = {0, 0, 0, 0};
    int64_t img_stride = 0;  // total_rows * attrs
    int64_t head_off = 0;    // row_offset * attrs
};

struct ConvArgs {
    const float* in = nullptr;  int64_t in_ldc = 0;  int in_coff = 0;
    int B = 0, Hi = 0, Wi = 0, Cin = 0;        // Cin = stored channels of the input view
    const float* w = nullptr;                   // packed [Npad][Kpad], k = (ky*kw+kx)*Cin + c
    const float* bias = nullptr;                // [Npad]
    int K = 0, Kpad = 0;
    int Npad = 0;                               // split path: rows of one K-chunk panel of the weight planes ([Kpad/32][Npad][32] f16)
    int kh = 1, kw = 1, stride = 1, pad = 0;
    int Ho = 0, Wo = 0, Cout = 0;
    float* out = nullptr;       int64_t out_ldc = 0; int out_coff = 0;
    const float* res = nullptr; int64_t res_ldc = 0; int res_coff = 0;   // fused shortcut
    int leaky = 0;                              // activation code: 0 linear, 1 leaky(0.1), 2 SiLU (apply_act)
    DecodeArgs dec;
    // split-precision path (conv_igemm_f16s3): pre-split, pre-scaled f16 weight planes [Npad][Kpad]
    // Weight planes are K-chunk major: [Kpad/32 chunks][Npad rows][32 halves], chunk kc = (c/32)*kh*kw + tap.  One (chunk, tap)
    // stage of a tile is then BN consecutive 64-byte rows: contiguous, full 128-byte lines (row-major [Npad][Kpad] planes made every
    // stage BN separate 64-byte pieces at a stride of Kpad*2 bytes: half-line requests, twice the address traffic for the same bytes)
    const _Float16* w_hi = nullptr;
    const _Float16* w_lo = nullptr;
    const float* inv_scale = nullptr;           // [Npad]: 1 / (2^e_n * SPLIT_SCALE)
    unsigned in_bytes = 0, w_bytes = 0;         // extents of the input buffer / one weight plane (buffer-load range check)
    int dbg = 0;                                // timing experiments, diagnostic build only (-DRTOD_STAMPS reads RTOD_DBG_ZERO); the product library never sets or reads it
    int32_t* ovf = nullptr;                     // split-format producers: device word that gets 1 OR-ed in when an activation saturates the f16 range
    int xcd_by_n = 0;                           // split kernels: workgroup -> XCD by output-channel tile instead of by pixel tile (see launch_band)
    int out_split = 0;                          // exact-fp32 kernel only: write the output in the split format
    // exact-fp32 kernel only: K slices (conv_igemm_f32.hip).  slice_chunks > 0: the K sum is formed slice by slice (a property of
    // the layer); partial != nullptr: one workgroup per slice, raw sums to this scratch ([slices][M][Npad] floats), then a reduction
    int slice_chunks = 0;
    float* partial = nullptr;
    int64_t partial_floats = 0;
    // split path, fused trailing 1x1 conv ("pointwise", conv_f16s3_common.h): the workgroup holds every channel of its
    // output pixels (Cout <= BN), so the next layer's 1x1 convolution runs as a second small GEMM in the epilogue
    const _Float16* pw_wh = nullptr;            // [pw_cout..][pw_k] hi / lo planes of the 1x1 conv (its own packed weights)
    const _Float16* pw_wl = nullptr;
    const float* pw_inv_scale = nullptr;
    const float* pw_bias = nullptr;
    float* pw_out = nullptr;    int64_t pw_out_ldc = 0; int pw_out_coff = 0;
    int pw_npad = 0;                            // Npad of the hosted 1x1 conv's weight planes
    int pw_cout = 0, pw_k = 0, pw_leaky = 0;    // pw_k = Cout of this conv = Kpad of the 1x1 (32 or 64); pw_cout 16, 32 or 64
};

// Split activation format: value * SPLIT_SCALE stored as f16 hi + f16 lo planes per pixel.
constexpr float SPLIT_SCALE = 8.0f;
constexpr float ACT_SCALE_F16S3 = SPLIT_SCALE;
constexpr float F16_MAX = 65504.0f;

#ifdef __HIPCC__
// v (already x SPLIT_SCALE) -> hi + lo.  Saturates at the f16 range instead of producing inf (inf + -inf = NaN in the
// next layer); `amax` tracks max|v| for the overflow sentinel (split_overflow_report).
__device__ __forceinline__ void split_f16(float v, _Float16& h, _Float16& l, float& amax) {
    amax = fmaxf(amax, fabsf(v));
    const float vc = __builtin_amdgcn_fmed3f(v, -F16_MAX, F16_MAX);
    h = (_Float16)vc;
    l = (_Float16)(vc - (float)h);
}
// Activation of a conv epilogue: 0 linear, 1 leaky(0.1) (src/darknet.py:497-501), 2 SiLU x * sigmoid(x) (cfg extension).
// Exact-fp32 kernels and the stems; the split-f16 epilogues have their own form (conv_f16s3_common.h: silu_scaled behind a
// uniform branch around the loops of the LDS-transposed epilogue).
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : v * 0.1f;
    if (act == 2) return v / (1.0f + expf(-v));
    return v;
}
__device__ __forceinline__ void split_overflow_report(int32_t* flag, float amax) {
    if (flag && !(amax <= F16_MAX)) atomicOr(flag, 1);        // also true for NaN
}
#endif

enum ConvVariant { CV_128x128 = 0, CV_128x64 = 1, CV_64x64 = 2, CV_128x32 = 3, CV_COUNT };
struct ConvVariantInfo { int bm, bn; const char* name; };
const ConvVariantInfo& conv_variant_info(int v);
int conv_f32_kernel_name(int variant, char* buf, size_t len);
int launch_conv(const ConvArgs& a, int variant, hipStream_t s);
int conv_f32_slices(const ConvArgs& a);                                       // K slices of an exact-fp32 launch (1: unsliced)
enum ConvF16Variant { HV_128x128 = 0, HV_128x64 = 1, HV_64x64 = 2, HV_64x128 = 3, HV_256x128 = 4, HV_128x256 = 5, HV_128x128_8W = 6, HV_128x64_8W = 7, HV_256x128_16W = 8,
                      HV_192x128_8W = 9, HV_96x128_8W = 10, HV_192x128_12W = 11, HV_COUNT };
const ConvVariantInfo& conv_f16s3_variant_info(int v);
int conv_f16s3_kernel_name(int variant, int epi, char* buf, size_t len);      // demangled instantiation name (rocprofv3)
int launch_conv_f16s3(const ConvArgs& a, int variant, hipStream_t s);
// 3x3 stride-1 pad-1 convs with an LDS-resident input band (conv_band_f16s3.hip); weights in band K order
bool conv_band_supported(int ksize, int stride, int pad, int cin, int w_in);
// A layer the band kernel supports ALWAYS runs on it (its split-K layers sum in a different order than the generic kernel, and
// a frame's output must not depend on the batch it rides in); autotune only picks the tile.
constexpr int BANDD_MODES = 9;             // conv_bandd_f16s3.hip (round 4): weight fragments straight from global memory, band by LDS-DMA
constexpr int BAND_LDS_MODES = 11;         // conv_band_f16s3.hip: 128x128/4x2 waves, 128x64/4x2, 192x128/4x2, 192x128/6x2, 96x128/2x4, 128x128/2x2, 64x128/2x4,
                                           // and with in-workgroup split-K (two wave groups): 96x128/2x4, 128x128/4x2, 64x128/2x4, 128x64/4x2 (13x13 grids)
constexpr int BAND_MODES = BAND_LDS_MODES + BANDD_MODES;
static_assert(50 + BAND_MODES <= 70, "band variant ids end where the ring kernel's begin");    // modes >= BAND_LDS_MODES: conv_bandd tile (mode - BAND_LDS_MODES)
constexpr int BAND_K2_MODE0 = 7;
const ConvVariantInfo& conv_bandd_mode_info(int idx);
int conv_bandd_mode_kg(int idx);
// the last bandd tile (one band buffer of up to 29 blocks, two workgroups per CU) also runs 3x3 stride-1 layers with 94 < W <= 160: same K
// order and MFMA sequence as the generic tiles, so it is one more autotune candidate of those (non-band) layers
constexpr int BANDD_WIDE_MODE = BAND_LDS_MODES + 7;       // band-family mode index -> variant BAND_VARIANT_BASE + BANDD_WIDE_MODE
bool conv_bandd_wide_supported(int ksize, int stride, int pad, int cin, int w_in);
int conv_bandd_kernel_name(int idx, int epi, char* buf, size_t len);
int launch_conv_bandd_f16s3(const ConvArgs& a, int idx, hipStream_t s);
int conv_band_layer_kg(int cin, int h, int w);
bool conv_band_mode_valid(int mode, int cin, int h, int w);
int conv_band_default_mode(int cin, int h, int w);
const ConvVariantInfo& conv_band_mode_info(int mode);
int conv_band_kernel_name(int mode, int epi, char* buf, size_t len);
int launch_conv_band_f16s3(const ConvArgs& a, int mode, hipStream_t s);
constexpr int BAND_VARIANT_BASE = 50;      // variant ids in [50, 70) select the band kernel: BAND_VARIANT_BASE + mode
// Persistent LDS-DMA ring kernel (conv_ring_f16s3.hip): same K order and MFMA sequence as the generic implicit GEMM, so
// its tiles are further autotune candidates for every layer the generic kernel runs (bit-identical results).
constexpr int RING_MODES = 8;
constexpr int RING_VARIANT_BASE = 70;      // variant ids >= this: RING_VARIANT_BASE + mode
const ConvVariantInfo& conv_ring_mode_info(int mode);
int conv_ring_kernel_name(int mode, int epi, char* buf, size_t len);
int launch_conv_ring_f16s3(const ConvArgs& a, int mode, hipStream_t s);
// 1x1 convolutions with the A operand staged in 64-channel slabs (full-line LDS-DMA) and weight fragments straight from global
// memory (conv_pwd_f16s3.hip, round 4): same K order and MFMA sequence as the generic / ring tiles -> further autotune candidates
// of the plain 1x1 layers (no fused decode, no hosted pointwise conv), bit-identical results.
constexpr int PWD_MODES = 11;
constexpr int PWD_VARIANT_BASE = 90;       // variant ids in [90, 110): PWD_VARIANT_BASE + mode
bool conv_pwd_supported(int ksize, int stride, int pad, int cin);
const ConvVariantInfo& conv_pwd_mode_info(int mode);
int conv_pwd_kernel_name(int mode, int epi, char* buf, size_t len);
int launch_conv_pwd_f16s3(const ConvArgs& a, int mode, hipStream_t s);
// Stem + first stride-2 convolution (+ hosted 1x1) in one persistent kernel (conv_stem2_f16s3.hip): the stem's output never
// leaves the CU.  Bit-identical to the stand-alone kernels.
constexpr int STEM2_VARIANT = 130;         // variant id reported for the fused launch
bool conv_stem2_supported(int k0, int s0, int p0, int cin0, int cout0, int k1, int s1, int p1, int cout1, int pw_cout);
int conv_stem2_kernel_name(int pw, char* buf, size_t len);
int launch_conv_stem2_f16s3(const float* x, int B, int H, int W, const _Float16* w0h, const _Float16* w0l, const float* inv0, const float* bias0,
                            int leaky0, const ConvArgs& c1, hipStream_t s);
// 2-D tiled 3x3 stride-1 kernel with an LDS-resident input patch (conv_patch_f16s3.hip): for images too wide for the band
// kernel; bit-identical to the generic / ring tiles, so its modes are further autotune candidates of those layers.
constexpr int PATCH_MODES = 5;
constexpr int PATCH_WRES_MODE = 4;         // weights resident in LDS (Cin = 32, Cout = 64 only)
bool conv_patch_mode_valid(int mode, int cin, int cout);
constexpr int PATCH_VARIANT_BASE = 110;    // variant ids >= this: PATCH_VARIANT_BASE + mode
bool conv_patch_supported(int ksize, int stride, int pad, int cin, int cout);
const ConvVariantInfo& conv_patch_mode_info(int mode);
int conv_patch_kernel_name(int mode, int epi, char* buf, size_t len);
int launch_conv_patch_f16s3(const ConvArgs& a, int mode, hipStream_t s);

int launch_conv_stem(const float* x_nchw, const float* w, const float* bias, const View& out, int B, int H, int W,
                     int Ho, int Wo, int stride, int Cout, int leaky, hipStream_t s);
int launch_conv_stem_split(const float* x_nchw, const _Float16* wh, const _Float16* wl, const float* inv_scale, const float* bias,
                           const View& out, int B, int H, int W, int Ho, int Wo, int stride, int Cout, int leaky, int32_t* ovf, hipStream_t s);
int launch_prep_image(const unsigned char* img, int h, int w, int bgr, int inp_dim, float* out, hipStream_t s);
int launch_pack_input(const float* x_nchw, int B, int C, int H, int W, float* out_nhwc, int Cp, hipStream_t s);
int launch_upsample2x(const View& in, const View& out, int B, hipStream_t s);
int launch_add(const View& a, const View& b, const View& out, int B, hipStream_t s);
int launch_maxpool(const View& in, const View& out, int B, int size, int stride, int pad, hipStream_t s);   // pad > 0: symmetric -inf padding
int launch_upsample_nearest2x(const View& in, const View& out, int B, hipStream_t s);
// batch-statistics BatchNorm (+ activation + shortcut) over a conv's raw output, in place (aux_kernels.hip); stats: 2*C doubles of scratch
int launch_bn_batch(const View& x, const View& y, const View* res, int B, double* stats, int sstride, const float* bn, int gstride, int act,
                    double* partial, int64_t partial_doubles, hipStream_t s);    // partial: scratch of the two-stage statistics (bn_partial_doubles), or null
size_t bn_partial_doubles(int max_channels);
// running_mean / running_var update of every BatchNorm layer of a batch-statistics plan in one launch (aux_kernels.hip)
struct BnUpdateEntry { float* running_mean; float* running_var; int64_t stats_off; double unbias; int sstride; int channels; };
constexpr int BN_UPDATE_MAX = 32;                 // entries per launch (by-value kernel argument: 32 x 40 bytes)
struct BnUpdateTable { BnUpdateEntry e[BN_UPDATE_MAX]; };
int launch_bn_update_running(const BnUpdateEntry* entries, int n, const double* stats, double momentum, hipStream_t s);
int launch_copy(const View& in, const View& out, int B, hipStream_t s);
int launch_view_to_nchw(const View& in, int B, float* out_nchw, hipStream_t s);
// strided decode: raw element (b, ch, y, x) at raw[b*sb + ch*sc + y*sy + x*sx]
int launch_decode(const float* raw, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B,
                  const DecodeArgs& d, float* out, hipStream_t s);
int launch_confidence_mask(const float* pred, int64_t rows, int attrs, float conf, float* out, hipStream_t s);
int launch_bbox_iou(const float* box1, const float* boxes, int k, int row_stride, float* iou, hipStream_t s);

size_t nms_workspace_bytes(int batch, int n);
int launch_write_results(const float* pred, int batch, int n, int num_class, float conf, float nms,
                         float* out, int cap, int32_t* counts, void* ws, size_t ws_bytes, hipStream_t s);
int launch_nms_class_offset(const float* pred, int batch, int n, int num_class, float conf, float iou_thr, float max_wh, int max_det,
                            float* out, int cap, int32_t* counts, void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace rtod
