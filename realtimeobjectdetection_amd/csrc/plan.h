// Static execution plan of a Darknet cfg on one GPU: layer IR, NHWC buffer arena with liveness
// reuse, zero-copy route concat, fused shortcut / head-decode epilogues, packed weights, launch list.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "rtod_internal.h"

namespace rtod {

enum LayerType { LT_CONV = 0, LT_SHORTCUT, LT_ROUTE, LT_UPSAMPLE, LT_MAXPOOL, LT_YOLO };
const char* layer_type_name(int t);

struct Layer {
    int index = 0;
    int type = LT_CONV;
    int cin = 0, cout = 0, hin = 0, win = 0, hout = 0, wout = 0;
    int size = 0, stride = 1, pad = 0;
    bool bn = false, leaky = false;
    int act = 0;                  // 0 linear, 1 leaky(0.1), 2 SiLU (cfg extension); leaky == (act == 1)
    bool nearest = false;         // upsample: mode=nearest (cfg extension; the reference builds bilinear)
    bool decode_v5 = false;       // yolo: decode=v5 (cfg extension: YOLOv5-style head arithmetic)
    int pool_pad = 0;             // maxpool: symmetric=1 -> (size-1)/2 of -inf padding per side (cfg extension)
    std::vector<int> srcs;                       // absolute layer indices (route / shortcut)
    std::vector<std::pair<int, int>> anchors;    // yolo: masked (w,h) pairs
    int classes = 0, row_offset = 0, rows = 0;
    int64_t w_off = 0;                           // float offset of this conv's block in the .weights payload
    // planning results
    int fused_into = -1;      // conv: index of the shortcut/yolo layer whose output it writes
    bool fused_away = false;  // shortcut / yolo executed inside the preceding conv's epilogue
    int buf = -1, coff = 0;   // materialised output: arena buffer id + channel offset (-1: none/alias)
    int alias_of = -1;        // single-source route / yolo: same view as that layer
};

struct Buffer {
    int C = 0, H = 0, W = 0;      // C = ldc (full pixel width)
    int first = 0, last = 0;      // launch-time interval (layer indices) in which it is live
    int64_t floats_per_frame = 0;
    int64_t offset = 0;           // floats from arena base, for max_batch frames
};

enum LaunchKind { LK_CONV = 0, LK_PACK = 1, LK_UPSAMPLE = 2, LK_ADD = 3, LK_MAXPOOL = 4, LK_DECODE = 5, LK_COPY = 6, LK_STEM = 7 };

struct Launch {
    int kind = LK_CONV;
    int layer = 0;
    int in_layer = -1;        // layer whose output view is read (-1: network input buffer)
    int in2_layer = -1;       // residual / second operand
    int out_layer = -1;       // layer whose view is written (-2: final [B,N,attrs] output)
    int out_buf = -1, out_coff = 0;   // for LK_COPY into a concat slice
    int conv_slot = -1;       // index into Plan::convs
    DecodeArgs dec;
    // split-f16 plans: a 1x1 conv that directly follows a conv whose workgroups hold all output channels runs in that
    // conv's epilogue (conv_f16s3_common.h).  pw_guest: launch index of the 1x1 conv hosted by this launch; pw_host:
    // launch index of the host (this launch is then skipped).  Candidates are found at plan build (liveness covers
    // both schedules); Plan::pw_active() decides at run time.
    int pw_guest = -1, pw_host = -1;
};

struct PackedConv {
    int layer = 0;
    int cin_p = 0, K = 0, Kpad = 0, Npad = 0;
    int64_t w_off = 0, b_off = 0;     // float offsets into the device weight arena
    bool stem = false;                // conv_stem.hip: weights [28][Cout] fp32, or (split) [Cout][32] f16 hi / lo planes + inv_scale
    bool split = false;               // f16 hi/lo planes (conv_igemm_f16s3) instead of an fp32 panel
    bool band = false;                // eligible for conv_band_f16s3 (3x3 s1 p1, band fits LDS)
    int64_t wl_off = 0, s_off = 0;    // split: w_off = hi plane, wl_off = lo plane (float units), s_off = inv_scale
    int64_t stats_off = -1;           // batch-statistics BatchNorm plans: this layer's [Npad] mean, [Npad] variance in Plan::d_bn_stats (doubles)
    int64_t bn_off = 0;               // batch-statistics BatchNorm plans: [Npad] beta, [Npad] gamma (the conv itself is packed unfolded)
    int slice_chunks = 0;             // exact-fp32 panels: K summed in slices of this many 32-wide chunks (0: one chain), conv_igemm_f32.hip
};

struct Plan {
    int height = 0, width = 0, max_batch = 0, device = 0;
    std::map<std::string, std::string> net_info;
    std::vector<Layer> layers;
    std::vector<Buffer> bufs;
    int input_buf = -1;
    std::vector<Launch> launches;
    std::vector<PackedConv> convs;
    int total_rows = 0, attrs = 0;
    int64_t n_weight_floats = 0, conv_flops = 0;
    int64_t arena_floats = 0, packed_floats = 0;
    float* d_arena = nullptr;
    float* d_weights = nullptr;
    double* d_bn_stats = nullptr;     // batch-statistics BatchNorm: per-channel mean / biased variance of every BN layer's last forward
    int64_t bn_stats_doubles = 0;
    double* d_bn_partial = nullptr;   // per-range partial sums of the two-stage statistics kernel
    int64_t bn_partial_count = 0;
    float* d_scratch = nullptr;       // exact-fp32 plans: slice panels of the one-workgroup-per-K-slice schedule
    int64_t scratch_floats = 0;
    bool weights_loaded = false;
    int train_decode = 0;
    int precision = 0;         // 0 = exact fp32 MFMA, 1 = f16 hi/lo split (3 products)
    bool keep_all = false;     // debug: no arena reuse, every layer output stays readable after forward
    // plan options (rtod_plan_set_option, before rtod_plan_load_weights).  Explicit per-plan state: the library reads no
    // environment variable.
    bool opt_fuse_pointwise = true;   // run a 1x1 conv in the previous conv's epilogue where the plan allows it
    bool opt_stem_kernel = true;      // dedicated NCHW-reading kernel for layer 0 (else pack + generic conv)
    bool opt_band_kernel = true;      // LDS-band kernel for the 3x3 stride-1 layers it supports
    bool opt_bn_batch_stats = false;  // exact-fp32 plans: BatchNorm on the statistics of the batch (what the reference runs: no .eval()), not folded
    bool opt_k_slices = true;         // exact-fp32 kernels: deep small-grid layers summed in K slices (own workgroups when the grid is small)
    bool opt_k_slice_workgroups = true;   // ... (off: always the in-workgroup schedule — same bits; A/B and tests)
    bool opt_stem2_kernel = true;     // stem + layer 1 (+ hosted 1x1) in one kernel when the cfg starts like Darknet-53 (split-f16 plans)
    bool opt_patch_kernel = true;     // 2-D patch tiles among the autotune candidates of the wide 3x3 stride-1 layers
    bool opt_ring_kernel = true;      // persistent LDS-DMA ring tiles among the autotune candidates of the other layers
    bool opt_pwd_kernel = true;       // slab tiles (conv_pwd_f16s3.hip) among the autotune candidates of the plain 1x1 layers
    bool opt_fuse_shortcut = true;    // shortcut in the producing conv's epilogue (else stand-alone add kernel)
    bool opt_fuse_decode = true;      // head decode in the head conv's epilogue (else stand-alone decode kernel)
    bool opt_zero_copy_concat = true; // route producers write straight into the concat buffer (else copy kernels)
    int opt_force_f16s3_variant = -1; // >= 0: tile variant for every split-f16 conv (>= BAND_VARIANT_BASE: band layers)
    int opt_force_f32_variant = -1;   // >= 0: tile variant for every exact-fp32 conv
    int32_t* overflow_flag = nullptr; // caller-owned device word: split-f16 producers OR 1 into it when a value saturates
    std::vector<hipEvent_t> events;

    ~Plan();
    int parse(const std::string& cfg_text);
    int resolve_shapes();
    int plan_buffers();
    void assign_arena();
    void layout_weights();
    bool uses_split(const Layer& L, int cin_p) const;
    int check_split_supported() const;
    int load_weights(const float* w, size_t n);
    int forward(const float* x, int batch, float* out, hipStream_t s, float* launch_ms, bool tune = false);
    int set_option(const char* name, int value);
    int set_tiles(int batch, const int* variants, int count);   // install a tile table (validated per launch)
    void reset_planning();
    View view_of(int layer) const;            // resolves aliases; base == nullptr if not materialised
    int choose_variant(const Layer& L, int batch) const;
    int choose_variant_f16s3(const Layer& L, int batch) const;
    int build_conv_args(const Launch& l, int batch, float* out, ConvArgs& a) const;
    int tune_launch(size_t li, ConvArgs& a, int batch, hipStream_t s);
    std::vector<int> tuning;                    // scratch of the tuning forward
    std::map<std::vector<int>, int> tune_cache;
    int launch_split_variant(ConvArgs& a, const PackedConv& pc, int v, hipStream_t s) const;
    int variant_for(const Launch& l, int batch) const;
    int f32_slice_mode(const Launch& l, int batch, int variant) const;   // 0 plain, 1 slices inside the workgroup, 2 one workgroup per slice
    bool pw_active() const;
    bool pwd_candidate(const Launch& l, const Layer& L) const;
    bool bandd_wide_candidate(const Launch& l, const Layer& L) const;                     // fused pointwise convs in use (precision 1, option fuse_pointwise)
    bool stem2_pattern = false;                 // launches 0 / 1 are a stem and the stride-2 conv conv_stem2_f16s3 fuses (set by plan_buffers)
    bool stem2_active() const;                  // ... and the plan runs them fused (split-f16 precision, option stem2_kernel)
    std::map<int, std::vector<int>> tuned;     // batch -> per-launch split-f16 tile variant (-1: heuristic)
    std::string describe() const;
    void fill_launch_info(int idx, rtod_launch_info* o, int batch) const;
};

}  // namespace rtod
