// extern "C" surface of librtod.so (declared in include/rtod.h).  Nothing throws across it.
#include <cstring>
#include <new>

#include "plan.h"

namespace rtod { const std::string& last_error_string(); }
using namespace rtod;

struct rtod_plan { Plan p; };

#define RTOD_GUARD_BEGIN try {
#define RTOD_GUARD_END                                                                   \
    } catch (const std::bad_alloc&) { set_error("out of host memory"); return RTOD_E_ARG; } \
      catch (const std::exception& e) { set_error("internal error: %s", e.what()); return RTOD_E_ARG; } \
      catch (...) { set_error("internal error"); return RTOD_E_ARG; }

extern "C" {

int rtod_version(void) { return 100; }

int rtod_last_error(char* buf, size_t len) {
    if (!buf || !len) return RTOD_E_ARG;
    const std::string& e = last_error_string();
    const size_t n = e.size() < len - 1 ? e.size() : len - 1;
    memcpy(buf, e.data(), n);
    buf[n] = 0;
    return RTOD_OK;
}

int rtod_device_count(int* out) {
    if (!out) return RTOD_E_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out = 0; (void)hipGetLastError(); return RTOD_OK; }
    *out = n;
    return RTOD_OK;
}

int rtod_plan_create(const char* cfg_text, size_t len, int height, int width, int max_batch, int device, rtod_plan** out) {
    RTOD_GUARD_BEGIN
    if (!cfg_text || !out) { set_error("plan_create: null pointer"); return RTOD_E_ARG; }
    *out = nullptr;
    if (height < 1 || width < 1 || max_batch < 1 || device < 0) { set_error("plan_create: bad geometry %dx%d batch %d device %d", height, width, max_batch, device); return RTOD_E_ARG; }
    if (height != width) { set_error("plan_create: only square inputs are supported (the reference derives every stride from net_info['height'])"); return RTOD_E_ARG; }
    rtod_plan* h = new rtod_plan();
    h->p.height = height; h->p.width = width; h->p.max_batch = max_batch; h->p.device = device;
    int rc = h->p.parse(std::string(cfg_text, len));
    if (!rc) rc = h->p.resolve_shapes();
    if (!rc) rc = h->p.plan_buffers();
    if (rc) { delete h; return rc; }
    *out = h;
    return RTOD_OK;
    RTOD_GUARD_END
}

int rtod_plan_destroy(rtod_plan* plan) {
    delete plan;
    return RTOD_OK;
}

int rtod_plan_get_info(const rtod_plan* plan, rtod_plan_info* o) {
    if (!plan || !o) { set_error("plan_get_info: null pointer"); return RTOD_E_ARG; }
    const Plan& p = plan->p;
    memset(o, 0, sizeof(*o));
    o->n_layers = (int32_t)p.layers.size(); o->n_launches = (int32_t)p.launches.size();
    o->height = p.height; o->width = p.width; o->max_batch = p.max_batch;
    o->total_rows = p.total_rows; o->attrs = p.attrs;
    o->n_weight_floats = p.n_weight_floats; o->conv_flops_per_frame = p.conv_flops;
    o->arena_bytes = p.arena_floats * 4; o->packed_weight_bytes = p.packed_floats * 4;
    return RTOD_OK;
}

int rtod_plan_get_launch(const rtod_plan* plan, int index, rtod_launch_info* out) {
    if (!plan || !out || index < 0 || index >= (int)plan->p.launches.size()) { set_error("plan_get_launch: bad index"); return RTOD_E_ARG; }
    plan->p.fill_launch_info(index, out, plan->p.tuned.empty() ? plan->p.max_batch : plan->p.tuned.rbegin()->first);
    return RTOD_OK;
}

int rtod_plan_describe(const rtod_plan* plan, char* buf, size_t len, size_t* needed) {
    RTOD_GUARD_BEGIN
    if (!plan) { set_error("plan_describe: null plan"); return RTOD_E_ARG; }
    const std::string s = plan->p.describe();
    if (needed) *needed = s.size() + 1;
    if (buf && len) {
        const size_t n = s.size() < len - 1 ? s.size() : len - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return RTOD_OK;
    RTOD_GUARD_END
}

const char* rtod_conv_variant_name(int variant) {
    if (variant == 100 + STEM2_VARIANT) return "conv_stem2_f16s3<2 x 4x16, stem + 3x3 s2 + 1x1>";
    if (variant >= 100 + PATCH_VARIANT_BASE && variant < 100 + PATCH_VARIANT_BASE + PATCH_MODES) return conv_patch_mode_info(variant - 100 - PATCH_VARIANT_BASE).name;
    if (variant >= 100 + PWD_VARIANT_BASE && variant < 100 + PWD_VARIANT_BASE + PWD_MODES) return conv_pwd_mode_info(variant - 100 - PWD_VARIANT_BASE).name;
    if (variant >= 100 + RING_VARIANT_BASE && variant < 100 + RING_VARIANT_BASE + RING_MODES) return conv_ring_mode_info(variant - 100 - RING_VARIANT_BASE).name;
    if (variant >= 100 + BAND_VARIANT_BASE && variant < 100 + BAND_VARIANT_BASE + BAND_MODES) return conv_band_mode_info(variant - 100 - BAND_VARIANT_BASE).name;
    if (variant >= 100 && variant < 100 + HV_COUNT) return conv_f16s3_variant_info(variant - 100).name;
    if (variant < 0 || variant % 10 >= CV_COUNT || variant / 10 > 2) return "";
    static const char* const kSliced[2][CV_COUNT] = {
        {"conv_igemm_f32<128x128,w64x64,k-slices>", "conv_igemm_f32<128x64,w64x32,k-slices>", "conv_igemm_f32<64x64,w32x32,k-slices>", "conv_igemm_f32<128x32,w32x32,k-slices>"},
        {"conv_igemm_f32<128x128,w64x64,wg per k-slice>", "conv_igemm_f32<128x64,w64x32,wg per k-slice>", "conv_igemm_f32<64x64,w32x32,wg per k-slice>", "conv_igemm_f32<128x32,w32x32,wg per k-slice>"}};
    if (variant >= 10) return kSliced[variant / 10 - 1][variant % 10];
    return conv_variant_info(variant).name;
}

int rtod_conv_kernel_name(int variant, int epilogue, char* buf, size_t len) {
    if (!buf || len == 0) { set_error("conv_kernel_name: null buffer"); return RTOD_E_ARG; }
    int n = -1;
    if (variant >= 100 + PATCH_VARIANT_BASE) n = conv_patch_kernel_name(variant - 100 - PATCH_VARIANT_BASE, epilogue, buf, len);
    else if (variant >= 100 + PWD_VARIANT_BASE) n = conv_pwd_kernel_name(variant - 100 - PWD_VARIANT_BASE, epilogue, buf, len);
    else if (variant >= 100 + RING_VARIANT_BASE) n = conv_ring_kernel_name(variant - 100 - RING_VARIANT_BASE, epilogue, buf, len);
    else if (variant >= 100 + BAND_VARIANT_BASE) n = conv_band_kernel_name(variant - 100 - BAND_VARIANT_BASE, epilogue, buf, len);
    else if (variant >= 100) n = conv_f16s3_kernel_name(variant - 100, epilogue, buf, len);
    else n = conv_f32_kernel_name(variant, buf, len);
    if (n < 0 || (size_t)n >= len) { set_error("conv_kernel_name: unknown variant %d or buffer too small", variant); return RTOD_E_ARG; }
    return RTOD_OK;
}

int rtod_plan_launch_kernel_name(const rtod_plan* plan, int index, char* buf, size_t len) {
    RTOD_GUARD_BEGIN
    if (!plan || !buf || len == 0 || index < 0 || index >= (int)plan->p.launches.size()) { set_error("launch_kernel_name: bad args"); return RTOD_E_ARG; }
    rtod_launch_info li;
    plan->p.fill_launch_info(index, &li, plan->p.tuned.empty() ? plan->p.max_batch : plan->p.tuned.rbegin()->first);
    buf[0] = 0;
    if (li.kind != LK_CONV || li.flops_per_frame == 0) return RTOD_OK;                 // non-conv launch / conv hosted by the previous launch: empty name
    const int epi = li.fused_decode ? 2 : (li.fused_pointwise ? (li.fused_residual ? 4 : 3) : (li.fused_residual ? 1 : 0));
    int n = -1;
    if (li.variant == 100 + STEM2_VARIANT) n = conv_stem2_kernel_name(li.fused_pointwise, buf, len);
    else return rtod_conv_kernel_name(li.variant, epi, buf, len);
    if (n < 0 || (size_t)n >= len) { set_error("launch_kernel_name: buffer too small"); return RTOD_E_ARG; }
    return RTOD_OK;
    RTOD_GUARD_END
}

int rtod_plan_set_precision(rtod_plan* plan, int mode) {
    if (!plan || (mode != 0 && mode != 1)) { set_error("set_precision: mode must be 0 (fp32) or 1 (f16 split)"); return RTOD_E_ARG; }
    if (plan->p.d_weights) { set_error("set_precision: must be called before rtod_plan_load_weights"); return RTOD_E_STATE; }
    if (mode == 1) { const int rc = plan->p.check_split_supported(); if (rc) return rc; }
    plan->p.precision = mode;
    plan->p.layout_weights();
    return RTOD_OK;
}

int rtod_plan_set_option(rtod_plan* plan, const char* name, int value) {
    RTOD_GUARD_BEGIN
    if (!plan) { set_error("set_option: null plan"); return RTOD_E_ARG; }
    return plan->p.set_option(name, value);
    RTOD_GUARD_END
}

int rtod_plan_set_overflow_flag(rtod_plan* plan, int32_t* flag_dev) {
    if (!plan) { set_error("set_overflow_flag: null plan"); return RTOD_E_ARG; }
    plan->p.overflow_flag = flag_dev;
    return RTOD_OK;
}

int rtod_plan_load_weights(rtod_plan* plan, const float* w, size_t n_floats) {
    RTOD_GUARD_BEGIN
    if (!plan) { set_error("load_weights: null plan"); return RTOD_E_ARG; }
    return plan->p.load_weights(w, n_floats);
    RTOD_GUARD_END
}

int rtod_forward(rtod_plan* plan, const float* x_dev, int batch, float* out_dev, void* stream) {
    RTOD_GUARD_BEGIN
    if (!plan) { set_error("forward: null plan"); return RTOD_E_ARG; }
    return plan->p.forward(x_dev, batch, out_dev, (hipStream_t)stream, nullptr);
    RTOD_GUARD_END
}

int rtod_plan_autotune(rtod_plan* plan, const float* x_dev, int batch, float* out_dev, void* stream) {
    RTOD_GUARD_BEGIN
    if (!plan) { set_error("autotune: null plan"); return RTOD_E_ARG; }
    return plan->p.forward(x_dev, batch, out_dev, (hipStream_t)stream, nullptr, true);
    RTOD_GUARD_END
}

int rtod_plan_get_tiles(const rtod_plan* plan, int batch, int* variants, int capacity) {
    if (!plan) { set_error("get_tiles: null plan"); return RTOD_E_ARG; }
    auto it = plan->p.tuned.find(batch);
    if (it == plan->p.tuned.end()) { set_error("get_tiles: batch %d has no tile table (run rtod_plan_autotune first)", batch); return RTOD_E_STATE; }
    const int n = (int)it->second.size();
    if (variants) {
        if (capacity < n) { set_error("get_tiles: capacity %d < %d launches", capacity, n); return RTOD_E_ARG; }
        for (int i = 0; i < n; ++i) variants[i] = it->second[i];
    }
    return n;
}

int rtod_plan_set_tiles(rtod_plan* plan, int batch, const int* variants, int count) {
    RTOD_GUARD_BEGIN
    if (!plan) { set_error("set_tiles: null plan"); return RTOD_E_ARG; }
    return plan->p.set_tiles(batch, variants, count);
    RTOD_GUARD_END
}

int rtod_forward_timed(rtod_plan* plan, const float* x_dev, int batch, float* out_dev, void* stream, float* launch_ms) {
    RTOD_GUARD_BEGIN
    if (!plan || !launch_ms) { set_error("forward_timed: null pointer"); return RTOD_E_ARG; }
    return plan->p.forward(x_dev, batch, out_dev, (hipStream_t)stream, launch_ms);
    RTOD_GUARD_END
}

int rtod_plan_set_train_decode(rtod_plan* plan, int train) {
    if (!plan) { set_error("set_train_decode: null plan"); return RTOD_E_ARG; }
    plan->p.train_decode = train ? 1 : 0;
    return RTOD_OK;
}

int rtod_plan_set_keep_all_layers(rtod_plan* plan, int keep) {
    if (!plan) { set_error("set_keep_all_layers: null plan"); return RTOD_E_ARG; }
    if (plan->p.d_arena) { set_error("set_keep_all_layers: must be called before rtod_plan_load_weights"); return RTOD_E_STATE; }
    plan->p.keep_all = keep != 0;
    plan->p.assign_arena();
    return RTOD_OK;
}

int rtod_plan_layer_shape(const rtod_plan* plan, int layer, int* c, int* h, int* w) {
    if (!plan || layer < 0 || layer >= (int)plan->p.layers.size() || !c || !h || !w) { set_error("layer_shape: bad args"); return RTOD_E_ARG; }
    const View v = plan->p.view_of(layer);
    if (v.C == 0) { set_error("layer %d is fused into its consumer and has no materialised output", layer); return RTOD_E_STATE; }
    *c = v.C; *h = v.H; *w = v.W;
    return RTOD_OK;
}

int rtod_plan_read_layer(rtod_plan* plan, int layer, int batch, float* out_dev_nchw, void* stream) {
    RTOD_GUARD_BEGIN
    if (!plan || layer < 0 || layer >= (int)plan->p.layers.size()) { set_error("read_layer: bad args"); return RTOD_E_ARG; }
    if (batch < 1 || batch > plan->p.max_batch) { set_error("read_layer: bad batch"); return RTOD_E_ARG; }
    const View v = plan->p.view_of(layer);
    if (!v.base) { set_error("layer %d has no materialised output (fused) or weights not loaded", layer); return RTOD_E_STATE; }
    return launch_view_to_nchw(v, batch, out_dev_nchw, (hipStream_t)stream);
    RTOD_GUARD_END
}

int rtod_plan_bn_batch_stats(rtod_plan* plan, int layer, double* mean_host, double* var_host, int channels, void* stream) {
    RTOD_GUARD_BEGIN
    if (!plan || !mean_host || !var_host || layer < 0 || layer >= (int)plan->p.layers.size()) { set_error("bn_batch_stats: bad args"); return RTOD_E_ARG; }
    if (!plan->p.opt_bn_batch_stats || !plan->p.d_bn_stats) { set_error("bn_batch_stats: the plan does not run batch-statistics BatchNorm (option bn_batch_stats, weights loaded)"); return RTOD_E_STATE; }
    for (const auto& pc : plan->p.convs) {
        if (pc.layer != layer) continue;
        if (pc.stats_off < 0 || channels != plan->p.layers[layer].cout) { set_error("bn_batch_stats: layer %d has no BatchNorm / %d channels", layer, plan->p.layers[layer].cout); return RTOD_E_ARG; }
        RTOD_HIP(hipSetDevice(plan->p.device));
        RTOD_HIP(hipStreamSynchronize((hipStream_t)stream));
        RTOD_HIP(hipMemcpy(mean_host, plan->p.d_bn_stats + pc.stats_off, sizeof(double) * channels, hipMemcpyDeviceToHost));
        RTOD_HIP(hipMemcpy(var_host, plan->p.d_bn_stats + pc.stats_off + pc.Npad, sizeof(double) * channels, hipMemcpyDeviceToHost));
        return RTOD_OK;
    }
    set_error("bn_batch_stats: layer %d is not a convolution", layer);
    return RTOD_E_ARG;
    RTOD_GUARD_END
}

int rtod_plan_bn_update_running(rtod_plan* plan, int batch, float* const* running_mean_dev, float* const* running_var_dev, int n_bn,
                                double momentum, void* stream) {
    RTOD_GUARD_BEGIN
    if (!plan || !running_mean_dev || !running_var_dev || batch < 1 || batch > plan->p.max_batch) { set_error("bn_update_running: bad args"); return RTOD_E_ARG; }
    if (!plan->p.opt_bn_batch_stats || !plan->p.d_bn_stats) { set_error("bn_update_running: the plan does not run batch-statistics BatchNorm (option bn_batch_stats, weights loaded)"); return RTOD_E_STATE; }
    std::vector<BnUpdateEntry> ent;
    for (const auto& pc : plan->p.convs) {                       // cfg order
        if (pc.stats_off < 0) continue;
        const int k = (int)ent.size();
        if (k >= n_bn) { set_error("bn_update_running: %d pointers for more BatchNorm layers", n_bn); return RTOD_E_ARG; }
        if (!running_mean_dev[k] || !running_var_dev[k]) { set_error("bn_update_running: null buffer %d", k); return RTOD_E_ARG; }
        const View v = plan->p.view_of(pc.layer);
        const auto& L = plan->p.layers[pc.layer];
        const int64_t n = (int64_t)batch * L.hout * L.wout;
        (void)v;
        ent.push_back(BnUpdateEntry{running_mean_dev[k], running_var_dev[k], pc.stats_off, (double)n / (double)(n > 1 ? n - 1 : 1), pc.Npad, L.cout});
    }
    if ((int)ent.size() != n_bn) { set_error("bn_update_running: the plan has %d BatchNorm layers, %d pointers given", (int)ent.size(), n_bn); return RTOD_E_ARG; }
    if (ent.empty()) return RTOD_OK;
    RTOD_HIP(hipSetDevice(plan->p.device));
    return launch_bn_update_running(ent.data(), (int)ent.size(), plan->p.d_bn_stats, momentum, (hipStream_t)stream);
    RTOD_GUARD_END
}

int rtod_predict_transform(const float* raw_dev, int batch, int attrs, int grid, int n_anchors, const float* anchors_wh,
                           int inp_dim, int train, float* out_dev, void* stream) {
    RTOD_GUARD_BEGIN
    if (!raw_dev || !out_dev || !anchors_wh) { set_error("predict_transform: null pointer"); return RTOD_E_ARG; }
    if (batch < 1 || attrs < 5 || grid < 1 || n_anchors < 1 || n_anchors > 4 || inp_dim < grid) { set_error("predict_transform: bad shape"); return RTOD_E_ARG; }
    const int stride = inp_dim / grid;                 // util.py:194
    if (inp_dim / stride != grid) { set_error("predict_transform: grid %d inconsistent with inp_dim %d (util.py:195 would mis-shape)", grid, inp_dim); return RTOD_E_ARG; }
    DecodeArgs d; d.enabled = 1; d.G = grid; d.attrs = attrs; d.n_anchors = n_anchors; d.train = train ? 1 : 0; d.stride = (float)stride;
    for (int a = 0; a < n_anchors; ++a) {
        d.aw[a] = (float)((double)anchors_wh[2 * a] / (double)stride);
        d.ah[a] = (float)((double)anchors_wh[2 * a + 1] / (double)stride);
    }
    d.img_stride = (int64_t)grid * grid * n_anchors * attrs; d.head_off = 0;
    // NCHW raw tensor: (b, ch, y, x) strides
    return launch_decode(raw_dev, (int64_t)n_anchors * attrs * grid * grid, (int64_t)grid * grid, grid, 1, batch, d, out_dev, (hipStream_t)stream);
    RTOD_GUARD_END
}

int rtod_confidence_mask(const float* pred_dev, int64_t rows, int attrs, float confidence, float* out_dev, void* stream) {
    return launch_confidence_mask(pred_dev, rows, attrs, confidence, out_dev, (hipStream_t)stream);
}

int rtod_bbox_iou(const float* box1_dev, const float* boxes_dev, int k, int row_stride, float* iou_dev, void* stream) {
    return launch_bbox_iou(box1_dev, boxes_dev, k, row_stride, iou_dev, (hipStream_t)stream);
}

int rtod_prep_image(const uint8_t* img_dev, int height, int width, int bgr, int inp_dim, float* out_dev, void* stream) {
    RTOD_GUARD_BEGIN
    return launch_prep_image(img_dev, height, width, bgr, inp_dim, out_dev, (hipStream_t)stream);
    RTOD_GUARD_END
}

int rtod_write_results_workspace(int batch, int n, size_t* bytes) {
    if (!bytes || batch < 1 || n < 1) { set_error("write_results_workspace: bad args"); return RTOD_E_ARG; }
    *bytes = nms_workspace_bytes(batch, n);
    return RTOD_OK;
}

int rtod_write_results(const float* pred_dev, int batch, int n, int num_class, float confidence, float nms_conf,
                       float* out_dev, int cap, int32_t* counts_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
    RTOD_GUARD_BEGIN
    return launch_write_results(pred_dev, batch, n, num_class, confidence, nms_conf, out_dev, cap, counts_dev,
                                workspace_dev, workspace_bytes, (hipStream_t)stream);
    RTOD_GUARD_END
}

int rtod_nms_class_offset(const float* pred_dev, int batch, int n, int num_class, float confidence, float iou_thr, float max_wh, int max_det,
                          float* out_dev, int cap, int32_t* counts_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
    RTOD_GUARD_BEGIN
    return launch_nms_class_offset(pred_dev, batch, n, num_class, confidence, iou_thr, max_wh, max_det, out_dev, cap, counts_dev,
                                   workspace_dev, workspace_bytes, (hipStream_t)stream);
    RTOD_GUARD_END
}

}  // extern "C"
