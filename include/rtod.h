/*
 * rtod.h — C ABI of librtod.so: the MI355X (gfx950) YOLOv3 inference hot path.
 *
 * The reference (uguryagmur/RealTimeObjectDetection) is pure Python and has no FFI layer; its
 * boundary for this path is the Python API of src/darknet.py and src/util.py.  The host layer
 * in realtimeobjectdetection_amd/{darknet,util}.py keeps that API and binds the entry points
 * below through ctypes (INTEGRATION.md shows the stub).  Each entry point names the reference
 * interface it replaces (file:line into /root/reference).
 *
 * Conventions
 *  - every function returns 0 on success or a negative rtod_status; it never throws and never
 *    synchronises the device unless its comment says so;  rtod_last_error() returns the message
 *    of the calling thread's last failure;
 *  - the library reads no environment variable: every behaviour switch is explicit per-plan state
 *    (rtod_plan_set_precision, rtod_plan_set_option);
 *  - "dev" pointers are device (HBM) addresses owned by the caller (torch.Tensor.data_ptr());
 *    kernels are enqueued on the caller-supplied hipStream_t (void*; 0 = default stream);
 *  - a plan is not thread-safe; distinct plans are independent;
 *  - all tensors are float32.  Activations inside a plan are NHWC; the API edges keep the
 *    reference's layouts (input NCHW [B,3,H,W], predictions [B,N,5+C], detections [D,8]).
 */
#ifndef RTOD_H
#define RTOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtod_status {
    RTOD_OK = 0,
    RTOD_E_ARG = -1,     /* bad argument / shape */
    RTOD_E_HIP = -2,     /* HIP runtime error (message has hipGetErrorString) */
    RTOD_E_CFG = -3,     /* cfg grammar / unknown block (reference: src/darknet.py:524-526 asserts) */
    RTOD_E_STATE = -4,   /* call order (e.g. forward before load_weights) */
    RTOD_E_SIZE = -5     /* weight stream too short (reference: view_as raises, src/darknet.py:348) */
} rtod_status;

typedef struct rtod_plan rtod_plan;

typedef struct rtod_plan_info {
    int32_t n_layers;          /* cfg blocks after [net] (107 for yolov3, 24 for tiny) */
    int32_t n_launches;        /* kernels enqueued by one rtod_forward */
    int32_t height, width, max_batch;
    int32_t total_rows;        /* N of the [B,N,5+C] output (22743 @608) */
    int32_t attrs;             /* 5 + classes */
    int64_t n_weight_floats;   /* floats a .weights payload must hold */
    int64_t conv_flops_per_frame;   /* 2*MACs of the direct convolutions (SURVEY.md §8 d) */
    int64_t arena_bytes;       /* activation arena the plan allocates for max_batch */
    int64_t packed_weight_bytes;
} rtod_plan_info;

typedef struct rtod_launch_info {
    int32_t layer;             /* cfg layer index this launch implements (fused layers: the conv) */
    int32_t kind;              /* 0 conv-igemm, 1 input-pack, 2 upsample, 3 add, 4 maxpool, 5 decode, 6 copy, 7 stem conv */
    int32_t variant;           /* conv tile variant id (see rtod_conv_variant_name) */
    int32_t ksize, stride, cin, cout, hout, wout;
    int32_t fused_residual, fused_decode;
    int32_t fused_pointwise;   /* 1: the next layer's 1x1 conv runs in this conv's epilogue (its own launch entry is then empty) */
    int64_t flops_per_frame;   /* algorithmic 2*MACs (0 for non-conv launches) */
    int64_t bytes_per_frame;   /* algorithmic bytes: input once + output once (+ residual) */
    int64_t weight_bytes;      /* read once per launch */
} rtod_launch_info;

int rtod_version(void);
/* Copies the calling thread's last error message (NUL-terminated, truncated to len). */
int rtod_last_error(char* buf, size_t len);
/* Number of HIP devices visible; does not initialise a device context. */
int rtod_device_count(int* out);

/* ---- plan = Darknet(cfg) ------------------------------------------------------------------
 * replaces Darknet.__init__ -> parse_cfg + create_modules   src/darknet.py:176-189, 412-603
 * Parses the cfg text (same grammar), resolves shapes for [*,3,height,width], plans NHWC buffers
 * (zero-copy route concat, fused shortcut / head decode) and the launch list.  Host-only: no
 * device memory is touched until rtod_plan_load_weights. */
int rtod_plan_create(const char* cfg_text, size_t len, int height, int width, int max_batch,
                     int device, rtod_plan** out);
int rtod_plan_destroy(rtod_plan* plan);
int rtod_plan_get_info(const rtod_plan* plan, rtod_plan_info* out);
int rtod_plan_get_launch(const rtod_plan* plan, int index, rtod_launch_info* out);
/* JSON description of the resolved layer IR (tests compare it with the Python IR / reference). */
int rtod_plan_describe(const rtod_plan* plan, char* buf, size_t len, size_t* needed);
const char* rtod_conv_variant_name(int variant);
/* Demangled name of the kernel instantiation a launch runs (what rocprofv3 --kernel-trace prints): variant from
 * rtod_launch_info, epilogue 0 plain, 1 fused shortcut, 2 fused head decode, 3 / 4 = 0 / 1 with a fused pointwise conv. */
int rtod_conv_kernel_name(int variant, int epilogue, char* buf, size_t len);
/* Same for launch `index` of a plan (all context taken from the plan; "" for launches that are not convolution kernels). */
int rtod_plan_launch_kernel_name(const rtod_plan* plan, int index, char* buf, size_t len);
/* Arithmetic of the convolutions (call before rtod_plan_load_weights):
 *   0  exact fp32 MFMA (v_mfma_f32_32x32x2_f32): bit-level fmaf chains, the parity anchor;
 *   1  split-precision f16 MFMA: a*w ~= ah*wh + ah*wl + al*wh with fp32 accumulation (22-bit
 *      operands, error ~2x fp32 per layer), 16x the MFMA rate per product.  Activations live in
 *      HBM as f16 hi/lo planes (x8 pre-scaled): needs |activation| < 8188, every conv after the
 *      stem with Cin % 32 == 0, no stand-alone shortcut / copy / decode launch; otherwise RTOD_E_CFG.
 * cfg grammar: the reference's (src/darknet.py:412-603) plus three extension keys for YOLOv5-style blocks (detect.py:255-285
 * fetches that model from the network; only its building blocks exist here): [convolutional] activation=silu,
 * [maxpool] symmetric=1 (-inf padding of (size-1)/2 per side), [upsample] mode=nearest, [yolo] decode=v5, [route] with up to
 * four sources. */
int rtod_plan_set_precision(rtod_plan* plan, int mode);
/* Plan options (call before rtod_plan_load_weights; re-plans buffers and launches).  All default to 1 / -1:
 *   "fuse_pointwise"    1x1 conv in the previous conv's epilogue where one workgroup holds all its input channels
 *   "fuse_shortcut"     shortcut (src/darknet.py:263-268) in the producing conv's epilogue; 0: stand-alone add kernel
 *   "fuse_decode"       predict_transform (src/util.py:175-239) in the head conv's epilogue; 0: stand-alone decode kernel
 *   "zero_copy_concat"  route producers (src/darknet.py:270-290) write into the concat buffer; 0: copy kernels
 *   "stem_kernel"       dedicated NCHW-reading kernel for layer 0; 0: input pack + generic conv
 *   "band_kernel"       LDS-band kernel for 3x3 stride-1 layers; 0: generic implicit GEMM
 *   "ring_kernel"       persistent LDS-DMA ring tiles among the autotune candidates (bit-identical to the generic tiles)
 *   "patch_kernel"      2-D patch tiles among the autotune candidates of the wide 3x3 stride-1 layers (bit-identical)
 *   "stem2_kernel"      layers 0-2 of Darknet-53 in one kernel (conv_stem2_f16s3.hip, bit-identical); 0: stand-alone kernels
 *   "bn_batch_stats"    (default 0) exact-fp32 plans: BatchNorm on batch statistics instead of the folded running statistics
 *   "k_slices"          exact-fp32 plans: deep small-grid layers summed in K slices (conv_igemm_f32.hip); 0: one chain
 *   "k_slice_workgroups" ... one workgroup per slice when the grid is small; 0: always inside the workgroup (same bits)
 *   "force_f16s3_variant" / "force_f32_variant"   >= 0: one tile variant for every conv (tests, A/B runs)
 * Options that leave a cfg inexpressible in the split-f16 format return RTOD_E_CFG when precision is 1. */
int rtod_plan_set_option(rtod_plan* plan, const char* name, int value);
/* Split-f16 plans store activations as f16 hi/lo planes of 8*x: |activation| must stay below 8188.  Producers
 * saturate at that range (never inf / NaN) and OR 1 into *flag_dev (caller-owned device int32, zero it yourself)
 * whenever a value saturated: read it at any synchronisation point.  NULL disables the report. */
int rtod_plan_set_overflow_flag(rtod_plan* plan, int32_t* flag_dev);

/* replaces Darknet.load_weights                              src/darknet.py:316-410
 * `w` is the float payload of a Darknet .weights file (after the 5xint32 header), host memory:
 * per convolutional block [bn.bias, bn.weight, running_mean, running_var] or [conv.bias], then
 * conv.weight (OIHW).  Folds eval-mode BatchNorm (eps 1e-5) into the conv, packs K-major panels,
 * allocates device memory on first call and uploads.  Synchronises the device. */
int rtod_plan_load_weights(rtod_plan* plan, const float* w, size_t n_floats);

/* replaces Darknet.forward                                   src/darknet.py:199-303
 * x_dev: [batch,3,H,W] NCHW float32; out_dev: [batch,N,5+C] (new contiguous tensor in the
 * reference; here caller-allocated).  batch <= max_batch.  Enqueues only: no measurement, no host
 * synchronisation, no allocation -> legal under hipStreamBeginCapture (a captured forward replays as a hipGraph). */
int rtod_forward(rtod_plan* plan, const float* x_dev, int batch, float* out_dev, void* stream);
/* One forward that also measures, per distinct split-f16 conv shape, every tile variant on the layer's real input
 * (HIP events on `stream`) and remembers the fastest for this batch size; later rtod_forward calls of that batch
 * size use the table (untuned batch sizes use closed-form heuristics).  out_dev receives a valid forward result.
 * Synchronises `stream` repeatedly; not capturable.  No-op (plain forward) for exact-fp32 plans.  Tile choice never
 * changes results: every candidate of a layer sums its K products in the same order. */
int rtod_plan_autotune(rtod_plan* plan, const float* x_dev, int batch, float* out_dev, void* stream);
/* The tile table of a batch size: one variant id per launch (-1: heuristic / not a split-f16 conv).  get: copies the table an
 * autotune run left (returns the number of launches, or RTOD_E_STATE if that batch size was never tuned; `variants` may be NULL
 * to ask for the count).  set: installs a table (e.g. one saved by an earlier process) so that later forwards of that batch size
 * launch exactly those kernels without measuring anything — profiled passes (rocprofv3 --pmc) then replay the launches of the
 * timing pass.  Every entry is checked against its layer (band tiles on band layers, valid split-K mode, ...): RTOD_E_ARG. */
int rtod_plan_get_tiles(const rtod_plan* plan, int batch, int* variants, int capacity);
int rtod_plan_set_tiles(rtod_plan* plan, int batch, const int* variants, int count);
/* Same, with a hipEvent pair around every launch (recorded on `stream`); synchronises and
 * writes the per-launch durations in ms to launch_ms[n_launches] (host).  For bench/roofline. */
int rtod_forward_timed(rtod_plan* plan, const float* x_dev, int batch, float* out_dev,
                       void* stream, float* launch_ms);
/* replaces `with model.train_mode():`                       src/darknet.py:305-314
 * train != 0: heads apply only the sigmoids (TRAIN=True in predict_transform, util.py:211). */
int rtod_plan_set_train_decode(rtod_plan* plan, int train);
/* Debug/test: keep != 0 disables liveness-based arena reuse so that every layer's output is still
 * intact after a forward (for rtod_plan_read_layer).  Call before rtod_plan_load_weights. */
int rtod_plan_set_keep_all_layers(rtod_plan* plan, int keep);
/* Debug/test: copies layer `layer`'s output (NHWC view -> dense NCHW float32) to out_dev
 * [batch,C,H,W] after a forward; enqueues on stream. */
int rtod_plan_layer_shape(const rtod_plan* plan, int layer, int* c, int* h, int* w);
int rtod_plan_read_layer(rtod_plan* plan, int layer, int batch, float* out_dev_nchw, void* stream);
/* Plans with option "bn_batch_stats" = 1 (exact fp32 only) run BatchNorm the way the reference's callers do: never calling
 * .eval(), nn.BatchNorm2d normalises with the statistics of the batch (src/darknet.py:493-495, detect.py:185-194; SURVEY.md
 * F2).  Per-channel mean and BIASED variance of a conv layer's last forward (host doubles; synchronises `stream`): what the
 * host class needs to update running_mean / running_var like torch does (momentum 0.1, unbiased variance). */
int rtod_plan_bn_batch_stats(rtod_plan* plan, int layer, double* mean_host, double* var_host, int channels, void* stream);
/* The side effect itself, for ALL BatchNorm layers of such a plan in ONE launch (no host round trip): after a forward of `batch`
 * frames, running_mean[k] = (1 - momentum) * running_mean[k] + momentum * batch mean, running_var[k] likewise with the unbiased
 * batch variance, in float32 like torch (nn.BatchNorm2d in training mode, src/darknet.py:493-495).  running_mean_dev /
 * running_var_dev: HOST arrays of n_bn device pointers (float32 [cout]), the plan's BatchNorm layers in cfg order.  Enqueues only. */
int rtod_plan_bn_update_running(rtod_plan* plan, int batch, float* const* running_mean_dev, float* const* running_var_dev, int n_bn,
                                double momentum, void* stream);

/* replaces predict_transform                                 src/util.py:175-239
 * raw_dev [batch, A*attrs, G, G] NCHW -> out_dev [batch, G*G*A, attrs]; anchors = A (w,h) pairs
 * in input pixels (host); train != 0 applies only the three sigmoids. */
int rtod_predict_transform(const float* raw_dev, int batch, int attrs, int grid, int n_anchors,
                           const float* anchors_wh, int inp_dim, int train, float* out_dev,
                           void* stream);

/* replaces confidence_mask                                   src/util.py:106-117 */
int rtod_confidence_mask(const float* pred_dev, int64_t rows, int attrs, float confidence,
                         float* out_dev, void* stream);
/* replaces bbox_iou (one box vs k boxes, row stride in floats) src/util.py:120-153 */
int rtod_bbox_iou(const float* box1_dev, const float* boxes_dev, int k, int row_stride,
                  float* iou_dev, void* stream);

/* replaces prep_image + letterbox_image                      src/util.py:349-397
 * img_dev: uint8 [height,width,3] (HWC; bgr != 0: OpenCV channel order, swapped to RGB like prep_image's
 * default mode) -> out_dev float32 [3,inp_dim,inp_dim]: aspect-preserving bicubic resize, grey 128 padding,
 * /255.  cv2 is unavailable offline, so parity with cv2.INTER_CUBIC is unpinned (see preprocess.hip). */
int rtod_prep_image(const uint8_t* img_dev, int height, int width, int bgr, int inp_dim, float* out_dev, void* stream);

/* replaces write_results                                     src/util.py:242-346
 * pred_dev [batch,n,5+num_class].  Writes detections rows [img,x1,y1,x2,y2,obj,score,cls] to
 * out_dev[cap][8] in the reference's order (image asc, class asc, objectness desc) and
 * counts_dev[0] = D (may exceed cap: then only cap rows were written), counts_dev[1] = number of
 * candidate rows (obj > conf) over the batch, counts_dev[2..2+batch) = detections per image.
 * Workspace: rtod_write_results_workspace(batch, n) bytes of device memory.  Enqueues only. */
int rtod_write_results_workspace(int batch, int n, size_t* bytes);
int rtod_write_results(const float* pred_dev, int batch, int n, int num_class, float confidence,
                       float nms_conf, float* out_dev, int cap, int32_t* counts_dev,
                       void* workspace_dev, size_t workspace_bytes, void* stream);

/* (new; YOLOv5-style post-processing — the reference's YOLOv5 path is a torch.hub fetch, detect.py:255-285, so this follows the
 * published class-offset batched NMS, parity unpinned)  pred_dev [batch,n,5+num_class] rows (cx,cy,w,h,obj,cls...): candidates
 * obj > confidence and conf = obj * max class score > confidence; greedy NMS by descending conf on boxes shifted by
 * class * max_wh, suppression at IoU > iou_thr (no +1); rows [img,x1,y1,x2,y2,conf,obj,cls] per image by descending conf, at
 * most max_det per image; counts as rtod_write_results; same workspace size.  Enqueues only. */
int rtod_nms_class_offset(const float* pred_dev, int batch, int n, int num_class, float confidence, float iou_thr, float max_wh,
                          int max_det, float* out_dev, int cap, int32_t* counts_dev, void* workspace_dev, size_t workspace_bytes,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RTOD_H */
