// write_results on the GPU: confidence filter + class arg-max + per-(image,class) greedy NMS.
//
// Replaces write_results / bbox_iou (reference: src/util.py:242-346, 120-153).
// Selection and order are integer work and must be bit-exact with the reference on identical
// inputs, so this file is compiled with -ffp-contract=off (no FMA contraction: the reference rounds
// after every add/mul) and relies on hipcc's correctly-rounded fp32 division.
//
//   K1 filter   one wave per 64 rows: lanes test obj > conf (strict), the wave then scans the
//               classes of each passing row cooperatively (coalesced 4*(5+C)-byte row read) and
//               reduces to the FIRST arg-max; the box is converted to corners and the candidate
//               gets a 64-bit sort key (class asc | objectness desc | row asc).
//   K2 nms      one workgroup per image: bitonic sort of the keys (LDS when they fit), segment
//               boundaries by class, one wave per segment runs the greedy suppression with the
//               reference's +1-pixel IoU and strict `iou < thr` keep rule, then a block-wide scan
//               compacts survivors in sorted order.
//   K3 emit     image-major concatenation into out[D][8] = [img,x1,y1,x2,y2,obj,score,cls].
//
// Ties in objectness (undefined in the reference, torch.sort is unstable) resolve to the lower row.
#include "rtod_internal.h"

namespace rtod {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NMS_LDS_CAP = 8192;      // keys that fit the LDS sort (64 KiB)
constexpr int NMS_MAX_CLASSES = 4096;  // 12 key bits
constexpr int NMS_ROW_BITS = 20;       // rows per image < 2^20
constexpr int NMS_BLOCK = 1024;

struct NmsWs {
    int32_t* cand;        // [B]  sortable candidates per image (score != 0)
    int32_t* cand_all;    // [B]  rows with obj > conf per image (one word per image: a single word serialises ~2000 atomics)
    int32_t* ndet;        // [B]  survivors per image
    uint64_t* keys;       // [B][P]  P = pow2 >= n
    uint8_t* alive;       // [B][P]
    int32_t* segs;        // [B][NMS_MAX_CLASSES + 1]
    float* rec;           // [B][n][8]  x1,y1,x2,y2,obj,score,cls,-
    int P;
};

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t nms_workspace_bytes(int batch, int n) {
    const size_t P = next_pow2(n < 1 ? 1 : n);
    size_t b = 0;
    b += align256(sizeof(int32_t) * (3 * (size_t)batch + 4));
    b += align256(sizeof(uint64_t) * batch * P);
    b += align256(batch * P);
    b += align256(sizeof(int32_t) * (size_t)batch * (NMS_MAX_CLASSES + 1));
    b += align256(sizeof(float) * (size_t)batch * n * 8);
    return b;
}

static NmsWs carve(void* ws, int batch, int n) {
    NmsWs w;
    const size_t P = next_pow2(n < 1 ? 1 : n);
    char* p = (char*)ws;
    w.cand = (int32_t*)p;
    w.cand_all = w.cand + batch;
    w.ndet = w.cand + 2 * batch + 4;
    p += align256(sizeof(int32_t) * (3 * (size_t)batch + 4));
    w.keys = (uint64_t*)p; p += align256(sizeof(uint64_t) * batch * P);
    w.alive = (uint8_t*)p; p += align256(batch * P);
    w.segs = (int32_t*)p; p += align256(sizeof(int32_t) * (size_t)batch * (NMS_MAX_CLASSES + 1));
    w.rec = (float*)p;
    w.P = (int)P;
    return w;
}

__device__ __forceinline__ uint32_t float_desc_key(float v) {
    uint32_t u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;   // ascending-sortable
    return ~u;                                     // descending
}

// ------------------------------------------------------------------------------------------ K1
// V5 = true: the YOLOv5-style variant (class-offset batched NMS; not reference behaviour, see launch_nms_class_offset):
// candidates need obj > conf AND obj * best class score > conf, the sort key carries that product.
template <bool V5>
__global__ __launch_bounds__(256)
void nms_filter_kernel(const float* __restrict__ pred, int B, int n, int num_class, float conf, NmsWs w) {
    const int attrs = 5 + num_class;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int waves_per_img = (n + 63) / 64;
    const int gw = blockIdx.x * 4 + wave_in_block;
    if (gw >= B * waves_per_img) return;
    const int b = gw / waves_per_img;
    const int row = (gw - b * waves_per_img) * 64 + lane;
    const float* img = pred + (int64_t)b * n * attrs;

    bool pass = false;
    if (row < n) {
        const float obj = img[(int64_t)row * attrs + 4];
        pass = (obj > conf) && (obj != 0.0f);      // mask (strict >), then nonzero(obj)  util.py:116,286
    }
    unsigned long long mask = __ballot(pass);
    if (lane == 0 && mask) atomicAdd(&w.cand_all[b], __popcll(mask));
    while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int r = (gw - b * waves_per_img) * 64 + src;
        const float* p = img + (int64_t)r * attrs;
        // first arg-max over classes (torch.max(dim) on CPU returns the first maximal index)
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int c = lane; c < num_class; c += 64) {
            const float v = p[5 + c];
            if (v > best) { best = v; bi = c; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (bi == 0x7fffffff) { bi = 0; best = p[5]; }            // no finite maximum (NaN / -inf scores)
        if (lane == 0) {
            const float cx = p[0], cy = p[1], bw = p[2], bh = p[3], obj = p[4];
            const float hw = bw / 2.0f, hh = bh / 2.0f;
            float* rec = w.rec + ((int64_t)b * n + r) * 8;
            f32x4 lo = {cx - hw, cy - hh, cx + hw, cy + hh};
            const float score = V5 ? obj * best : obj;             // V5: conf = obj * cls (one rounding, as x[:, 5:] *= x[:, 4:5])
            f32x4 hi = V5 ? f32x4{score, obj, (float)bi, 0.f} : f32x4{obj, best, (float)bi, 0.f};
            *reinterpret_cast<f32x4*>(rec) = lo;
            *reinterpret_cast<f32x4*>(rec + 4) = hi;
            if (V5 ? (score > conf) : (best != 0.0f)) {    // reference: class rows with score == 0 are dropped  util.py:305
                const int slot = atomicAdd(&w.cand[b], 1);
                const uint64_t key = ((uint64_t)bi << (32 + NMS_ROW_BITS)) |
                                     ((uint64_t)float_desc_key(score) << NMS_ROW_BITS) | (uint64_t)r;
                if (slot < w.P) w.keys[(int64_t)b * w.P + slot] = key;   // (always true with zeroed counters; never write past the image's slots)
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ K2
__device__ __forceinline__ float iou_ref(const f32x4 a, const f32x4 b) {
    // bbox_iou, src/util.py:138-151 — same operation order, fp32, one rounding per op
    const float ix1 = fmaxf(a[0], b[0]), iy1 = fmaxf(a[1], b[1]);
    const float ix2 = fminf(a[2], b[2]), iy2 = fminf(a[3], b[3]);
    const float iw = fmaxf((ix2 - ix1) + 1.0f, 0.0f);
    const float ih = fmaxf((iy2 - iy1) + 1.0f, 0.0f);
    const float inter = iw * ih;
    const float a1 = ((a[2] - a[0]) + 1.0f) * ((a[3] - a[1]) + 1.0f);
    const float a2 = ((b[2] - b[0]) + 1.0f) * ((b[3] - b[1]) + 1.0f);
    return inter / ((a1 + a2) - inter);
}

// IoU of the class-offset batched NMS: boxes shifted by class * max_wh before the test (so that one NMS pass serves all
// classes; the shift costs coordinate bits in fp32 and is therefore restated, not optimised away), no +1, no contraction.
__device__ __forceinline__ float iou_offset(const f32x4 a, const f32x4 b, float shift) {
    const float ax1 = a[0] + shift, ay1 = a[1] + shift, ax2 = a[2] + shift, ay2 = a[3] + shift;
    const float bx1 = b[0] + shift, by1 = b[1] + shift, bx2 = b[2] + shift, by2 = b[3] + shift;
    const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.0f);
    const float ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.0f);
    const float inter = iw * ih;
    const float a1 = (ax2 - ax1) * (ay2 - ay1), a2 = (bx2 - bx1) * (by2 - by1);
    return inter / ((a1 + a2) - inter);
}

template <bool V5>
__global__ __launch_bounds__(NMS_BLOCK)
void nms_sort_suppress_kernel(int n, float nms_thr, NmsWs w, float max_wh, int max_det) {
    __shared__ uint64_t s_keys[NMS_LDS_CAP];
    __shared__ uint8_t s_alive[NMS_LDS_CAP];
    __shared__ int s_scan[NMS_BLOCK / 64 + 1];
    __shared__ int s_nseg;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int cnt = min(w.cand[b], w.P);
    uint64_t* gk = w.keys + (int64_t)b * w.P;
    int* segs = w.segs + (int64_t)b * (NMS_MAX_CLASSES + 1);
    if (cnt == 0) { if (tid == 0) w.ndet[b] = 0; return; }

    int P = 1; while (P < cnt) P <<= 1;
    const bool in_lds = P <= NMS_LDS_CAP;
    uint64_t* keys = in_lds ? s_keys : gk;
    // LDS operations of one wave execute in order, so a lane's flag store is seen by the wave's next
    // read; on the (rare, > 8192 candidates) global path the workgroup fence below drains the stores.
    volatile uint8_t* alive = in_lds ? s_alive : (w.alive + (int64_t)b * w.P);
    for (int i = tid; i < P; i += NMS_BLOCK) {
        const uint64_t k = i < cnt ? gk[i] : ~0ull;
        keys[i] = k;
    }
    if (tid == 0) s_nseg = 0;
    __syncthreads();
    // bitonic sort, ascending
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += NMS_BLOCK) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const uint64_t a = keys[i], c = keys[l];
                const bool up = (i & k) == 0;
                if ((a > c) == up) { keys[i] = c; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    // segment starts (class changes); order of the list is irrelevant, segments are independent
    for (int i = tid; i < cnt; i += NMS_BLOCK) {
        alive[i] = 1;
        const uint32_t c = (uint32_t)(keys[i] >> (32 + NMS_ROW_BITS));
        if (i == 0 || c != (uint32_t)(keys[i - 1] >> (32 + NMS_ROW_BITS))) segs[atomicAdd(&s_nseg, 1)] = i;
    }
    __syncthreads();
    const int nseg = s_nseg;
    const int lane = tid & 63, wave = tid >> 6;
    const float* rec = w.rec + (int64_t)b * n * 8;
    const uint64_t rowmask = (1ull << NMS_ROW_BITS) - 1;
    for (int sgi = wave; sgi < nseg; sgi += NMS_BLOCK / 64) {
        const int s0 = segs[sgi];
        const uint32_t cls = (uint32_t)(keys[s0] >> (32 + NMS_ROW_BITS));
        int s1 = s0 + 1;   // find the end of the segment (wave-uniform scan in 64-wide steps)
        for (;;) {
            const int i = s1 + lane;
            const bool same = i < cnt && (uint32_t)(keys[i] >> (32 + NMS_ROW_BITS)) == cls;
            const unsigned long long m = __ballot(same);
            if (m == ~0ull) { s1 += 64; continue; }
            s1 += __ffsll((long long)~m) - 1;
            break;
        }
        for (int i = s0; i < s1; ++i) {
            if (!alive[i]) continue;                               // wave-uniform
            const f32x4 bi = *reinterpret_cast<const f32x4*>(rec + (keys[i] & rowmask) * 8);
            for (int j = i + 1 + lane; j < s1; j += 64) {
                if (!alive[j]) continue;
                const f32x4 bj = *reinterpret_cast<const f32x4*>(rec + (keys[j] & rowmask) * 8);
                if constexpr (V5) {
                    if (iou_offset(bi, bj, (float)cls * max_wh) > nms_thr) alive[j] = 0;   // suppressed iff iou > thr
                } else {
                    const float iou = iou_ref(bi, bj);
                    if (!(iou < nms_thr)) alive[j] = 0;            // keep iff iou < thr (strict)
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // compact survivors in sorted order: block-wide exclusive scan over chunks of NMS_BLOCK
    int base = 0;
    for (int c0 = 0; c0 < cnt; c0 += NMS_BLOCK) {
        const int i = c0 + tid;
        const bool a = i < cnt && alive[i];
        const unsigned long long m = __ballot(a);
        const int wprefix = __popcll(m & ((1ull << lane) - 1));
        if (lane == 0) s_scan[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int q = 0; q < NMS_BLOCK / 64; ++q) { const int v = s_scan[q]; if (q < wave) woff += v; tot += v; }
        const uint64_t k = a ? keys[i] : 0;
        __syncthreads();                       // all reads of keys[i] / s_scan done before overwrite
        if (a) gk[base + woff + wprefix] = k;  // survivors' keys, compacted (gk is free: sorted copy is in `keys`
        base += tot;                           //   or, on the global path, positions < i were already read)
        __syncthreads();
    }
    if constexpr (V5) {
        // survivors are in (class, score) order; the batched NMS returns them by descending score over all classes, capped
        // at max_det: re-key by (score desc, row asc) and sort once more
        __syncthreads();
        int Q = 1; while (Q < base) Q <<= 1;
        uint64_t* k2 = (Q <= NMS_LDS_CAP) ? s_keys : gk;
        for (int i = tid; i < Q; i += NMS_BLOCK) {                // (global path: in place — a thread rewrites only the slot it read)
            uint64_t k = ~0ull;
            if (i < base) {
                const uint64_t row = gk[i] & rowmask;
                k = ((uint64_t)float_desc_key(rec[row * 8 + 4]) << NMS_ROW_BITS) | row;
            }
            k2[i] = k;
        }
        __syncthreads();
        for (int k = 2; k <= Q; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (Q >> 1); t += NMS_BLOCK) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int l = i | j;
                    const uint64_t a = k2[i], c = k2[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { k2[i] = c; k2[l] = a; }
                }
                __syncthreads();
            }
        }
        if (k2 != gk) for (int i = tid; i < base; i += NMS_BLOCK) gk[i] = k2[i];
        if (tid == 0) w.ndet[b] = base < max_det ? base : max_det;
    } else {
        if (tid == 0) w.ndet[b] = base;
    }
}

__global__ void nms_zero_kernel(int32_t* __restrict__ p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}

// ------------------------------------------------------------------------------------------ K3
__global__ __launch_bounds__(256)
void nms_emit_kernel(int B, int n, NmsWs w, float* __restrict__ out, int cap, int32_t* __restrict__ counts) {
    const int b = blockIdx.x;
    int off = 0, total = 0;
    for (int i = 0; i < B; ++i) { const int v = w.ndet[i]; if (i < b) off += v; total += v; }
    const int nd = w.ndet[b];
    if (threadIdx.x == 0) {
        counts[2 + b] = nd;
        if (b == 0) { int ca = 0; for (int i = 0; i < B; ++i) ca += w.cand_all[i]; counts[0] = total; counts[1] = ca; }
    }
    const uint64_t* gk = w.keys + (int64_t)b * w.P;
    const float* rec = w.rec + (int64_t)b * n * 8;
    const uint64_t rowmask = (1ull << NMS_ROW_BITS) - 1;
    for (int j = threadIdx.x; j < nd; j += blockDim.x) {
        const int o = off + j;
        if (o >= cap) break;
        const float* r = rec + (gk[j] & rowmask) * 8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(r);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(r + 4);
        f32x4 o0 = {(float)b, lo[0], lo[1], lo[2]};
        f32x4 o1 = {lo[3], hi[0], hi[1], hi[2]};
        *reinterpret_cast<f32x4*>(out + (int64_t)o * 8) = o0;
        *reinterpret_cast<f32x4*>(out + (int64_t)o * 8 + 4) = o1;
    }
}

int launch_write_results(const float* pred, int batch, int n, int num_class, float conf, float nms,
                         float* out, int cap, int32_t* counts, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!pred || !out || !counts || !ws) { set_error("write_results: null pointer"); return RTOD_E_ARG; }
    if (batch < 1 || n < 1 || num_class < 1 || num_class > NMS_MAX_CLASSES || n >= (1 << NMS_ROW_BITS) || cap < 0) {
        set_error("write_results: unsupported shape (batch=%d n=%d classes=%d)", batch, n, num_class); return RTOD_E_ARG;
    }
    if (ws_bytes < nms_workspace_bytes(batch, n)) { set_error("write_results: workspace too small"); return RTOD_E_ARG; }
    if (((uintptr_t)ws & 15) || ((uintptr_t)out & 15)) { set_error("write_results: workspace/out must be 16-byte aligned"); return RTOD_E_ARG; }
    NmsWs w = carve(ws, batch, n);
    // counters zeroed by a kernel, not hipMemsetAsync: captured into a HIP graph the 28-byte memset node left the counters
    // of the previous replay in place on ROCm 7.2 (second replay: candidates appended past the buffers, GPU fault)
    hipLaunchKernelGGL(nms_zero_kernel, dim3(1), dim3(64), 0, s, w.cand, 3 * batch + 4);
    const int waves = batch * ((n + 63) / 64);
    hipLaunchKernelGGL(nms_filter_kernel<false>, dim3((waves + 3) / 4), dim3(256), 0, s, pred, batch, n, num_class, conf, w);
    hipLaunchKernelGGL(nms_sort_suppress_kernel<false>, dim3(batch), dim3(NMS_BLOCK), 0, s, n, nms, w, 0.f, 0);
    hipLaunchKernelGGL(nms_emit_kernel, dim3(batch), dim3(256), 0, s, batch, n, w, out, cap, counts);
    return hip_fail(hipGetLastError(), "write_results launch");
}

// Class-offset batched NMS (YOLOv5-style post-processing; cfg-extension companion, NOT reference behaviour: the reference's
// YOLOv5 path is a torch.hub fetch, detect.py:255-285).  pred [batch,n,5+C] rows (cx,cy,w,h,obj,cls...): keep obj > conf, conf =
// obj * max class score > conf; greedy NMS by descending conf on boxes shifted by class * max_wh, suppression at IoU > thr
// (no +1); rows out [img,x1,y1,x2,y2,conf,obj,cls], per image by descending conf (ties: lower row first), at most max_det.
int launch_nms_class_offset(const float* pred, int batch, int n, int num_class, float conf, float iou_thr, float max_wh, int max_det,
                            float* out, int cap, int32_t* counts, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!pred || !out || !counts || !ws) { set_error("nms_class_offset: null pointer"); return RTOD_E_ARG; }
    if (batch < 1 || n < 1 || num_class < 1 || num_class > NMS_MAX_CLASSES || n >= (1 << NMS_ROW_BITS) || cap < 0 || max_det < 1 || !(max_wh >= 0.f)) {
        set_error("nms_class_offset: unsupported shape (batch=%d n=%d classes=%d)", batch, n, num_class); return RTOD_E_ARG;
    }
    if (ws_bytes < nms_workspace_bytes(batch, n)) { set_error("nms_class_offset: workspace too small"); return RTOD_E_ARG; }
    if (((uintptr_t)ws & 15) || ((uintptr_t)out & 15)) { set_error("nms_class_offset: workspace/out must be 16-byte aligned"); return RTOD_E_ARG; }
    NmsWs w = carve(ws, batch, n);
    hipLaunchKernelGGL(nms_zero_kernel, dim3(1), dim3(64), 0, s, w.cand, 3 * batch + 4);
    const int waves = batch * ((n + 63) / 64);
    hipLaunchKernelGGL(nms_filter_kernel<true>, dim3((waves + 3) / 4), dim3(256), 0, s, pred, batch, n, num_class, conf, w);
    hipLaunchKernelGGL(nms_sort_suppress_kernel<true>, dim3(batch), dim3(NMS_BLOCK), 0, s, n, iou_thr, w, max_wh, max_det);
    hipLaunchKernelGGL(nms_emit_kernel, dim3(batch), dim3(256), 0, s, batch, n, w, out, cap, counts);
    return hip_fail(hipGetLastError(), "nms_class_offset launch");
}

// ---------------------------------------------------------------------------------------------
// bbox_iou(box1[1,>=4], boxes[k,>=4]) -> iou[k]   (src/util.py:120-153)
__global__ void bbox_iou_kernel(const float* __restrict__ box1, const float* __restrict__ boxes, int k, int rs, float* __restrict__ iou) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const f32x4 a = {box1[0], box1[1], box1[2], box1[3]};
    const float* p = boxes + (int64_t)i * rs;
    const f32x4 b = {p[0], p[1], p[2], p[3]};
    iou[i] = iou_ref(a, b);
}

int launch_bbox_iou(const float* box1, const float* boxes, int k, int row_stride, float* iou, hipStream_t s) {
    if (!box1 || !boxes || !iou || k < 0 || row_stride < 4) { set_error("bbox_iou: bad args"); return RTOD_E_ARG; }
    if (k == 0) return RTOD_OK;
    hipLaunchKernelGGL(bbox_iou_kernel, dim3((k + 255) / 256), dim3(256), 0, s, box1, boxes, k, row_stride, iou);
    return hip_fail(hipGetLastError(), "bbox_iou launch");
}

}  // namespace rtod
