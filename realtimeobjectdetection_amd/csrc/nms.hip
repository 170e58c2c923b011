// write_results on the GPU: confidence filter + class arg-max + per-(image,class) greedy NMS.
//
// Replaces write_results / bbox_iou (reference: src/util.py:242-346, 120-153).
// Selection and order are integer work and must be bit-exact with the reference on identical
// inputs, so this file is compiled with -ffp-contract=off (no FMA contraction: the reference rounds
// after every add/mul) and relies on hipcc's correctly-rounded fp32 division.
//
//   K1 filter   one workgroup per 256 rows, one wave per 64: lanes test obj > conf (strict), the wave then scans the
//               classes of each passing row cooperatively (coalesced 4*(5+C)-byte row read; the loads of four passing
//               rows in flight together) and reduces to the FIRST arg-max; the box is converted to corners and the
//               candidate gets a 64-bit sort key (class asc | objectness desc | row asc).  Keys go to the workgroup's
//               own 256 slots, its two counts to gcnt: no global atomics (2000 candidates on one counter line cost
//               25 us), nothing to zero beforehand.
//   K2 nms      one workgroup per image: prefix over the filter workgroups' counts, keys gathered into LDS and sorted.
//               Up to 2048 candidates (the usual case): sorted by counting (rank = number of smaller keys; boxes fetched
//               meanwhile), class segments from a histogram, every (i < j) pair of a segment tested once, the pair tests
//               of a segment spread evenly over its boxes' threads, then each thread replays its segment's greedy order
//               on the 64-bit hit masks — no per-segment serial loop (segments of up to 64 boxes; longer: next line).
//               Above: bitonic network (the j <= 64 stages of a 128-key chunk belong to one wave and run without
//               workgroup barriers), one wave per class segment runs the sequential greedy loop.  Either way the
//               reference's +1-pixel IoU and strict `iou < thr` keep rule, then a block-wide scan
//               compacts survivors in sorted order.
//   K3 emit     image-major concatenation into out[D][8] = [img,x1,y1,x2,y2,obj,score,cls].
//
// Ties in objectness (undefined in the reference, torch.sort is unstable) resolve to the lower row.
#include "rtod_internal.h"
#include <atomic>

namespace rtod {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NMS_LDS_CAP = 8192;      // keys that fit the LDS sort (64 KiB)
constexpr int NMS_RANK_CAP = 2048;     // up to here the keys are sorted by counting (two per thread)
constexpr int NMS_BOX_CAP = 4096;      // candidate boxes staged in LDS for the suppression (64 KiB)
constexpr int NMS_MAX_CLASSES = 4096;  // 12 key bits
constexpr int NMS_ROW_BITS = 20;       // rows per image < 2^20
constexpr int NMS_BLOCK = 1024;
constexpr int NMS_FROWS = 256;         // rows per filter workgroup
// K2's dynamic LDS: sort keys | boxes (before the sort: the filter workgroups' offsets) | alive flags | segment starts
constexpr int NMS_OFF_BOX = NMS_LDS_CAP * 8;
constexpr int NMS_OFF_ALIVE = NMS_OFF_BOX + NMS_BOX_CAP * 16;
constexpr int NMS_OFF_SEGS = NMS_OFF_ALIVE + NMS_LDS_CAP;
constexpr int NMS_LDS_BYTES = NMS_OFF_SEGS + (NMS_MAX_CLASSES + 1) * 4;
static_assert(NMS_LDS_BYTES <= 160 * 1024, "K2 LDS");
static_assert(((1 << NMS_ROW_BITS) / NMS_FROWS + 1) * 4 <= NMS_BOX_CAP * 16, "group offsets alias the box region");

#ifdef RTOD_NMS_STAMPS    // diagnostic build: workgroup 0's phase times (shader clock) land in image 0's global alive flags
#define NMS_STAMP(i) { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(w.alive)[i] = __builtin_amdgcn_s_memtime(); }
#else
#define NMS_STAMP(i)
#endif

struct NmsWs {
    int32_t* cand;        // [B]  sortable candidates per image (score != 0), written by K2
    int32_t* cand_all;    // [B]  rows with obj > conf per image, written by K2
    int32_t* ndet;        // [B]  survivors per image
    int32_t* gcnt;        // [B][G][2]  per filter workgroup: sortable candidates, rows with obj > conf
    uint64_t* keys;       // [B][P]  P = pow2 >= n; K1 writes workgroup g's keys at [g * 256 ...], K2 compacts
    uint8_t* alive;       // [B][P]  (only when an image has more than NMS_LDS_CAP candidates)
    float* rec;           // [B][n][8]  x1,y1,x2,y2,obj,score,cls,-
    int P, G;
};

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t nms_counter_ints(int batch, int n) { return 3 * (size_t)batch + 4 + 2 * (size_t)batch * ((n + NMS_FROWS - 1) / NMS_FROWS); }

size_t nms_workspace_bytes(int batch, int n) {
    const size_t P = next_pow2(n < 1 ? 1 : n);
    size_t b = 0;
    b += align256(sizeof(int32_t) * nms_counter_ints(batch, n));
    b += align256(sizeof(uint64_t) * batch * P);
    b += align256(batch * P);
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
    w.gcnt = w.cand + 3 * batch + 4;
    p += align256(sizeof(int32_t) * nms_counter_ints(batch, n));
    w.keys = (uint64_t*)p; p += align256(sizeof(uint64_t) * batch * P);
    w.alive = (uint8_t*)p; p += align256(batch * P);
    w.rec = (float*)p;
    w.P = (int)P;
    w.G = (n + NMS_FROWS - 1) / NMS_FROWS;
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
__global__ __launch_bounds__(NMS_FROWS)
void nms_filter_kernel(const float* __restrict__ pred, int B, int n, int num_class, float conf, NmsWs w) {
    __shared__ int s_cnt[2];                      // this workgroup's sortable candidates / rows with obj > conf
    const int attrs = 5 + num_class;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x / w.G, g = blockIdx.x - b * w.G;
    const int row0 = g * NMS_FROWS + (threadIdx.x & ~63);       // the wave's first row
    const int row = row0 + lane;
    const float* img = pred + (int64_t)b * n * attrs;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();

    bool pass = false;
    if (row < n) {
        const float obj = img[(int64_t)row * attrs + 4];
        pass = (obj > conf) && (obj != 0.0f);      // mask (strict >), then nonzero(obj)  util.py:116,286
    }
    unsigned long long mask = __ballot(pass);
    if (lane == 0 && mask) atomicAdd(&s_cnt[1], __popcll(mask));
    uint64_t* gk = w.keys + (int64_t)b * w.P + g * NMS_FROWS;   // candidates of these rows: at most NMS_FROWS, and g * 256 + i < n <= P
    // one passing row: reduce the lanes' (best, index) pairs to the FIRST arg-max (torch.max(dim) on CPU returns the first maximal
    // index), convert the box to corners, record it and append the sort key
    auto finish = [&](int r, const float* p, float best, int bi, f32x4 box, float obj) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (bi == 0x7fffffff) { bi = 0; best = p[5]; }            // no finite maximum (NaN / -inf scores)
        if (lane == 0) {
            const float cx = box[0], cy = box[1], bw = box[2], bh = box[3];
            const float hw = bw / 2.0f, hh = bh / 2.0f;
            float* rec = w.rec + ((int64_t)b * n + r) * 8;
            f32x4 lo = {cx - hw, cy - hh, cx + hw, cy + hh};
            const float score = V5 ? obj * best : obj;             // V5: conf = obj * cls (one rounding, as x[:, 5:] *= x[:, 4:5])
            f32x4 hi = V5 ? f32x4{score, obj, (float)bi, 0.f} : f32x4{obj, best, (float)bi, 0.f};
            *reinterpret_cast<f32x4*>(rec) = lo;
            *reinterpret_cast<f32x4*>(rec + 4) = hi;
            if (V5 ? (score > conf) : (best != 0.0f)) {    // reference: class rows with score == 0 are dropped  util.py:305
                const int slot = atomicAdd(&s_cnt[0], 1);
                gk[slot] = ((uint64_t)bi << (32 + NMS_ROW_BITS)) | ((uint64_t)float_desc_key(score) << NMS_ROW_BITS) | (uint64_t)r;
            }
        }
    };
    if (num_class <= 128) {
        // up to 128 classes (two per lane): the loads of four passing rows are issued together, a wave with four candidates waits
        // for memory once instead of four times (the kernel's duration is its slowest wave's chain of load latencies)
        while (mask) {
            int rr[4]; float v0[4], v1[4], ob[4]; f32x4 bx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // branch-free: a missing row re-reads the wave's first row (in range: the loop runs only with a candidate), a
                // missing class lane re-reads class 0, every lane reads the box — as `cond ? p[i] : x` each load was a branch with
                // its own memory wait and the four rows' latencies ran one after the other
                const bool have = mask != 0;
                const int r = have ? row0 + __ffsll((long long)mask) - 1 : row0;
                rr[q] = have ? r : -1;
                mask &= mask - 1;                                // 0 stays 0
                const float* p = img + (int64_t)r * attrs;
                const float t0 = p[5 + (lane < num_class ? lane : 0)], t1 = p[5 + (lane + 64 < num_class ? lane + 64 : 0)];
                v0[q] = lane < num_class ? t0 : -INFINITY;
                v1[q] = lane + 64 < num_class ? t1 : -INFINITY;
                bx[q] = f32x4{p[0], p[1], p[2], p[3]}; ob[q] = p[4];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (rr[q] < 0) break;
                float best = -INFINITY; int bi = 0x7fffffff;
                if (v0[q] > best) { best = v0[q]; bi = lane; }
                if (v1[q] > best) { best = v1[q]; bi = lane + 64; }
                finish(rr[q], img + (int64_t)rr[q] * attrs, best, bi, bx[q], ob[q]);
            }
        }
    } else {
        while (mask) {
            const int r = row0 + __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float* p = img + (int64_t)r * attrs;
            float best = -INFINITY; int bi = 0x7fffffff;
            for (int c = lane; c < num_class; c += 64) {
                const float v = p[5 + c];
                if (v > best) { best = v; bi = c; }
            }
            f32x4 bx = {0.f, 0.f, 0.f, 0.f}; float ob = 0.f;
            if (lane == 0) { bx = f32x4{p[0], p[1], p[2], p[3]}; ob = p[4]; }
            finish(r, p, best, bi, bx, ob);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) w.gcnt[((int64_t)b * w.G + g) * 2 + threadIdx.x] = s_cnt[threadIdx.x];
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
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* s_keys = reinterpret_cast<uint64_t*>(smem);
    f32x4* s_box = reinterpret_cast<f32x4*>(smem + NMS_OFF_BOX);
    int* s_goff = reinterpret_cast<int*>(smem + NMS_OFF_BOX);          // [G + 1], dead once the keys are gathered
    uint8_t* s_alive = smem + NMS_OFF_ALIVE;
    int* segs = reinterpret_cast<int*>(smem + NMS_OFF_SEGS);
    __shared__ int s_scan[2 * (NMS_BLOCK / 64)];
    __shared__ int s_nseg, s_long;
    int* s_rank = reinterpret_cast<int*>(smem + NMS_OFF_ALIVE);      // [NMS_BLOCK] partial ranks, dead before the alive flags are written
    int* s_hist = segs;                                              // [NMS_MAX_CLASSES + 1] candidates per class (ranked path; the general loop's segment list otherwise)
    int* s_cstart = reinterpret_cast<int*>(smem + NMS_OFF_BOX + NMS_RANK_CAP * 16);   // [NMS_MAX_CLASSES] first sorted place of a class: above the ranked path's boxes
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t* gk = w.keys + (int64_t)b * w.P;
    const uint64_t rowmask = (1ull << NMS_ROW_BITS) - 1;

    NMS_STAMP(0)
    // exclusive prefix over the filter workgroups' candidate counts (G <= 4096: at most 4 consecutive groups per thread)
    const int G = w.G, per = (G + NMS_BLOCK - 1) / NMS_BLOCK;
    const int32_t* gc = w.gcnt + (int64_t)b * G * 2;
    int mine = 0, mine_all = 0;
    for (int q = 0; q < per; ++q) {
        const int g = tid * per + q;
        if (g < G) { mine += gc[2 * g]; mine_all += gc[2 * g + 1]; }
    }
    int incl = mine, all = mine_all;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
        all += __shfl_xor(all, off);
    }
    if (lane == 63) s_scan[wave] = incl;
    if (lane == 0) s_scan[NMS_BLOCK / 64 + wave] = all;
    if (tid == 0) { s_nseg = 0; s_long = 0; }
    s_rank[tid] = 0;
    for (int c = tid; c <= NMS_MAX_CLASSES; c += NMS_BLOCK) s_hist[c] = 0;
    __syncthreads();
    int woff = 0, cnt = 0, cnt_all = 0;
    for (int q = 0; q < NMS_BLOCK / 64; ++q) { const int v = s_scan[q]; if (q < wave) woff += v; cnt += v; cnt_all += s_scan[NMS_BLOCK / 64 + q]; }
    {
        int off = woff + incl - mine;
        for (int q = 0; q < per; ++q) {
            const int g = tid * per + q;
            if (g < G) { s_goff[g] = off; off += gc[2 * g]; }
        }
        if (tid == 0) { s_goff[G] = cnt; w.cand[b] = cnt; w.cand_all[b] = cnt_all; }
    }
    if (cnt == 0) { if (tid == 0) w.ndet[b] = 0; return; }
    __syncthreads();

    NMS_STAMP(1)
    int P = 1; while (P < cnt) P <<= 1;
    const bool in_lds = P <= NMS_LDS_CAP;
    const bool boxed = cnt <= NMS_BOX_CAP;                          // (implies in_lds)
    const bool ranked = cnt <= NMS_RANK_CAP;                        // sorted by counting, into the upper half of the key buffer
    if (in_lds) {
        // destination-driven (one independent load per key; walking the G * 256 slots cost a load latency per group): the
        // group of key d is the last one whose offset is <= d (empty groups share their successor's offset)
        for (int d = tid; d < cnt; d += NMS_BLOCK) {
            int lo = 0, hi = G;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_goff[mid] <= d) lo = mid; else hi = mid; }
            const uint64_t k = gk[lo * NMS_FROWS + (d - s_goff[lo])];
            s_keys[d] = k;
            if (ranked) atomicAdd(&s_hist[(uint32_t)(k >> (32 + NMS_ROW_BITS))], 1);
        }
        if (!ranked) for (int i = cnt + tid; i < P; i += NMS_BLOCK) s_keys[i] = ~0ull;
        __syncthreads();
    } else {
        // in place: a chunk's keys move down (off + i <= slot), past nothing that is still unread
        for (int c0 = 0; c0 < G * NMS_FROWS; c0 += NMS_BLOCK) {
            const int slot = c0 + tid, g = slot / NMS_FROWS, i = slot - g * NMS_FROWS;
            const bool have = g < G && i < s_goff[g + 1] - s_goff[g];
            const uint64_t k = have ? gk[slot] : 0;
            __syncthreads();
            if (have) gk[s_goff[g] + i] = k;
        }
        __syncthreads();
        for (int i = cnt + tid; i < P; i += NMS_BLOCK) gk[i] = ~0ull;
        __syncthreads();
    }
    NMS_STAMP(2)
    // bitonic sort, ascending
    auto exchange = [](auto* kp, int t, int j, int k) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const uint64_t a = kp[i], c = kp[l];
        const bool up = (i & k) == 0;
        if ((a > c) == up) { kp[i] = c; kp[l] = a; }
    };
    const float* rec = w.rec + (int64_t)b * n * 8;
    uint64_t* const sorted = s_keys + NMS_LDS_CAP / 2;              // ranked: the sorted keys
    if (ranked) {
        // keys are distinct (the row is part of them): a key's place is the number of smaller keys.  Every lane reads the same
        // LDS word (broadcast), so this is cnt compares per key and two barriers where the network takes (log2 P)^2 / 2 steps.
        // Up to 1024 keys the compares of a key are split over 1024 / P threads (partial ranks summed by LDS atomics); the
        // key's box is fetched meanwhile and lands in LDS at the key's sorted place.
        const int kw = P < 64 ? 64 : P;                             // key slots: a power of two, whole waves
        const int nparts = cnt <= NMS_BLOCK ? NMS_BLOCK / kw : 1;
        const int kid = cnt <= NMS_BLOCK ? (tid & (kw - 1)) : tid;
        const int part = cnt <= NMS_BLOCK ? __builtin_amdgcn_readfirstlane(tid / kw) : 0;
        const bool have0 = kid < cnt, have1 = cnt > NMS_BLOCK && tid + NMS_BLOCK < cnt;
        const uint64_t k0 = have0 ? s_keys[kid] : 0, k1 = have1 ? s_keys[tid + NMS_BLOCK] : 0;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (have0 && part == 0) b0 = *reinterpret_cast<const f32x4*>(rec + (k0 & rowmask) * 8);
        if (have1) b1 = *reinterpret_cast<const f32x4*>(rec + (k1 & rowmask) * 8);
        // class histogram -> first sorted place of every class (exclusive scan, 4 classes per thread; finished after the barrier below)
        int hc[4], hsum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { hc[q] = s_hist[4 * tid + q]; hsum += hc[q]; }
        int hincl = hsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(hincl, off); if (lane >= off) hincl += v; }
        if (lane == 63) s_scan[wave] = hincl;
        int r0 = 0, r1 = 0;
        if (cnt > NMS_BLOCK || ((wave * 64) & (kw - 1)) < cnt) {    // waves that hold keys
            const int i0 = part * cnt / nparts, i1 = (part + 1) * cnt / nparts;
            if (cnt > NMS_BLOCK) {
#pragma unroll 4
                for (int i = i0; i < i1; ++i) { const uint64_t v = s_keys[i]; r0 += v < k0; r1 += v < k1; }
            } else {
#pragma unroll 4
                for (int i = i0; i < i1; ++i) r0 += s_keys[i] < k0;
            }
        }
        if (nparts > 1 && have0) atomicAdd(&s_rank[kid], r0);
        __syncthreads();
        if (nparts > 1 && have0) r0 = s_rank[kid];
        {
            int off = hincl - hsum;
            for (int q = 0; q < wave; ++q) off += s_scan[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) { s_cstart[4 * tid + q] = off; off += hc[q]; }
        }
        if (have0 && part == 0) { sorted[r0] = k0; s_box[r0] = b0; }
        if (have1) { sorted[r1] = k1; s_box[r1] = b1; }
        __syncthreads();
    } else if (in_lds) {
        // comparators [64m, 64m + 63] of a stage with j <= 64 touch only keys [128m, 128m + 127]: those stages run wave by wave
        // (LDS operations of a wave are ordered), workgroup barriers only around the j >= 128 stages
        for (int k = 2; k <= P; k <<= 1) {
            int j = k >> 1;
            for (; j >= 128; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += NMS_BLOCK) exchange(s_keys, t, j, k);
                __syncthreads();
            }
            for (int t0 = 0; t0 < (P >> 1); t0 += NMS_BLOCK) {
                const int t = t0 + tid;
                for (int jj = j; jj > 0; jj >>= 1) {
                    if (t < (P >> 1)) exchange(s_keys, t, jj, k);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (k >= 128) __syncthreads();
        }
        __syncthreads();
    } else {
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += NMS_BLOCK) exchange(gk, t, j, k);
                __syncthreads();
            }
        }
    }
    NMS_STAMP(3)
    int base = 0;                                                   // survivors
    // everything after the sort, once per address space of the sorted keys / alive flags (LDS, or global when an image has more
    // than NMS_LDS_CAP candidates): called with LDS pointers the compiler emits ds_ instead of flat_ accesses.
    // LDS operations of one wave execute in order, so a lane's flag store is seen by the wave's next
    // read; on the global path the workgroup fence below drains the stores.
    auto general = [&](const uint64_t* keys, volatile uint8_t* alive) __attribute__((always_inline)) {
        // segment starts (class changes); order of the list is irrelevant, segments are independent
        for (int i = tid; i < cnt; i += NMS_BLOCK) {
            alive[i] = 1;
            if (boxed) s_box[i] = *reinterpret_cast<const f32x4*>(rec + (keys[i] & rowmask) * 8);
            const uint32_t c = (uint32_t)(keys[i] >> (32 + NMS_ROW_BITS));
            if (i == 0 || c != (uint32_t)(keys[i - 1] >> (32 + NMS_ROW_BITS))) segs[atomicAdd(&s_nseg, 1)] = i;
        }
        __syncthreads();
        const int nseg = s_nseg;
        auto box = [&](int i) { return boxed ? s_box[i] : *reinterpret_cast<const f32x4*>(rec + (keys[i] & rowmask) * 8); };
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
            const int len = s1 - s0;
            if (len == 1) continue;
            if (len <= 64) {
                // one box per lane: every (i < j) test first (independent), then the greedy order resolved on the bit masks:
                // j survives iff no SURVIVING i < j met the suppression test — the sequential loop's result without its
                // store -> fence -> load chain per i
                const bool in = lane < len;
                const f32x4 bj = in ? box(s0 + lane) : f32x4{0.f, 0.f, 0.f, 0.f};
                unsigned long long sup = 0;                        // bit i: box i (i < lane) suppresses this lane's box if it survives
                for (int i = 0; i + 1 < len; ++i) {
                    const f32x4 bi = {__shfl(bj[0], i), __shfl(bj[1], i), __shfl(bj[2], i), __shfl(bj[3], i)};
                    bool hit;
                    if constexpr (V5) hit = iou_offset(bi, bj, (float)cls * max_wh) > nms_thr;   // suppressed iff iou > thr
                    else hit = !(iou_ref(bi, bj) < nms_thr);                                     // keep iff iou < thr (strict)
                    if (in && lane > i && hit) sup |= 1ull << i;
                }
                unsigned long long live = 0;
                for (int i = 0; i < len; ++i) {
                    const unsigned long long m = __shfl(sup, i);
                    if ((m & live) == 0) live |= 1ull << i;
                }
                if (in) alive[s0 + lane] = (uint8_t)((live >> lane) & 1);
                continue;
            }
            for (int i = s0; i < s1; ++i) {
                if (!alive[i]) continue;                               // wave-uniform
                const f32x4 bi = box(i);
                for (int j = i + 1 + lane; j < s1; j += 64) {
                    if (!alive[j]) continue;
                    const f32x4 bj = box(j);
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
    };
    // compact survivors in sorted order: block-wide exclusive scan over chunks of NMS_BLOCK
    auto compact = [&](const uint64_t* keys, const volatile uint8_t* alive) __attribute__((always_inline)) {
        for (int c0 = 0; c0 < cnt; c0 += NMS_BLOCK) {
            const int i = c0 + tid;
            const bool a = i < cnt && alive[i];
            const unsigned long long m = __ballot(a);
            const int wprefix = __popcll(m & ((1ull << lane) - 1));
            if (lane == 0) s_scan[wave] = __popcll(m);
            __syncthreads();
            int wo = 0, tot = 0;
            for (int q = 0; q < NMS_BLOCK / 64; ++q) { const int v = s_scan[q]; if (q < wave) wo += v; tot += v; }
            const uint64_t k = a ? keys[i] : 0;
            __syncthreads();                       // all reads of keys[i] / s_scan done before overwrite
            if (a) gk[base + wo + wprefix] = k;  // survivors' keys, compacted (gk is free: sorted copy is in `keys`
            base += tot;                           //   or, on the global path, positions < i were already read)
            __syncthreads();
        }
    };
    if (ranked) {
        // at most 2048 candidates (the usual case), one or two per thread, no per-segment loop: thread j knows its place p in
        // its class segment from the class histogram, box j is tested against the p boxes ahead of it (bit i of sup: box i of the
        // segment suppresses j if it survives), and after a barrier j replays the greedy order of its segment's first p boxes on the masks:
        // j survives iff no SURVIVING i < j met the suppression test, which is what the sequential loop computes.
        // A segment longer than 64 sends the whole image to the general loop below.
        // Within a segment of L boxes, box p has p tests and its mirror L - 1 - p has the rest of L - 1: the pair shares them
        // evenly (the lower one runs its own tests and the first ones of its mirror), so that a wave's loop is L / 2 long, not L.
        uint64_t* s_supA = s_keys;                                  // [j] bits found by j's own thread (the unsorted keys are dead)
        uint64_t* s_supB = s_keys + NMS_LDS_CAP / 2 + NMS_RANK_CAP; // [j] bits found by j's mirror
        int ps[2], ss0[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = tid + q * NMS_BLOCK;
            ps[q] = 0; ss0[q] = 0;
            if (j < cnt) {
                const uint32_t cls = (uint32_t)(sorted[j] >> (32 + NMS_ROW_BITS));
                const int s0 = s_cstart[cls], L = s_hist[cls], p = j - s0, m = L - 1 - p, h = L >> 1;
                ps[q] = p; ss0[q] = s0;
                if (L > 64) { s_long = 1; continue; }
                const int nA = p <= m ? p : p - (h - m), a0 = p <= m ? 0 : h - m, nt = nA + (p < m ? h - p : 0);
                unsigned long long supA = 0, supB = 0;
#pragma unroll 2
                for (int t = 0; t < nt; ++t) {
                    const bool own = t < nA;
                    const int ii = own ? a0 + t : t - nA;
                    const f32x4 bi = s_box[s0 + ii], bj = s_box[s0 + (own ? p : m)];
                    bool hit;
                    if constexpr (V5) hit = iou_offset(bi, bj, (float)cls * max_wh) > nms_thr;   // suppressed iff iou > thr
                    else hit = !(iou_ref(bi, bj) < nms_thr);                                     // keep iff iou < thr (strict)
                    if (hit) { if (own) supA |= 1ull << ii; else supB |= 1ull << ii; }
                }
                s_supA[j] = supA;
                if (p <= m) s_supB[j] = 0;
                if (p < m) s_supB[s0 + m] = supB;
            }
        }
        __syncthreads();
        NMS_STAMP(4)
        if (s_long == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = tid + q * NMS_BLOCK;
                if (j < cnt) {
                    const int s0 = ss0[q];
                    unsigned long long live = 0;
#pragma unroll 4
                    for (int i = 0; i < ps[q]; ++i) if (((s_supA[s0 + i] | s_supB[s0 + i]) & live) == 0) live |= 1ull << i;
                    s_alive[j] = ((s_supA[j] | s_supB[j]) & live) == 0;
                }
            }
            __syncthreads();
        } else {
            general(sorted, s_alive);
        }
        NMS_STAMP(5)
        compact(sorted, s_alive);
    } else if (in_lds) {
        general(s_keys, s_alive);
        compact(s_keys, s_alive);
    } else {
        general(gk, w.alive + (int64_t)b * w.P);
        compact(gk, w.alive + (int64_t)b * w.P);
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
    NMS_STAMP(6)
}

// ------------------------------------------------------------------------------------------ K3
// (a separate launch: folded into K2 behind a completion ticket, the device-scope fences around the ticket cost K2 8-15 us,
// twice what this launch does)
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

// > 64 KiB of dynamic LDS needs the opt-in, once per device (idempotent: a second thread may repeat the calls)
static int nms_lds_opt_in() {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError(), "write_results hipGetDevice");
    if ((done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull) return RTOD_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_suppress_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, NMS_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_suppress_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, NMS_LDS_BYTES) != hipSuccess)
        return hip_fail(hipGetLastError(), "write_results LDS attribute");
    done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    return RTOD_OK;
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
    // no counter is accumulated across workgroups, so nothing needs zeroing between calls (round 1 zeroed global counters by
    // hipMemsetAsync: captured into a HIP graph, the memset node left the previous replay's counts in place on ROCm 7.2)
    if (int rc = nms_lds_opt_in()) return rc;
    hipLaunchKernelGGL(nms_filter_kernel<false>, dim3(batch * w.G), dim3(NMS_FROWS), 0, s, pred, batch, n, num_class, conf, w);
    hipLaunchKernelGGL(nms_sort_suppress_kernel<false>, dim3(batch), dim3(NMS_BLOCK), NMS_LDS_BYTES, s, n, nms, w, 0.f, 0);
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
    if (int rc = nms_lds_opt_in()) return rc;
    hipLaunchKernelGGL(nms_filter_kernel<true>, dim3(batch * w.G), dim3(NMS_FROWS), 0, s, pred, batch, n, num_class, conf, w);
    hipLaunchKernelGGL(nms_sort_suppress_kernel<true>, dim3(batch), dim3(NMS_BLOCK), NMS_LDS_BYTES, s, n, iou_thr, w, max_wh, max_det);
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
