// prep_image on the GPU (reference: src/util.py:349-397 letterbox_image + prep_image): aspect-preserving
// bicubic resize of a uint8 HWC image into a grey (128) inp_dim x inp_dim canvas, channel swap to RGB,
// /255, planar [3,R,R] float32 — the network input — so only the uint8 image crosses PCIe.
//
// Parity status: UNPINNED.  The reference calls cv2.resize(..., INTER_CUBIC); OpenCV is not installed in the
// build container, so this kernel follows OpenCV's documented bicubic (a = -0.75, half-pixel centres,
// replicated border, result rounded and saturated to uint8) in float arithmetic; OpenCV's own uint8 path
// uses 11-bit fixed-point coefficients and can differ by 1/255 in a pixel.  Checked against
// oracle/prep_ref.py (numpy restatement of the same definition).
#include "rtod_internal.h"

namespace rtod {

__device__ __forceinline__ void cubic_coeffs(float t, float* c) {
    const float A = -0.75f;
    c[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    c[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    c[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

__global__ void letterbox_kernel(const unsigned char* __restrict__ img, int h, int w, int new_h, int new_w,
                                 int off_y, int off_x, int swap_rb, int R, float* __restrict__ out) {
    const int total = R * R;
    const float sx = (float)w / (float)new_w, sy = (float)h / (float)new_h;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int oy = p / R, ox = p - oy * R;
        const int ry = oy - off_y, rx = ox - off_x;
        float v[3] = {128.f, 128.f, 128.f};
        if (ry >= 0 && ry < new_h && rx >= 0 && rx < new_w) {
            float fy = ((float)ry + 0.5f) * sy - 0.5f, fx = ((float)rx + 0.5f) * sx - 0.5f;
            const int iy = (int)floorf(fy), ix = (int)floorf(fx);
            float cy[4], cx[4];
            cubic_coeffs(fy - (float)iy, cy);
            cubic_coeffs(fx - (float)ix, cx);
            float acc[3] = {0.f, 0.f, 0.f};
            for (int j = 0; j < 4; ++j) {
                int yy = iy - 1 + j; yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
                float row[3] = {0.f, 0.f, 0.f};
                for (int i = 0; i < 4; ++i) {
                    int xx = ix - 1 + i; xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
                    const unsigned char* q = img + ((int64_t)yy * w + xx) * 3;
                    row[0] += cx[i] * (float)q[0]; row[1] += cx[i] * (float)q[1]; row[2] += cx[i] * (float)q[2];
                }
                acc[0] += cy[j] * row[0]; acc[1] += cy[j] * row[1]; acc[2] += cy[j] * row[2];
            }
            for (int c = 0; c < 3; ++c) { float r = rintf(acc[c]); v[c] = r < 0.f ? 0.f : (r > 255.f ? 255.f : r); }
        }
        const int c0 = swap_rb ? 2 : 0, c2 = swap_rb ? 0 : 2;
        out[p] = v[c0] / 255.0f;
        out[total + p] = v[1] / 255.0f;
        out[2 * total + p] = v[c2] / 255.0f;
    }
}

int launch_prep_image(const unsigned char* img, int h, int w, int bgr, int inp_dim, float* out, hipStream_t s) {
    if (!img || !out || h < 1 || w < 1 || inp_dim < 1) { set_error("prep_image: bad args"); return RTOD_E_ARG; }
    // letterbox geometry exactly as the reference computes it (util.py:360-370; Python float division, int() truncation)
    const double sc = std::min((double)inp_dim / (double)w, (double)inp_dim / (double)h);
    const int new_w = (int)((double)w * sc), new_h = (int)((double)h * sc);
    if (new_w < 1 || new_h < 1) { set_error("prep_image: degenerate image %dx%d", w, h); return RTOD_E_ARG; }
    const int off_y = (inp_dim - new_h) / 2, off_x = (inp_dim - new_w) / 2;
    const int total = inp_dim * inp_dim;
    int grid = (total + 255) / 256; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(letterbox_kernel, dim3(grid), dim3(256), 0, s, img, h, w, new_h, new_w, off_y, off_x, bgr ? 1 : 0, inp_dim, out);
    return hip_fail(hipGetLastError(), "prep_image launch");
}

}  // namespace rtod
