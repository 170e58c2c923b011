"""ORACLE (test infrastructure only) — numpy restatement of the reference's letterbox_image + prep_image
(src/util.py:349-397) with cv2.resize(INTER_CUBIC) replaced by its documented definition: bicubic with
a = -0.75, half-pixel centres, replicated border, rounded + saturated to uint8.

Parity status: UNPINNED — OpenCV is not installed in the build container and the reference holds no
fixture for this step, so this cannot be checked against cv2 itself (OpenCV's uint8 path uses 11-bit
fixed-point coefficients; expect occasional 1/255 differences).  It pins the HIP kernel to a written spec."""
import numpy as np


def _coeffs(t):
    A = np.float32(-0.75)
    t = t.astype(np.float32)
    c0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
    c1 = ((A + 2) * t - (A + 3)) * t * t + 1
    c2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
    c3 = 1 - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], -1).astype(np.float32)


def resize_cubic_u8(img, new_w, new_h):
    h, w = img.shape[:2]
    sx, sy = np.float32(w) / np.float32(new_w), np.float32(h) / np.float32(new_h)
    fx = (np.arange(new_w, dtype=np.float32) + np.float32(0.5)) * sx - np.float32(0.5)
    fy = (np.arange(new_h, dtype=np.float32) + np.float32(0.5)) * sy - np.float32(0.5)
    ix, iy = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
    cx, cy = _coeffs(fx - ix), _coeffs(fy - iy)
    src = img.astype(np.float32)
    out = np.zeros((new_h, new_w, 3), np.float32)
    for j in range(4):
        yy = np.clip(iy - 1 + j, 0, h - 1)
        row = np.zeros((new_h, new_w, 3), np.float32)
        for i in range(4):
            xx = np.clip(ix - 1 + i, 0, w - 1)
            row += cx[None, :, i, None] * src[yy][:, xx]
        out += cy[:, j, None, None] * row
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def letterbox_image(img, inp_dim):
    img_w, img_h = img.shape[1], img.shape[0]
    w, h = inp_dim
    new_w = int(img_w * min(w / img_w, h / img_h))
    new_h = int(img_h * min(w / img_w, h / img_h))
    resized = resize_cubic_u8(img, new_w, new_h)
    canvas = np.full((inp_dim[1], inp_dim[0], 3), 128)
    canvas[(h - new_h) // 2:(h - new_h) // 2 + new_h, (w - new_w) // 2:(w - new_w) // 2 + new_w, :] = resized
    return canvas


def prep_image(img, inp_dim, mode="BGR"):
    assert mode in ("BGR", "RGB")
    c = letterbox_image(img, (inp_dim, inp_dim))
    c = c.transpose((2, 0, 1)).copy() if mode == "RGB" else c[:, :, ::-1].transpose((2, 0, 1)).copy()
    return (c.astype(np.float32) / np.float32(255.0))[None]
