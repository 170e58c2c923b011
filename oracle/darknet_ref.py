"""ORACLE — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product path
(``realtimeobjectdetection_amd``) never imports it and has no CPU fallback.

What it restates (reference = uguryagmur/RealTimeObjectDetection, file:line into /root/reference):

* ``RefDarknet.forward``      <- Darknet.forward            src/darknet.py:199-303
  conv -> BatchNorm(eval, eps 1e-5) -> LeakyReLU(0.1)       src/darknet.py:467-501
  shortcut / route / bilinear x2 upsample / maxpool          src/darknet.py:263-290, 587-593, 17-46, 547-555
* ``RefDarknet.load_weight_stream`` <- Darknet.load_weights  src/darknet.py:316-410
* ``predict_transform``       <- predict_transform          src/util.py:175-239
* ``confidence_mask``         <- confidence_mask            src/util.py:106-117
* ``bbox_iou``                <- bbox_iou                   src/util.py:120-153
* ``write_results``           <- write_results              src/util.py:242-346

The arithmetic engine is PyTorch's CPU ops (what the reference itself executes)
for the network and plain float32 numpy for the post-processing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference in the build
container (with a stub ``cv2`` module) and writes the fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against them
(network outputs bit-identical, ``write_results`` bit-identical including row order) and, when
/root/reference is present, against the live reference.  BatchNorm runs in eval mode (running
statistics): the canonical mode chosen in SURVEY.md F2.
"""
import numpy as np
import torch
import torch.nn.functional as F

from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir


class RefDarknet:
    def __init__(self, cfg_text: str, height: int, width: int = None):
        self.blocks = parse_cfg_text(cfg_text)
        self.ir = build_ir(self.blocks, height, width)
        self.height = height
        self.params = {}   # layer index -> dict of torch tensors

    # -- weights ---------------------------------------------------------------------------
    def load_weight_stream(self, w: np.ndarray):
        """Consume a ``.weights`` float stream (src/darknet.py:316-410)."""
        p = 0
        w = np.ascontiguousarray(w, dtype=np.float32)
        for L in self.ir.layers:
            if L.type != "convolutional":
                continue
            d = {}
            c = L.cout
            if L.bn:
                for name in ("beta", "gamma", "mean", "var"):
                    d[name] = torch.from_numpy(w[p:p + c].copy()); p += c
            else:
                d["bias"] = torch.from_numpy(w[p:p + c].copy()); p += c
            n = c * L.cin * L.size * L.size
            d["weight"] = torch.from_numpy(w[p:p + n].copy()).view(c, L.cin, L.size, L.size)
            p += n
            self.params[L.index] = d
        return p

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, keep_layers=False, batch_stats=False):
        """``x`` float32 ``[B,3,H,W]`` -> ``[B,N,5+C]``.  BatchNorm: running statistics (eval, the canonical mode) or, with
        ``batch_stats``, the statistics of the batch — what the reference's callers run without .eval();
        the running buffers are not updated here."""
        outputs = {}
        detections = None
        for L in self.ir.layers:
            i = L.index
            if L.type == "convolutional":
                p = self.params[i]
                x = F.conv2d(x, p["weight"], p.get("bias"), L.stride, L.pad)
                if L.bn:
                    if batch_stats:
                        x = F.batch_norm(x, None, None, p["gamma"], p["beta"], training=True, momentum=0.1, eps=1e-5)
                    else:
                        x = F.batch_norm(x, p["mean"], p["var"], p["gamma"], p["beta"],
                                         training=False, momentum=0.1, eps=1e-5)
                if L.leaky:
                    x = F.leaky_relu(x, 0.1)
                elif L.silu:                       # cfg extension, not reference behaviour: checked against torch's own op
                    x = F.silu(x)
            elif L.type == "upsample":
                if L.nearest:                      # cfg extension (YOLOv5-style blocks)
                    x = F.interpolate(x, scale_factor=2, mode="nearest")
                else:
                    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
            elif L.type == "maxpool":
                if L.pool_pad:                     # cfg extension: symmetric -inf padding (SPPF: size 5, stride 1, pad 2)
                    x = F.max_pool2d(x, L.size, L.stride, L.pool_pad)
                elif L.stride != 1:
                    x = F.max_pool2d(x, L.size, L.stride)
                else:
                    x = F.pad(x, (0, L.size - 1, 0, L.size - 1), mode="replicate")
                    x = F.max_pool2d(x, L.size, L.size - 1)
            elif L.type == "shortcut":
                x = outputs[L.srcs[0]] + outputs[L.srcs[1]]
            elif L.type == "route":
                if len(L.srcs) == 1:
                    x = outputs[L.srcs[0]]
                else:
                    x = torch.cat([outputs[s] for s in L.srcs], 1)
            elif L.type == "yolo":
                x = (predict_transform_v5 if L.decode_v5 else predict_transform)(x, self.height, L.anchors, L.classes)
                detections = x if detections is None else torch.cat((detections, x), 1)
                outputs[i] = outputs[i - 1]
                continue
            outputs[i] = x
        if keep_layers:
            return detections, outputs
        return detections

    __call__ = forward


def predict_transform_v5(prediction: torch.Tensor, inp_dim: int, anchors, num_class: int) -> torch.Tensor:
    """YOLOv5-style head arithmetic (cfg extension ``decode=v5``; NOT reference source — the published Detect layer restated:
    y = sigmoid(x); xy = (y * 2 - 0.5 + grid) * stride; wh = (y * 2) ** 2 * anchor (pixels); PARITY UNPINNED).  Rows are kept
    in this path's order (cell-major, anchor inner), like predict_transform's."""
    B, G = prediction.size(0), prediction.size(2)
    stride = inp_dim // G
    attrs, A = 5 + num_class, len(anchors)
    p = prediction.reshape(B, attrs * A, G * G).transpose(1, 2).contiguous().view(B, G * G * A, attrs)
    y = torch.sigmoid(p)
    g = torch.arange(G)
    gy, gx = torch.meshgrid(g, g, indexing="ij")
    off = torch.stack((gx.reshape(-1), gy.reshape(-1)), 1).repeat(1, A).view(-1, 2).unsqueeze(0).float()
    anc = torch.FloatTensor([(float(a[0]), float(a[1])) for a in anchors]).repeat(G * G, 1).unsqueeze(0)
    out = y.clone()
    out[:, :, 0:2] = (y[:, :, 0:2] * 2.0 - 0.5 + off) * stride
    out[:, :, 2:4] = (y[:, :, 2:4] * 2) ** 2 * anc
    return out


def predict_transform(prediction: torch.Tensor, inp_dim: int, anchors, num_class: int,
                      train: bool = False) -> torch.Tensor:
    """Head decode, SURVEY.md App. B.6 (src/util.py:193-237)."""
    B = prediction.size(0)
    G = prediction.size(2)
    stride = inp_dim // G
    G = inp_dim // stride
    attrs = 5 + num_class
    A = len(anchors)
    p = prediction.reshape(B, attrs * A, G * G).transpose(1, 2).contiguous().view(B, G * G * A, attrs)
    p = p.clone()
    p[:, :, 0] = torch.sigmoid(p[:, :, 0])
    p[:, :, 1] = torch.sigmoid(p[:, :, 1])
    p[:, :, 4:] = torch.sigmoid(p[:, :, 4:])
    if not train:
        anc = torch.FloatTensor([(a[0] / stride, a[1] / stride) for a in anchors])
        g = torch.arange(G)
        gy, gx = torch.meshgrid(g, g, indexing="ij")
        off = torch.stack((gx.reshape(-1), gy.reshape(-1)), 1)          # [G*G, 2] (x, y)
        off = off.repeat(1, A).view(-1, 2).unsqueeze(0)                  # [1, G*G*A, 2]
        p[:, :, :2] += off
        p[:, :, 2:4] = torch.exp(p[:, :, 2:4]) * anc.repeat(G * G, 1).unsqueeze(0)
        p[:, :, :4] *= stride
    return p


def confidence_mask(t: torch.Tensor, confidence: float) -> torch.Tensor:
    return t * (t[:, :, 4] > confidence).float().unsqueeze(2)


def bbox_iou_np(b1: np.ndarray, b2: np.ndarray) -> np.ndarray:
    """IoU with the +1 pixel convention, float32, reference op order (src/util.py:131-151)."""
    f = np.float32
    ix1 = np.maximum(b1[..., 0], b2[..., 0])
    iy1 = np.maximum(b1[..., 1], b2[..., 1])
    ix2 = np.minimum(b1[..., 2], b2[..., 2])
    iy2 = np.minimum(b1[..., 3], b2[..., 3])
    iw = np.maximum((ix2 - ix1).astype(f) + f(1), f(0)).astype(f)
    ih = np.maximum((iy2 - iy1).astype(f) + f(1), f(0)).astype(f)
    inter = (iw * ih).astype(f)
    a1 = (((b1[..., 2] - b1[..., 0]).astype(f) + f(1)) * ((b1[..., 3] - b1[..., 1]).astype(f) + f(1))).astype(f)
    a2 = (((b2[..., 2] - b2[..., 0]).astype(f) + f(1)) * ((b2[..., 3] - b2[..., 1]).astype(f) + f(1))).astype(f)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / ((a1 + a2).astype(f) - inter).astype(f)).astype(f)


def bbox_iou(box1: torch.Tensor, box2: torch.Tensor) -> torch.Tensor:
    return torch.from_numpy(bbox_iou_np(box1.numpy().astype(np.float32), box2.numpy().astype(np.float32)))


def write_results(prediction, num_class: int, confidence: float = 0.6, nms_conf: float = 0.4):
    """Per-class greedy NMS, SURVEY.md App. B.7 (src/util.py:242-346).

    Returns float32 ``[D,8]`` rows ``[img,x1,y1,x2,y2,obj,cls_score,cls]`` (torch tensor), an
    empty ``[0,8]`` tensor when candidates existed but every one had class score 0 (the
    reference concatenates an empty block in that case), or the int ``0``.
    Tie-break for equal objectness (undefined in the reference: torch.sort is unstable):
    lower row index first.
    """
    p = prediction.numpy() if isinstance(prediction, torch.Tensor) else np.asarray(prediction)
    p = p.astype(np.float32, copy=False)
    f = np.float32
    B = p.shape[0]
    conf = f(confidence)
    thr = f(nms_conf)
    out = []
    wrote = False
    for b in range(B):
        img = p[b]
        obj = img[:, 4]
        keep = np.nonzero((obj > conf) & (obj != 0))[0]
        if keep.size == 0:
            continue
        r = img[keep]
        half_w = r[:, 2] / f(2)
        half_h = r[:, 3] / f(2)
        boxes = np.stack([r[:, 0] - half_w, r[:, 1] - half_h, r[:, 0] + half_w, r[:, 1] + half_h], 1).astype(f)
        cls_scores = r[:, 5:5 + num_class]
        cls = np.argmax(cls_scores, 1)                       # first max
        score = cls_scores[np.arange(keep.size), cls]
        for c in np.unique(cls):
            wrote = True
            sel = np.nonzero((cls == c) & (score != 0))[0]
            if sel.size == 0:
                continue
            order = np.lexsort((keep[sel], -r[sel, 4].astype(np.float64)))   # obj desc, row asc
            sel = sel[order]
            bx = boxes[sel]
            alive = np.ones(sel.size, dtype=bool)
            for i in range(sel.size):
                if not alive[i]:
                    continue
                if i + 1 < sel.size:
                    iou = bbox_iou_np(bx[i][None, :], bx[i + 1:])
                    alive[i + 1:] &= (iou < thr)
            for i in np.nonzero(alive)[0]:
                s = sel[i]
                out.append([f(b), bx[i, 0], bx[i, 1], bx[i, 2], bx[i, 3], r[s, 4], score[s], f(c)])
    if not wrote:
        return 0
    if not out:
        return torch.zeros((0, 8), dtype=torch.float32)
    return torch.from_numpy(np.asarray(out, dtype=np.float32))


def nms_class_offset(prediction: torch.Tensor, conf_thres=0.25, iou_thres=0.45, max_wh=7680.0, max_det=300) -> np.ndarray:
    """CPU restatement of the published YOLOv5 post-processing (class-offset batched NMS).  NOT reference source: the
    reference obtains YOLOv5 through torch.hub (detect.py:255-285), nothing of it exists offline, and torchvision is not
    installed — so this follows the documented algorithm step by step in float32 numpy (PARITY UNPINNED): obj > conf_thres;
    cls *= obj; box = xywh -> xyxy; (conf, j) = max over classes; conf > conf_thres; sort by conf descending; boxes +
    j * max_wh; greedy NMS with IoU = inter / (a1 + a2 - inter) (no +1), suppress at IoU > iou_thres; first max_det.
    Returns rows [img, x1, y1, x2, y2, conf, obj, cls]."""
    p = prediction.detach().cpu().numpy().astype(np.float32)
    rows = []
    for b in range(p.shape[0]):
        x = p[b]
        x = x[x[:, 4] > np.float32(conf_thres)]
        if not len(x):
            continue
        obj = x[:, 4].copy()
        cls = x[:, 5:] * x[:, 4:5]
        half_w, half_h = x[:, 2] / np.float32(2), x[:, 3] / np.float32(2)
        box = np.stack([x[:, 0] - half_w, x[:, 1] - half_h, x[:, 0] + half_w, x[:, 1] + half_h], 1).astype(np.float32)
        j = cls.argmax(1)                                    # first maximal index
        conf = cls[np.arange(len(cls)), j]
        keep = conf > np.float32(conf_thres)
        box, conf, j, obj = box[keep], conf[keep], j[keep], obj[keep]
        order = np.argsort(-conf, kind="stable")             # ties: lower row first
        box, conf, j, obj = box[order], conf[order], j[order], obj[order]
        sb = box + (j.astype(np.float32) * np.float32(max_wh))[:, None]
        area = (sb[:, 2] - sb[:, 0]) * (sb[:, 3] - sb[:, 1])
        alive = np.ones(len(sb), bool)
        kept = []
        for i in range(len(sb)):
            if not alive[i]:
                continue
            kept.append(i)
            if len(kept) == max_det:
                break
            iw = np.maximum(np.minimum(sb[i, 2], sb[i + 1:, 2]) - np.maximum(sb[i, 0], sb[i + 1:, 0]), np.float32(0))
            ih = np.maximum(np.minimum(sb[i, 3], sb[i + 1:, 3]) - np.maximum(sb[i, 1], sb[i + 1:, 1]), np.float32(0))
            inter = iw * ih
            iou = inter / ((area[i] + area[i + 1:]) - inter)
            alive[i + 1:] &= ~(iou > np.float32(iou_thres))
        for i in kept:
            rows.append([b, box[i, 0], box[i, 1], box[i, 2], box[i, 3], conf[i], obj[i], j[i]])
    return np.array(rows, dtype=np.float32).reshape(-1, 8)
