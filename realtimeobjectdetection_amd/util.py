"""Host-side mirror of the hot-path half of the reference's ``src/util.py`` on librtod.so.

Same names, argument meaning and return conventions as the reference (SURVEY.md §8 b):

* ``predict_transform(prediction, inp_dim, anchors, num_class, CUDA, TRAIN=False)``  src/util.py:175-239
* ``write_results(prediction, num_class, confidence=0.6, nms_conf=0.4)``              src/util.py:242-346
  -> float32 ``[D,8]`` tensor on prediction's device, or the Python int ``0``
* ``confidence_mask(tensor, confidence)``                                              src/util.py:106-117
* ``bbox_iou(box1, box2)``                                                             src/util.py:120-153

All tensors must be CUDA (ROCm) float32 tensors; there is no CPU fallback.
"""
import ctypes as C

import torch

from . import _ffi

_ws_cache = {}

COCO_NAMES = ("person bicycle car motorbike aeroplane bus train truck boat|traffic light|fire hydrant|stop sign|parking meter|bench "
              "bird cat dog horse sheep cow elephant bear zebra giraffe backpack umbrella handbag tie suitcase frisbee skis snowboard|"
              "sports ball|kite|baseball bat|baseball glove|skateboard surfboard|tennis racket|bottle|wine glass|cup fork knife spoon bowl "
              "banana apple sandwich orange broccoli carrot|hot dog|pizza donut cake chair sofa pottedplant bed diningtable toilet "
              "tvmonitor laptop mouse remote keyboard|cell phone|microwave oven toaster sink refrigerator book clock vase scissors|"
              "teddy bear|hair drier|toothbrush")


def load_classes(names_file_path: str = None) -> list:
    """Class labels (reference: src/util.py:400-411 reads data/coco.names).  Without a file the 80 COCO names
    of the Darknet distribution are returned."""
    if names_file_path:
        with open(names_file_path, "r") as fp:
            return fp.read().split("\n")[:-1]
    out = []
    for chunk in COCO_NAMES.split("|"):
        chunk = chunk.strip()
        if " " in chunk and chunk in ("traffic light", "fire hydrant", "stop sign", "parking meter", "sports ball", "baseball bat",
                                      "baseball glove", "tennis racket", "wine glass", "hot dog", "cell phone", "teddy bear",
                                      "hair drier"):
            out.append(chunk)
        else:
            out.extend(chunk.split())
    return out


def prep_image(img, inp_dim, mode="BGR", device=None) -> torch.Tensor:
    """uint8 HWC image (numpy or torch) -> float32 ``[1,3,inp_dim,inp_dim]`` network input on the GPU
    (reference: src/util.py:375-397 + letterbox_image 349-372).  Only the uint8 pixels cross PCIe; resize,
    padding, channel swap and /255 run in a HIP kernel.  ``mode='BGR'`` (OpenCV order, the reference's default)
    is swapped to RGB.  Parity with cv2.INTER_CUBIC is unpinned (OpenCV is not available offline)."""
    assert mode == "BGR" or mode == "RGB"
    t = torch.as_tensor(img)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.size(2) != 3:
        raise ValueError("prep_image: expected uint8 [H,W,3], got %s %s" % (t.dtype, tuple(t.shape)))
    if device is None:
        device = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    t = t.to(device).contiguous()
    out = torch.empty((1, 3, int(inp_dim), int(inp_dim)), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _ffi.check(_ffi.lib().rtod_prep_image(C.c_void_p(t.data_ptr()), t.size(0), t.size(1), 1 if mode == "BGR" else 0,
                                              int(inp_dim), C.c_void_p(out.data_ptr()), _stream(t.device)))
    return out


def rescale_boxes(output: torch.Tensor, im_dim_list: torch.Tensor, inp_dim: int) -> torch.Tensor:
    """Undo the letterbox on detection rows ``[img,x1,y1,x2,y2,...]`` and clamp to the image
    (reference: detect.py:120-136, which hard-codes 416 for the scale; here the actual ``inp_dim``).
    ``im_dim_list``: ``[n_images, 2]`` = (width, height) per image.  Returns a new tensor."""
    out = output.clone()
    dims = im_dim_list.to(out.device, torch.float32)[out[:, 0].long()]
    scale = torch.min(float(inp_dim) / dims, 1)[0].view(-1, 1)
    out[:, [1, 3]] -= (inp_dim - scale * dims[:, 0].view(-1, 1)) / 2
    out[:, [2, 4]] -= (inp_dim - scale * dims[:, 1].view(-1, 1)) / 2
    out[:, 1:5] /= scale
    out[:, [1, 3]] = torch.minimum(torch.clamp(out[:, [1, 3]], min=0.0), dims[:, 0].view(-1, 1))
    out[:, [2, 4]] = torch.minimum(torch.clamp(out[:, [2, 4]], min=0.0), dims[:, 1].view(-1, 1))
    return out


def _need_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s: expected a CUDA (ROCm) tensor; this build has no CPU path" % name)
    if t.dtype != torch.float32:
        raise ValueError("%s: expected float32, got %s" % (name, t.dtype))


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def predict_transform(prediction, inp_dim, anchors, num_class, CUDA=True, TRAIN=False) -> torch.Tensor:
    _need_cuda(prediction, "predict_transform")
    if prediction.dim() != 4 or prediction.size(2) != prediction.size(3):
        raise ValueError("predict_transform: expected [B, A*(5+C), G, G], got %s" % (tuple(prediction.shape),))
    lib = _ffi.lib()
    B, ch, G = prediction.size(0), prediction.size(1), prediction.size(2)
    attrs = 5 + int(num_class)
    A = len(anchors)
    if ch != A * attrs:
        raise ValueError("predict_transform: %d channels != %d anchors * %d attrs" % (ch, A, attrs))
    x = prediction.contiguous()
    out = torch.empty((B, G * G * A, attrs), dtype=torch.float32, device=x.device)
    anc = (C.c_float * (2 * A))(*[float(v) for a in anchors for v in a])
    with torch.cuda.device(x.device):
        _ffi.check(lib.rtod_predict_transform(C.c_void_p(x.data_ptr()), B, attrs, G, A, anc, int(inp_dim),
                                              1 if TRAIN else 0, C.c_void_p(out.data_ptr()), _stream(x.device)))
    return out


def confidence_mask(tensor: torch.Tensor, confidence: float) -> torch.Tensor:
    _need_cuda(tensor, "confidence_mask")
    if tensor.dim() != 3 or tensor.size(2) < 5:
        raise ValueError("confidence_mask: expected [B,N,>=5]")
    x = tensor.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _ffi.check(_ffi.lib().rtod_confidence_mask(C.c_void_p(x.data_ptr()), x.size(0) * x.size(1), x.size(2),
                                                   float(confidence), C.c_void_p(out.data_ptr()), _stream(x.device)))
    return out


def bbox_iou(box1: torch.Tensor, box2: torch.Tensor) -> torch.Tensor:
    """IoU of one box against k boxes with the reference's +1 pixel convention."""
    _need_cuda(box1, "bbox_iou")
    _need_cuda(box2, "bbox_iou")
    b1 = box1.reshape(-1, box1.size(-1))
    b2 = box2.reshape(-1, box2.size(-1))
    if b1.size(0) != 1 or b1.size(1) < 4 or b2.size(1) < 4:
        raise NotImplementedError("bbox_iou: one box [1,>=4] against k boxes [k,>=4] (the reference's only use)")
    b1 = b1.contiguous()
    b2 = b2.contiguous()
    k = b2.size(0)
    out = torch.empty((k,), dtype=torch.float32, device=b2.device)
    with torch.cuda.device(b2.device):
        _ffi.check(_ffi.lib().rtod_bbox_iou(C.c_void_p(b1.data_ptr()), C.c_void_p(b2.data_ptr()), k, b2.size(1),
                                            C.c_void_p(out.data_ptr()), _stream(b2.device)))
    return out


_host_cache = {}


def _host_ints(dev):
    """Pinned int32[4] per device: landing place of write_results' counts and the split-f16 range flag."""
    h = _host_cache.get(dev.index)
    if h is None:
        h = _host_cache[dev.index] = torch.empty(4, dtype=torch.int32).pin_memory()
    return h


def _nms_buffers(dev, B, n, cap):
    key = (dev.index, B, n, cap)
    buf = _ws_cache.get(key)
    if buf is None:
        nbytes = C.c_size_t()
        _ffi.check(_ffi.lib().rtod_write_results_workspace(B, n, C.byref(nbytes)))
        ws = torch.empty((nbytes.value + 15) // 16 * 4, dtype=torch.int32, device=dev)
        out = torch.empty((max(cap, 1), 8), dtype=torch.float32, device=dev)
        counts = torch.zeros((2 + B + 2,), dtype=torch.int32, device=dev)
        buf = (ws, out, counts, nbytes.value)
        if len(_ws_cache) > 16:
            _ws_cache.clear()
        _ws_cache[key] = buf
    return buf


def write_results_async(prediction, num_class, confidence=0.6, nms_conf=0.4, cap=None):
    """Enqueue filter + NMS; returns device tensors ``(rows[cap,8], counts)`` without synchronising.

    ``counts[0]`` = D (valid rows; may exceed cap), ``counts[1]`` = candidates over the batch,
    ``counts[2:2+B]`` = detections per image.  The buffers are reused by the next call with the
    same shape: consume (or clone) them before calling again.
    """
    _need_cuda(prediction, "write_results")
    if prediction.dim() != 3 or prediction.size(2) != 5 + int(num_class):
        raise ValueError("write_results: expected [B,N,5+num_class], got %s" % (tuple(prediction.shape),))
    x = prediction.contiguous()
    B, n = x.size(0), x.size(1)
    if cap is None:
        cap = min(B * n, 16384)
    ws, out, counts, nbytes = _nms_buffers(x.device, B, n, int(cap))
    with torch.cuda.device(x.device):
        _ffi.check(_ffi.lib().rtod_write_results(C.c_void_p(x.data_ptr()), B, n, int(num_class), float(confidence),
                                                 float(nms_conf), C.c_void_p(out.data_ptr()), int(cap),
                                                 C.c_void_p(counts.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes,
                                                 _stream(x.device)))
    return out, counts


def nms_class_offset(prediction, num_class, conf_thres=0.25, iou_thres=0.45, max_wh=7680.0, max_det=300):
    """YOLOv5-style post-processing (NOT reference behaviour: the reference fetches its YOLOv5 model from the hub,
    detect.py:255-285; this follows the published class-offset batched NMS — parity unpinned).  ``prediction``
    ``[B,N,5+C]`` rows ``(cx,cy,w,h,obj,cls...)`` -> ``[D,8]`` rows ``[img,x1,y1,x2,y2,conf,obj,cls]``, per image by
    descending ``conf = obj * max cls`` (ties: lower row first), at most ``max_det`` per image; one host sync."""
    _need_cuda(prediction, "nms_class_offset")
    if prediction.dim() != 3 or prediction.size(2) != 5 + int(num_class):
        raise ValueError("nms_class_offset: expected [B,N,5+num_class], got %s" % (tuple(prediction.shape),))
    x = prediction.contiguous()
    B, n = x.size(0), x.size(1)
    cap = B * min(n, int(max_det))
    ws, out, counts, nbytes = _nms_buffers(x.device, B, n, int(cap))
    with torch.cuda.device(x.device):
        _ffi.check(_ffi.lib().rtod_nms_class_offset(C.c_void_p(x.data_ptr()), B, n, int(num_class), float(conf_thres), float(iou_thres),
                                                    float(max_wh), int(max_det), C.c_void_p(out.data_ptr()), int(cap),
                                                    C.c_void_p(counts.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, _stream(x.device)))
    return out[:int(counts[0].item())].clone()


def nms_class_offset_async(prediction, num_class, conf_thres=0.25, iou_thres=0.45, max_wh=7680.0, max_det=300, cap=None):
    """nms_class_offset without the host synchronisation: device tensors ``(rows[cap,8], counts)``, ``counts[0]`` = valid rows
    (buffers are reused by the next call of the same shape, like write_results_async's)."""
    _need_cuda(prediction, "nms_class_offset")
    if prediction.dim() != 3 or prediction.size(2) != 5 + int(num_class):
        raise ValueError("nms_class_offset: expected [B,N,5+num_class], got %s" % (tuple(prediction.shape),))
    x = prediction.contiguous()
    B, n = x.size(0), x.size(1)
    if cap is None:
        cap = B * min(n, int(max_det))
    ws, out, counts, nbytes = _nms_buffers(x.device, B, n, int(cap))
    with torch.cuda.device(x.device):
        _ffi.check(_ffi.lib().rtod_nms_class_offset(C.c_void_p(x.data_ptr()), B, n, int(num_class), float(conf_thres), float(iou_thres),
                                                    float(max_wh), int(max_det), C.c_void_p(out.data_ptr()), int(cap),
                                                    C.c_void_p(counts.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, _stream(x.device)))
    return out, counts


def write_results(prediction, num_class, confidence=0.6, nms_conf=0.4):
    """Reference-compatible: new ``[D,8]`` tensor, an empty ``[0,8]`` tensor when candidates existed
    but none had a non-zero class score (what the reference's concatenation yields), or int ``0``."""
    B, n = prediction.size(0), prediction.size(1)
    out, counts = write_results_async(prediction, num_class, confidence, nms_conf)
    from .darknet import take_pending_overflow, raise_overflow
    model, flag = take_pending_overflow(prediction)     # split-f16 range guard of the forward that produced `prediction`
    host = _host_ints(prediction.device)
    host[:2].copy_(counts[:2], non_blocking=True)
    if flag is not None:
        host[2:3].copy_(flag.reshape(-1)[:1], non_blocking=True)
    torch.cuda.current_stream(prediction.device).synchronize()   # ONE host sync (counts + flag), like the reference's own .tolist()/nonzero
    D, cand = int(host[0]), int(host[1])
    if flag is not None and int(host[2]) != 0:
        raise_overflow(model)
    if D > out.size(0):                        # more detections than the default capacity: redo at full size
        out, counts = write_results_async(prediction, num_class, confidence, nms_conf, cap=B * n)
        D = int(counts[0].item())
    if cand == 0:
        return 0
    return out[:D].clone()
