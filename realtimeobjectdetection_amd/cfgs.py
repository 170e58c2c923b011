"""Generators for the Darknet ``.cfg`` texts of the two networks the reference ships.

The reference reads ``cfg/yolov3.cfg`` and ``cfg/yolov3-tiny.cfg`` (reference:
src/darknet.py:412-447 parses them, detect.py:185 picks the path from params.json).  Those
files do not travel to the GPU box, so this module *generates* equivalent cfg text from a
compact architecture description.  ``tests/test_oracle_golden.py::test_generated_cfg_equals_reference_cfg``
checks (in the build container, where /root/reference is mounted) that parsing the generated text
yields layer blocks identical to parsing the reference's files.

Only keys that the hot path reads are emitted.
"""

_ANCHORS_V3 = "10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326"
_ANCHORS_TINY = "10,14,  23,27,  37,58,  81,82,  135,169,  344,319"


def _net(height, width):
    return ["[net]", "batch=1", "subdivisions=1", f"width={width}", f"height={height}",
            "channels=3", ""]


def _conv(filters, size, stride, bn=True, act="leaky"):
    out = ["[convolutional]"]
    if bn:
        out.append("batch_normalize=1")
    out += [f"filters={filters}", f"size={size}", f"stride={stride}", "pad=1",
            f"activation={act}", ""]
    return out


def _shortcut(frm=-3):
    return ["[shortcut]", f"from={frm}", "activation=linear", ""]


def _route(*layers):
    return ["[route]", "layers = " + ", ".join(str(l) for l in layers), ""]


def _upsample():
    return ["[upsample]", "stride=2", ""]


def _maxpool(size, stride):
    return ["[maxpool]", f"size={size}", f"stride={stride}", ""]


def _yolo(mask, anchors, num, classes=80):
    return ["[yolo]", "mask = " + ",".join(str(m) for m in mask), f"anchors = {anchors}",
            f"classes={classes}", f"num={num}", "jitter=.3", "ignore_thresh = .5",
            "truth_thresh = 1", "random=1", ""]


def yolov3_cfg(height=416, width=416, classes=80) -> str:
    """Darknet-53 backbone + 3 YOLO heads: 107 layer blocks."""
    L = _net(height, width)
    L += _conv(32, 3, 1)
    ch = 32
    for n_res in (1, 2, 8, 8, 4):
        ch *= 2
        L += _conv(ch, 3, 2)                       # downsample
        for _ in range(n_res):
            L += _conv(ch // 2, 1, 1) + _conv(ch, 3, 1) + _shortcut(-3)
    nout = 3 * (5 + classes)

    def neck(c):
        out = []
        for _ in range(3):
            out += _conv(c, 1, 1) + _conv(2 * c, 3, 1)
        return out

    L += neck(512) + _conv(nout, 1, 1, bn=False, act="linear") + _yolo((6, 7, 8), _ANCHORS_V3, 9, classes)
    L += _route(-4) + _conv(256, 1, 1) + _upsample() + _route(-1, 61)
    L += neck(256) + _conv(nout, 1, 1, bn=False, act="linear") + _yolo((3, 4, 5), _ANCHORS_V3, 9, classes)
    L += _route(-4) + _conv(128, 1, 1) + _upsample() + _route(-1, 36)
    L += neck(128) + _conv(nout, 1, 1, bn=False, act="linear") + _yolo((0, 1, 2), _ANCHORS_V3, 9, classes)
    return "\n".join(L) + "\n"


def yolov3_tiny_cfg(height=416, width=416, classes=80) -> str:
    """YOLOv3-tiny: 24 layer blocks."""
    nout = 3 * (5 + classes)
    L = _net(height, width)
    for f in (16, 32, 64, 128, 256):
        L += _conv(f, 3, 1) + _maxpool(2, 2)
    L += _conv(512, 3, 1) + _maxpool(2, 1)
    L += _conv(1024, 3, 1) + _conv(256, 1, 1) + _conv(512, 3, 1)
    L += _conv(nout, 1, 1, bn=False, act="linear") + _yolo((3, 4, 5), _ANCHORS_TINY, 6, classes)
    L += _route(-4) + _conv(128, 1, 1) + _upsample() + _route(-1, 8)
    L += _conv(256, 3, 1) + _conv(nout, 1, 1, bn=False, act="linear") + _yolo((0, 1, 2), _ANCHORS_TINY, 6, classes)
    return "\n".join(L) + "\n"


def mini_cfg(height=64, width=64, classes=80) -> str:
    """Small residual network in the Darknet cfg grammar for tests: 24 layer blocks, every conv after the stem has
    Cin % 32 == 0 (expressible in the split-f16 format), heads at stride 32 and 16.  At 64x64 the first head is a 2x2
    grid: fewer cells than the decode epilogue's row step, ragged M tails on every tile shape."""
    nout = 3 * (5 + classes)
    L = _net(height, width)
    L += _conv(32, 3, 1) + _conv(64, 3, 2)
    L += _conv(32, 1, 1) + _conv(64, 3, 1) + _shortcut(-3)
    L += _conv(128, 3, 2)
    L += _conv(64, 1, 1) + _conv(128, 3, 1) + _shortcut(-3)
    L += _conv(256, 3, 2) + _conv(512, 3, 2) + _conv(1024, 3, 2)
    L += _conv(512, 1, 1) + _conv(1024, 3, 1)
    L += _conv(nout, 1, 1, bn=False, act="linear") + _yolo((6, 7, 8), _ANCHORS_V3, 9, classes)
    L += _route(-4) + _conv(256, 1, 1) + _upsample() + _route(-1, 10)
    L += _conv(256, 1, 1) + _conv(512, 3, 1)
    L += _conv(nout, 1, 1, bn=False, act="linear") + _yolo((3, 4, 5), _ANCHORS_V3, 9, classes)
    return "\n".join(L) + "\n"


def mini_fallback_cfg(height=64, width=64, classes=3) -> str:
    """Legal cfg whose graph defeats every fusion of the planner, so the stand-alone kernels run (exact-fp32 plans):
    a shortcut whose producer is an upsample (add kernel), a second concat of an already-placed layer (copy kernel),
    a head conv that a later route also reads (stand-alone predict_transform), a max-pool of each kind."""
    nout = 3 * (5 + classes)
    L = _net(height, width)
    L += _conv(32, 3, 1) + _conv(64, 3, 2)            # 0, 1: 64 @ 32x32
    L += _conv(64, 3, 2)                               # 2: 64 @ 16x16
    L += _conv(128, 3, 2) + _conv(64, 1, 1)            # 3, 4: 64 @ 8x8
    L += _upsample()                                   # 5: 64 @ 16x16
    L += _shortcut(-4)                                 # 6: upsample + layer 2 -> stand-alone add
    L += _route(6, 2)                                  # 7: 128 @ 16x16, zero-copy
    L += _conv(64, 1, 1)                               # 8
    L += _route(8, 2)                                  # 9: layer 2 already lives in route 7's buffer -> copy kernel
    L += _maxpool(2, 2) + _maxpool(2, 1)               # 10, 11: 128 @ 8x8
    L += _conv(nout, 1, 1, bn=False, act="linear")     # 12: head conv, also read by route 14
    L += _yolo((6, 7, 8), _ANCHORS_V3, 9, classes)     # 13: stand-alone decode (8x8 grid, stride 8)
    L += _route(-2)                                    # 14: the raw head conv output again
    L += _conv(32, 3, 2)                               # 15: 32 @ 4x4
    L += _conv(nout, 1, 1, bn=False, act="linear") + _yolo((3, 4, 5), _ANCHORS_V3, 9, classes)   # 16, 17: fused decode, 4x4 grid
    return "\n".join(L) + "\n"


def v5_style_mini_cfg(height=128, width=128, classes=80, act="silu") -> str:
    """YOLOv5-style building blocks in the (extended) cfg grammar — NOT a reference network.  The reference's YOLOv5 path is a
    torch.hub fetch (detect.py:255-285) whose model source does not exist offline, so no YOLOv5 graph can be pinned; this cfg
    exercises the kernel-level pieces such a graph needs, each checked against PyTorch's own CPU op (parity unpinned):
    a 6x6 stride-2 pad-2 stem, SiLU activations (also on a residual block and on the hosted / fused epilogues), an SPPF-like
    run of 5x5 stride-1 pad-2 max-pools joined by routes, and nearest x2 upsampling.  Extension keys: activation=silu,
    [maxpool] symmetric=1, [upsample] mode=nearest.  Every conv after the stem has Cin % 32 == 0, so the graph is expressible in
    the split-f16 format as well (SiLU there: the kernels with the LDS-transposed epilogue, i.e. generic and band tiles)."""
    nout = 3 * (5 + classes)
    L = _net(height, width)
    L += _conv(32, 6, 2, act=act)                                    # 0: 6x6 / 2 stem (pad = (6 - 1) // 2 = 2), 32 @ H/2
    L += _conv(64, 3, 2, act=act)                                    # 1: 64 @ H/4
    L += _conv(32, 1, 1, act=act) + _conv(64, 3, 1, act=act) + _shortcut(-3)      # 2-4: bottleneck with shortcut
    L += _conv(128, 3, 2, act=act)                                   # 5: 128 @ H/8
    L += _conv(256, 3, 2, act=act)                                   # 6: 256 @ H/16
    L += _conv(128, 1, 1, act=act)                                   # 7: SPPF entry, 128 @ H/16
    L += ["[maxpool]", "size=5", "stride=1", "symmetric=1", ""]         # 8
    L += _route(-1, -2)                                                 # 9: pooled + entry, 256 ch
    L += ["[maxpool]", "size=5", "stride=1", "symmetric=1", ""]         # 10: pool of the concat (9x9 receptive field on half of it)
    L += _conv(256, 1, 1, act=act)                                   # 11
    L += _conv(nout, 1, 1, bn=False, act="linear") + _yolo((6, 7, 8), _ANCHORS_V3, 9, classes)   # 12, 13: head at stride 16
    L += _route(-3) + _conv(128, 1, 1, act=act)                      # 14, 15
    L += ["[upsample]", "stride=2", "mode=nearest", ""]                 # 16: 128 @ H/8
    L += _route(-1, 5)                                                  # 17: 256 @ H/8
    L += _conv(128, 1, 1, act=act) + _conv(256, 3, 1, act=act)    # 18, 19
    L += _conv(nout, 1, 1, bn=False, act="linear") + _yolo((3, 4, 5), _ANCHORS_V3, 9, classes)   # 20, 21: head at stride 8
    return "\n".join(L) + "\n"


def yolov5s_style_cfg(height=640, width=640, classes=80) -> str:
    """A YOLOv5s-SHAPED graph in the extended cfg grammar, written from the PUBLISHED architecture description of YOLOv5 v6.0
    (yolov5s: width 0.50, depth 0.33 — CSP backbone of C3 blocks, SPPF, PANet neck, three Detect heads) — NOT from its source,
    which the reference only reaches through torch.hub (detect.py:255-285) and which does not exist offline.  It is a workload of
    BASELINE config (5)'s shape for the kernels (synthetic weights, parity unpinned), nothing more.
        Conv        = [convolutional] batch_normalize=1, activation=silu
        C3(c, n, s) = cv1 1x1 c/2 -> n x (1x1, 3x3 [, + shortcut]); cv2 1x1 c/2 on the block input; concat; cv3 1x1 c
        SPPF        = 1x1 c/2; three chained 5x5 stride-1 symmetric max-pools; concat of all four; 1x1 c
        Detect      = 1x1 linear conv to 3 * (5 + classes) per scale, [yolo] decode=v5
    Extension keys used: activation=silu, [maxpool] symmetric=1, [upsample] mode=nearest, [yolo] decode=v5, 4-source [route]."""
    nout = 3 * (5 + classes)
    L = _net(height, width)
    n = [0]                                            # running layer index

    def emit(lines):
        L.extend(lines)
        n[0] += 1
        return n[0] - 1

    def conv(c, k, s):
        return emit(_conv(c, k, s, act="silu"))

    def route(*idx):
        return emit(_route(*idx))

    def c3(x, c, reps, shortcut):
        h = c // 2
        if x != n[0] - 1:
            route(x)
        m = conv(h, 1, 1)                              # cv1
        for _ in range(reps):
            conv(h, 1, 1)
            m2 = conv(h, 3, 1)
            m = emit(_shortcut(-3)) if shortcut else m2
        route(x)
        b = conv(h, 1, 1)                              # cv2 on the block input
        route(m, b)
        return conv(c, 1, 1)                           # cv3

    conv(32, 6, 2)                                     # 0  P1/2
    x = conv(64, 3, 2)                                 #    P2/4
    x = c3(x, 64, 1, True)
    x = conv(128, 3, 2)                                #    P3/8
    p3 = c3(x, 128, 2, True)
    x = conv(256, 3, 2)                                #    P4/16
    p4 = c3(x, 256, 3, True)
    x = conv(512, 3, 2)                                #    P5/32
    x = c3(x, 512, 1, True)
    s0 = conv(256, 1, 1)                               # SPPF
    s1 = emit(["[maxpool]", "size=5", "stride=1", "symmetric=1", ""])
    s2 = emit(["[maxpool]", "size=5", "stride=1", "symmetric=1", ""])
    s3 = emit(["[maxpool]", "size=5", "stride=1", "symmetric=1", ""])
    route(s0, s1, s2, s3)
    x = conv(512, 1, 1)
    h10 = conv(256, 1, 1)                              # neck, top-down
    emit(["[upsample]", "stride=2", "mode=nearest", ""])
    route(n[0] - 1, p4)
    x = c3(n[0] - 1, 256, 1, False)
    h14 = conv(128, 1, 1)
    emit(["[upsample]", "stride=2", "mode=nearest", ""])
    route(n[0] - 1, p3)
    o3 = c3(n[0] - 1, 128, 1, False)                   # P3/8 output
    conv(128, 3, 2)                                    # bottom-up
    route(n[0] - 1, h14)
    o4 = c3(n[0] - 1, 256, 1, False)                   # P4/16 output
    conv(256, 3, 2)
    route(n[0] - 1, h10)
    o5 = c3(n[0] - 1, 512, 1, False)                   # P5/32 output
    for src, mask in ((o3, (0, 1, 2)), (o4, (3, 4, 5)), (o5, (6, 7, 8))):
        if src != n[0] - 1:
            route(src)
        emit(_conv(nout, 1, 1, bn=False, act="linear"))
        y = _yolo(mask, _ANCHORS_V3, 9, classes)
        emit(y[:-1] + ["decode=v5", ""])
    return "\n".join(L) + "\n"


def write_cfg(path, text):
    with open(path, "w") as f:
        f.write(text)
    return path


SHIPPED = {"yolov3.cfg": yolov3_cfg, "yolov3-tiny.cfg": yolov3_tiny_cfg}


def ensure_cfg_files(directory):
    """Write ``yolov3.cfg`` / ``yolov3-tiny.cfg`` (the two networks the reference ships under cfg/) into ``directory``
    when they are missing; ``__graft_entry__.build()`` does this for ``<repo>/cfg`` so ``params.json`` resolves."""
    import os
    os.makedirs(directory, exist_ok=True)
    out = []
    for name, gen in SHIPPED.items():
        p = os.path.join(directory, name)
        if not os.path.exists(p):
            write_cfg(p, gen())
        out.append(p)
    return out
