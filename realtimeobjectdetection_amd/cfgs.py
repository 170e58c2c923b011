"""Generators for the Darknet ``.cfg`` texts of the two networks the reference ships.

The reference reads ``cfg/yolov3.cfg`` and ``cfg/yolov3-tiny.cfg`` (reference:
src/darknet.py:412-447 parses them, detect.py:185 picks the path from params.json).  Those
files do not travel to the GPU box, so this module *generates* equivalent cfg text from a
compact architecture description.  ``tests/test_cfg.py`` checks (in the build container,
where /root/reference is mounted) that parsing the generated text yields layer blocks
identical to parsing the reference's files.

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


def write_cfg(path, text):
    with open(path, "w") as f:
        f.write(text)
    return path
