"""Darknet cfg grammar + layer IR (host side).

Behavioural contract (reference: src/darknet.py:412-447 ``parse_cfg``; 449-603
``create_modules``):

* split on newlines, drop empty lines, drop lines whose *first* character is ``#`` (tested
  before stripping), strip both ends;
* ``[name]`` opens a block whose ``type`` is ``name``; any other line is ``key=value`` split on
  the single ``=``, key right-stripped, value left-stripped; values stay strings;
* block 0 is ``[net]``.

``build_ir`` restates the channel/shape bookkeeping of ``create_modules`` and the shape flow of
``Darknet.forward`` (src/darknet.py:199-253) as a flat list of ``Layer`` records.  The C++ plan
(csrc/plan.cpp) holds the same IR natively; tests compare the two.
"""
from dataclasses import dataclass, field
from typing import List, Tuple


def parse_cfg_text(text: str) -> List[dict]:
    lines = text.split("\n")
    lines = [x for x in lines if len(x) > 0]
    lines = [x for x in lines if x[0] != "#"]
    lines = [x.strip() for x in lines]
    blocks, block = [], {}
    for line in lines:
        if not line:
            # a whitespace-only line survives the reference's filters and would raise there
            # (IndexError on line[0]); treat it as blank instead.
            continue
        if line[0] == "[":
            if block:
                blocks.append(block)
                block = {}
            block["type"] = line[1:-1].rstrip()
        else:
            key, value = line.split("=")
            block[key.rstrip()] = value.lstrip()
    blocks.append(block)
    return blocks


def parse_cfg(path: str) -> List[dict]:
    with open(path, "r") as f:
        return parse_cfg_text(f.read())


@dataclass
class Layer:
    index: int
    type: str                      # convolutional|shortcut|route|upsample|maxpool|yolo
    cin: int = 0
    cout: int = 0
    hin: int = 0
    win: int = 0
    hout: int = 0
    wout: int = 0
    size: int = 0                  # conv / maxpool kernel
    stride: int = 1
    pad: int = 0
    bn: bool = False
    leaky: bool = False
    silu: bool = False             # extension (not in the reference grammar): activation=silu, x * sigmoid(x)
    nearest: bool = False          # extension: [upsample] mode=nearest (the reference always builds bilinear, darknet.py:589)
    decode_v5: bool = False        # extension: [yolo] decode=v5 (YOLOv5-style head arithmetic)
    pool_pad: int = 0              # extension: [maxpool] symmetric=1 -> (size-1)//2 of -inf padding on every side (SPPF pools)
    srcs: Tuple[int, ...] = ()     # absolute source layer indices (route / shortcut)
    anchors: Tuple[Tuple[int, int], ...] = ()
    classes: int = 0
    row_offset: int = 0            # yolo: first output row of this head
    rows: int = 0                  # yolo: rows produced

    @property
    def flops(self) -> int:
        """2*MACs of the direct convolution (SURVEY.md §8 d)."""
        if self.type != "convolutional":
            return 0
        return 2 * self.hout * self.wout * self.cout * self.cin * self.size * self.size


@dataclass
class NetIR:
    height: int
    width: int
    layers: List[Layer] = field(default_factory=list)
    total_rows: int = 0
    attrs: int = 0                 # 5 + classes (0 if no yolo layer)

    @property
    def conv_flops(self) -> int:
        return sum(l.flops for l in self.layers)

    @property
    def n_weights(self) -> int:
        n = 0
        for l in self.layers:
            if l.type == "convolutional":
                n += (4 * l.cout if l.bn else l.cout) + l.cout * l.cin * l.size * l.size
        return n


def build_ir(blocks: List[dict], height: int, width: int = None) -> NetIR:
    """Shape-resolved layer list for input ``[B,3,height,width]``."""
    if width is None:
        width = height
    ir = NetIR(height=height, width=width)
    prev_c, prev_h, prev_w = 3, height, width
    shapes = []  # (c, h, w) per layer output
    row_off = 0
    for i, b in enumerate(blocks[1:]):
        t = b["type"]
        L = Layer(index=i, type=t, cin=prev_c, hin=prev_h, win=prev_w)
        if t == "convolutional":
            try:
                bn = bool(int(b["batch_normalize"]))
            except (ValueError, KeyError):
                bn = False
            L.bn = bn
            L.cout = int(b["filters"])
            L.size = int(b["size"])
            L.stride = int(b["stride"])
            L.pad = (L.size - 1) // 2 if int(b["pad"]) else 0
            L.leaky = b["activation"] == "leaky"
            L.silu = b["activation"] in ("silu", "swish")
            L.hout = (prev_h + 2 * L.pad - L.size) // L.stride + 1
            L.wout = (prev_w + 2 * L.pad - L.size) // L.stride + 1
        elif t == "upsample":
            L.cout, L.hout, L.wout, L.stride = prev_c, prev_h * 2, prev_w * 2, 2
            L.nearest = b.get("mode", "bilinear") == "nearest"
        elif t == "maxpool":
            L.size, L.stride = int(b["size"]), int(b["stride"])
            L.cout = prev_c
            if int(b.get("symmetric", 0)):
                L.pool_pad = (L.size - 1) // 2
                L.hout = (prev_h + 2 * L.pool_pad - L.size) // L.stride + 1
                L.wout = (prev_w + 2 * L.pool_pad - L.size) // L.stride + 1
            elif L.stride != 1:
                L.hout = (prev_h - L.size) // L.stride + 1
                L.wout = (prev_w - L.size) // L.stride + 1
            else:  # MaxPoolStride1: replicate-pad right/bottom by size-1, then size/1 pool
                L.hout, L.wout = prev_h, prev_w
        elif t == "shortcut":
            src = i + int(b["from"])
            L.srcs = (i - 1, src)
            L.cout, L.hout, L.wout = shapes[i - 1]
        elif t == "route":
            lay = b["layers"]
            if isinstance(lay, str):
                lay = lay.split(",")
            srcs = []
            for a in lay:
                a = int(a)
                srcs.append(a if a > 0 else i + a)
            L.srcs = tuple(srcs)
            L.cout = sum(shapes[s][0] for s in srcs)
            L.hout, L.wout = shapes[srcs[0]][1], shapes[srcs[0]][2]
            L.cin = L.cout
        elif t == "yolo":
            mask = [int(x) for x in b["mask"].split(",")]
            a = [int(x) for x in b["anchors"].split(",")]
            pairs = [(a[j], a[j + 1]) for j in range(0, len(a), 2)]
            L.anchors = tuple(pairs[m] for m in mask)
            L.classes = int(b["classes"])
            L.decode_v5 = b.get("decode", "") == "v5"
            L.cout, L.hout, L.wout = prev_c, prev_h, prev_w
            L.row_offset = row_off
            L.rows = prev_h * prev_w * len(L.anchors)
            row_off += L.rows
            ir.attrs = 5 + L.classes
        else:
            raise AssertionError("Unknown block error: A unknown block is provided: %r" % t)
        ir.layers.append(L)
        shapes.append((L.cout, L.hout, L.wout))
        if t != "yolo":
            prev_c, prev_h, prev_w = L.cout, L.hout, L.wout
        else:
            # reference: outputs[i] = outputs[i-1]; x becomes the decoded tensor but the next
            # block in every shipped cfg is a route, so x is never consumed (darknet.py:247)
            prev_c, prev_h, prev_w = shapes[i - 1]
            shapes[i] = shapes[i - 1]
    ir.total_rows = row_off
    return ir
