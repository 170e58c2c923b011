"""Host-side mirror of the reference's ``Darknet`` class, backed by librtod.so.

Drop-in surface (reference: src/darknet.py:138-603, SURVEY.md §8 b): ``Darknet(cfg_file_path, CUDA)``,
``.blocks`` / ``.net_info`` (mutable; callers set ``net_info["height"]``) / ``.module_list`` /
``.header`` / ``.seen`` / ``.CUDA`` / ``.TRAIN``, ``get_blocks()``, ``get_module_list()``,
``load_weights(path)``, ``load_state_dict`` with the reference's key names
(``module_list.{i}.conv_{i}.weight`` ...), ``forward(x) -> [B,N,5+C]``, ``train_mode()``, and after
the first forward ``.anchors`` / ``.num_classes``.

Everything numeric happens in hand-written HIP kernels behind the C ABI (include/rtod.h).  The
``nn.Module`` tree only *holds* the parameters in the reference's layout so ``state_dict`` /
``.parameters()`` / ``.cuda()`` keep working; PyTorch executes none of the network.  There is no
CPU fallback: a non-CUDA input or a missing librtod.so raises.

BatchNorm follows the module's mode like ``nn.BatchNorm2d``: ``.eval()`` = running statistics folded into the
convolutions — the canonical, fast, frame-independent mode of SURVEY.md F2.  The reference's callers never call
``.eval()``, so there BN uses batch statistics (frames of a batch influence each other); a model left in training
mode runs exactly that on the exact-fp32 kernels (a slow parity path, with a one-time warning) unless
``bn_running_stats_in_train`` asks for eval semantics regardless of the mode.
"""
import ctypes as C
import json
import os
import warnings
import weakref
from contextlib import contextmanager

import numpy as np
import torch
import torch.nn as nn

from . import _ffi
from .cfg import parse_cfg, build_ir


_OVERFLOW_MSG = ("Darknet (precision f16s3): an activation reached the split-f16 range limit (|x| >= 8188) and was saturated; "
                 "the results of that forward are not valid. Use precision='fp32' (exact MFMA kernels) for these weights.")


def take_pending_overflow(prediction):
    """``(model, flag tensor)`` of the forward that produced ``prediction`` (or ``(None, None)``): util.write_results reads the
    flag in the same host synchronisation as its detection counts instead of paying a second round trip.  The link is a tag on
    the output tensor OBJECT itself — (weak reference to the model, its forward sequence number), set by Darknet.forward —
    not the tensor's address: a freed-and-reused address can no longer attach a stale model to another tensor."""
    tag = getattr(prediction, "_rtod_forward", None)
    if tag is None:
        return None, None
    prediction._rtod_forward = None                            # consumed: a second write_results on the same tensor reads nothing
    model = tag[0]()
    if model is None or model._ovf is None:
        return None, None
    return model, model._ovf


def raise_overflow(model):
    model._ovf.zero_()
    raise FloatingPointError(_OVERFLOW_MSG)


class EmptyLayer(nn.Module):
    """Placeholder for route / shortcut blocks (reference: src/darknet.py:49-54)."""


class DetectionLayer(nn.Module):
    """Holds the masked anchors of a yolo block (reference: src/darknet.py:57-97)."""

    def __init__(self, anchors, CUDA=False):
        super().__init__()
        self.anchors = anchors
        self.CUDA = CUDA


class MaxPoolStride1(nn.Module):
    """Marker for ``[maxpool] stride=1`` (reference: src/darknet.py:17-46); executed by librtod."""

    def __init__(self, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.pad = kernel_size - 1


def _blocks_to_cfg_text(blocks) -> str:
    """Serialise parsed blocks back to cfg text for the native parser (route ``layers`` may have
    been split into a list by create_modules, as in the reference, src/darknet.py:564)."""
    out = []
    for b in blocks:
        out.append("[%s]" % b["type"])
        for k, v in b.items():
            if k == "type":
                continue
            if isinstance(v, (list, tuple)):
                v = ",".join(str(a) for a in v)
            out.append("%s=%s" % (k, v))
        out.append("")
    return "\n".join(out) + "\n"


class Darknet(nn.Module):
    def __init__(self, cfg_file_path, CUDA):
        super().__init__()
        self.blocks = self.parse_cfg(cfg_file_path)
        self.net_info, self.module_list = self.create_modules(self.blocks)
        self.header = torch.IntTensor([0, 0, 0, 0])
        self.seen = 0
        self.CUDA = CUDA
        self.TRAIN = False
        self.bn_running_stats_in_train = False
        self._warned_batch_bn = False
        self.update_running_stats = True   # training-mode forward updates running_mean / running_var / num_batches_tracked like torch does
        self.precision = os.environ.get("RTOD_PRECISION", "auto")   # "fp32" (exact MFMA) | "f16s3" (split f16, 3 products) | "auto"
        self.keep_all_layers = False      # debug: no activation-arena reuse (read_layer after forward)
        self.autotune = True              # split-f16 plans: measure the tile variants once per batch size (rtod_plan_autotune)
        self.options = {}                 # rtod_plan_set_option name -> int (fusion / kernel-selection switches, tests and A/B runs)
        # split-f16 range guard (|activation| < 8188): producers saturate and raise a device flag.  "write_results": the flag
        # is read at write_results' existing host sync; "forward": read (one host sync) after every forward, and with
        # precision "auto" the plan falls back to the exact-fp32 kernels and re-runs; "off": never read.
        self.overflow_check = "write_results"
        self._ovf = None
        self._forward_seq = 0             # forwards whose range flag is still to be read (tag on the output tensor, see take_pending_overflow)
        self._tuned = set()
        self._cfg_text = _blocks_to_cfg_text(self.blocks)
        self._plan = None
        self._plan_key = None
        self._weights_version = 0
        self._stats_version = 0             # running_mean / running_var updates of training-mode forwards (eval plans fold them)
        self._plan_weights_version = -1
        self._info = None

    # ------------------------------------------------------------------ reference getters
    def get_blocks(self) -> list:
        return self.blocks

    def get_module_list(self) -> nn.ModuleList:
        return self.module_list

    @staticmethod
    def parse_cfg(cfg_file_path):
        return parse_cfg(cfg_file_path)

    @staticmethod
    def create_modules(blocks):
        """Same module tree / parameter names as the reference builds (src/darknet.py:449-603) so
        state_dicts are interchangeable; shapes come from the shared IR."""
        net_info = blocks[0]
        ir = build_ir(blocks, int(net_info.get("height", 416)))
        module_list = nn.ModuleList()
        for L, blk in zip(ir.layers, blocks[1:]):
            i = L.index
            m = nn.Sequential()
            if L.type == "convolutional":
                m.add_module("conv_%d" % i, nn.Conv2d(L.cin, L.cout, L.size, L.stride, L.pad, bias=not L.bn))
                if L.bn:
                    m.add_module("batch_norm_%d" % i, nn.BatchNorm2d(L.cout))
                if L.leaky:
                    m.add_module("leaky_%d" % i, nn.LeakyReLU(0.1, inplace=True))
                elif L.silu:                                   # cfg extension (activation=silu)
                    m.add_module("silu_%d" % i, nn.SiLU(inplace=True))
            elif L.type == "upsample":
                if L.nearest:                                  # cfg extension (mode=nearest)
                    m.add_module("upsample_%d" % i, nn.Upsample(scale_factor=2, mode="nearest"))
                else:
                    m.add_module("upsample_%d" % i, nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
            elif L.type == "route":
                if isinstance(blk["layers"], str):
                    blk["layers"] = blk["layers"].split(",")        # the reference splits in place
                m.add_module("route_%d" % i, EmptyLayer())
            elif L.type == "shortcut":
                m.add_module("shortcut_%d" % i, EmptyLayer())
            elif L.type == "maxpool":
                if L.pool_pad:                                 # cfg extension (symmetric=1)
                    m.add_module("maxpool_%d" % i, nn.MaxPool2d(L.size, L.stride, L.pool_pad))
                else:
                    m.add_module("maxpool_%d" % i, nn.MaxPool2d(L.size, L.stride) if L.stride != 1 else MaxPoolStride1(L.size))
            elif L.type == "yolo":
                m.add_module("Detection_%d" % i, DetectionLayer([tuple(a) for a in L.anchors]))
            module_list.append(m)
        return net_info, module_list

    # ------------------------------------------------------------------ weights
    def configure_weights(self, weight_file_path):
        with open(weight_file_path, "rb") as fp:
            header = np.fromfile(fp, dtype=np.int32, count=5)
            weights = np.fromfile(fp, dtype=np.float32)
        self.header = torch.from_numpy(header)
        self.seen = self.header[3]
        return weights

    def load_weights(self, weight_file_path: str):
        """Darknet binary -> module tensors (reference: src/darknet.py:316-410)."""
        w = self.configure_weights(weight_file_path)
        self.load_weight_stream(w)

    def load_weight_stream(self, w: np.ndarray):
        ptr = 0
        with torch.no_grad():
            for m in self.module_list:
                conv = bn = None
                for name, sub in m.named_children():
                    if name.startswith("conv_"):
                        conv = sub
                    elif name.startswith("batch_norm_"):
                        bn = sub
                if conv is None:
                    continue
                c = conv.out_channels
                if bn is not None:
                    for t in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                        t.copy_(torch.from_numpy(w[ptr:ptr + c]).view_as(t)); ptr += c
                else:
                    conv.bias.copy_(torch.from_numpy(w[ptr:ptr + c]).view_as(conv.bias)); ptr += c
                n = conv.weight.numel()
                conv.weight.copy_(torch.from_numpy(w[ptr:ptr + n]).view_as(conv.weight)); ptr += n
        self._weights_version += 1
        return ptr

    def weight_stream(self) -> np.ndarray:
        """Current parameters as a ``.weights`` float payload (host, float32)."""
        parts = []
        for m in self.module_list:
            conv = bn = None
            for name, sub in m.named_children():
                if name.startswith("conv_"):
                    conv = sub
                elif name.startswith("batch_norm_"):
                    bn = sub
            if conv is None:
                continue
            if bn is not None:
                parts += [bn.bias, bn.weight, bn.running_mean, bn.running_var]
            else:
                parts.append(conv.bias)
            parts.append(conv.weight)
        return np.concatenate([p.detach().float().cpu().numpy().reshape(-1) for p in parts]).astype(np.float32, copy=False)

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self._weights_version += 1
        return r

    def invalidate_weights(self):
        """Call after mutating parameters in place so the next forward re-packs them."""
        self._weights_version += 1

    # ------------------------------------------------------------------ plan
    def _destroy_plan(self):
        if self._plan is not None:
            _ffi.lib().rtod_plan_destroy(self._plan)
            self._plan = None
            self._plan_key = None

    def __del__(self):
        try:
            self._destroy_plan()
        except Exception:
            pass

    def prepare(self, max_batch: int, device=None):
        """Build the native plan for ``net_info['height']`` and ``max_batch`` frames and upload the
        BN-folded, K-major packed weights.  Called lazily by ``forward``."""
        lib = _ffi.lib()
        if device is None:
            device = torch.cuda.current_device()
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        inp_dim = int(self.net_info["height"])
        if self.precision not in ("fp32", "f16s3", "auto"):
            raise ValueError("Darknet.precision must be 'fp32', 'f16s3' or 'auto'")
        # BatchNorm semantics follow the module's mode like nn.BatchNorm2d: eval() = running statistics, folded into the convs
        # (the fast path, frame-independent); training mode = the statistics of the batch — what the reference's callers
        # actually run, since they never call .eval() (detect.py:185-194).  That mode is a parity path on the
        # exact-fp32 kernels (conv -> per-channel statistics -> normalise), several times slower and batch-dependent.
        batch_bn = bool(self.training and not self.bn_running_stats_in_train)
        if batch_bn and self.precision == "f16s3":
            raise RuntimeError("Darknet is in training mode (batch-statistics BatchNorm, like the reference without .eval()): that "
                               "path runs on the exact-fp32 kernels only; call .eval() for the split-f16 kernels or set precision='auto'")
        if batch_bn and not self._warned_batch_bn:
            warnings.warn("Darknet is in training mode: BatchNorm uses the statistics of the batch, as the reference does when its "
                          "callers skip .eval() (slow parity path, results depend on the batch); call .eval() for the fast, "
                          "frame-independent path", RuntimeWarning)
            self._warned_batch_bn = True
        opts = dict(self.options)
        if batch_bn:
            opts["bn_batch_stats"] = 1
        key = (inp_dim, int(max_batch), device.index, bool(self.keep_all_layers), "fp32" if batch_bn else self.precision, tuple(sorted(opts.items())))
        if self._plan is None or self._plan_key != key:
            self._destroy_plan()
            h = C.c_void_p()
            txt = self._cfg_text.encode()
            _ffi.check(lib.rtod_plan_create(txt, len(txt), inp_dim, inp_dim, int(max_batch), device.index, C.byref(h)))
            self._plan, self._plan_key = h, key
            self._tuned = set()
            if self.keep_all_layers:
                _ffi.check(lib.rtod_plan_set_keep_all_layers(self._plan, 1))
            for name, value in sorted(opts.items()):
                _ffi.check(lib.rtod_plan_set_option(self._plan, name.encode(), int(value)))
            if self._ovf is None or self._ovf.device != device:
                self._ovf = torch.zeros(1, dtype=torch.int32, device=device)
            _ffi.check(lib.rtod_plan_set_overflow_flag(self._plan, C.c_void_p(self._ovf.data_ptr())))
            # "auto": the split-f16 kernels when the cfg supports them (yolov3 does; cfgs with maxpool or
            # Cin % 32 != 0 such as yolov3-tiny do not), else the exact-fp32 MFMA kernels.  Both are HIP paths.
            if self.precision == "fp32" or batch_bn:
                self.active_precision = "fp32"
            else:
                rc = lib.rtod_plan_set_precision(self._plan, 1)
                if rc == 0:
                    self.active_precision = "f16s3"
                elif self.precision == "auto" and rc == -3:
                    self.active_precision = "fp32"
                else:
                    _ffi.check(rc)
            if self.active_precision == "fp32":
                _ffi.check(lib.rtod_plan_set_precision(self._plan, 0))
            self._plan_weights_version = -1
            info = _ffi.PlanInfo()
            _ffi.check(lib.rtod_plan_get_info(self._plan, C.byref(info)))
            self._info = info
        version = (self._weights_version, 0 if batch_bn else self._stats_version)
        if self._plan_weights_version != version:
            w = np.ascontiguousarray(self.weight_stream())
            with torch.cuda.device(device):
                _ffi.check(lib.rtod_plan_load_weights(self._plan, w.ctypes.data_as(C.c_void_p), w.size))
            self._plan_weights_version = version
        return self._info

    # ------------------------------------------------------------------ tile tables (autotune results)
    def get_tiles(self, batch: int) -> list:
        """Tile variant per launch that autotune chose for this batch size (rtod_plan_get_tiles)."""
        lib = _ffi.lib()
        n = lib.rtod_plan_get_tiles(self._plan, int(batch), None, 0)
        if n < 0:
            _ffi.check(n)
        arr = (C.c_int * n)()
        got = lib.rtod_plan_get_tiles(self._plan, int(batch), arr, n)
        if got < 0:
            _ffi.check(got)
        return list(arr)

    def set_tiles(self, batch: int, variants) -> None:
        """Install a tile table (e.g. saved by another process) for this batch size: forwards of that size then launch exactly
        those kernels and measure nothing.  The plan must exist (prepare / a forward at a larger or equal batch size)."""
        arr = (C.c_int * len(variants))(*[int(v) for v in variants])
        _ffi.check(_ffi.lib().rtod_plan_set_tiles(self._plan, int(batch), arr, len(variants)))
        self._tuned.add(int(batch))

    def plan_description(self) -> dict:
        lib = _ffi.lib()
        need = C.c_size_t()
        _ffi.check(lib.rtod_plan_describe(self._plan, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        _ffi.check(lib.rtod_plan_describe(self._plan, buf, need.value, None))
        return json.loads(buf.value.decode())

    def launch_infos(self):
        lib = _ffi.lib()
        out = []
        for i in range(self._info.n_launches):
            li = _ffi.LaunchInfo()
            _ffi.check(lib.rtod_plan_get_launch(self._plan, i, C.byref(li)))
            out.append(li)
        return out

    # ------------------------------------------------------------------ forward
    def _check_input(self, x):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError("Darknet.forward: input must be a CUDA (ROCm) tensor; this build has no CPU path")
        if x.dtype != torch.float32 or x.dim() != 4 or x.size(1) != 3:
            raise ValueError("Darknet.forward: expected float32 [B,3,H,W], got %s %s" % (x.dtype, tuple(x.shape)))
        inp_dim = int(self.net_info["height"])
        if x.size(2) != inp_dim or x.size(3) != inp_dim:
            raise ValueError("Darknet.forward: input is %dx%d but net_info['height'] = %d (set it like detect.py:47 does)"
                             % (x.size(2), x.size(3), inp_dim))
        return inp_dim

    def forward(self, x, _launch_ms=None):
        self._check_input(x)
        lib = _ffi.lib()
        B = x.size(0)
        max_batch = B if self._plan_key is None else max(B, self._plan_key[1])
        info = self.prepare(max_batch, x.device)
        x = x.contiguous()
        out = torch.empty((B, info.total_rows, info.attrs), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        with torch.cuda.device(x.device):
            _ffi.check(lib.rtod_plan_set_train_decode(self._plan, 1 if self.TRAIN else 0))
            if _launch_ms is None:
                if self.autotune and self.active_precision == "f16s3" and B not in self._tuned:
                    # first forward of a batch size: measures the tile variants (synchronises); rtod_forward itself never does
                    _ffi.check(lib.rtod_plan_autotune(self._plan, C.c_void_p(x.data_ptr()), B, C.c_void_p(out.data_ptr()), stream))
                    self._tuned.add(B)
                else:
                    _ffi.check(lib.rtod_forward(self._plan, C.c_void_p(x.data_ptr()), B, C.c_void_p(out.data_ptr()), stream))
            else:
                _ffi.check(lib.rtod_forward_timed(self._plan, C.c_void_p(x.data_ptr()), B, C.c_void_p(out.data_ptr()), stream, _launch_ms))
        if self.training and not self.bn_running_stats_in_train and self.update_running_stats and _launch_ms is None:
            self._update_running_stats(B, stream)
        # attributes the reference sets as a side effect of forward (src/darknet.py:239-243, 260)
        anchors = []
        for m, blk in zip(self.module_list, self.blocks[1:]):
            if blk["type"] == "yolo":
                anchors.extend(m[0].anchors)
                self.num_classes = int(blk["classes"])
        self.anchors = anchors
        if self.active_precision == "f16s3" and self.overflow_check != "off":
            if self.overflow_check == "forward":
                if self.overflowed():
                    if self.precision != "auto":
                        raise FloatingPointError(_OVERFLOW_MSG)
                    warnings.warn("Darknet: an activation left the split-f16 range (|x| >= 8188); precision 'auto' falls back to the "
                                  "exact-fp32 MFMA kernels for this model", RuntimeWarning)
                    self.precision = "fp32"
                    return self.forward(x)
            else:
                self._forward_seq += 1
                out._rtod_forward = (weakref.ref(self), self._forward_seq)
        return out

    def _update_running_stats(self, batch, stream):
        """Training-mode side effect of nn.BatchNorm2d (momentum 0.1): running_mean / running_var move towards the batch's mean
        and UNBIASED variance, num_batches_tracked counts up — the reference mutates them on every forward.
        One kernel launch for all BatchNorm layers (rtod_plan_bn_update_running), no host round trip: the module buffers are
        updated in place on the device, behind the forward on the same stream."""
        lib = _ffi.lib()
        bns = [sub for m in self.module_list for sub in m.children() if isinstance(sub, nn.BatchNorm2d)]
        if not bns:
            return
        dev = torch.device("cuda", self._plan_key[2])
        mom = {bn.momentum if bn.momentum is not None else 0.1 for bn in bns}
        if len(mom) != 1:
            raise RuntimeError("Darknet: BatchNorm layers with different momenta are not supported")
        for bn in bns:                                             # the buffers live where the plan runs (the reference: model.cuda())
            for name in ("running_mean", "running_var", "num_batches_tracked"):
                t = getattr(bn, name)
                if t.device != dev:
                    setattr(bn, name, t.to(dev))
            if bn.running_mean.dtype != torch.float32 or not bn.running_mean.is_contiguous() or not bn.running_var.is_contiguous():
                raise RuntimeError("Darknet: BatchNorm running statistics must be contiguous float32 tensors")
        n = len(bns)
        rm = (C.c_void_p * n)(*[bn.running_mean.data_ptr() for bn in bns])
        rv = (C.c_void_p * n)(*[bn.running_var.data_ptr() for bn in bns])
        with torch.cuda.device(dev):
            _ffi.check(lib.rtod_plan_bn_update_running(self._plan, int(batch), rm, rv, n, float(mom.pop()), stream))
            torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
        self._stats_version += 1                # an eval-mode plan built later folds the updated statistics

    def overflowed(self) -> bool:
        """True when a split-f16 producer saturated since the last call (one small host sync); clears the flag."""
        if self._ovf is None:
            return False
        hit = bool(int(self._ovf.item()))
        if hit:
            self._ovf.zero_()
        return hit

    def check_overflow(self):
        if self.overflowed():
            raise FloatingPointError(_OVERFLOW_MSG)

    def make_graphed(self, example_x, post=None):
        """Capture ``forward`` (and optionally ``post(y)``, e.g. ``util.write_results_async``) for ``example_x``'s shape into a
        HIP graph and return ``run(x) -> (y, post result)``.  ``rtod_forward`` only enqueues, so the whole launch list replays
        as one graph launch: for small batches (YOLOv3-tiny batch 1: 24 launches of a few microseconds each) the per-launch
        host cost is what limits the latency.  ``run`` copies ``x`` into a static input buffer and returns the static output
        tensors: consume or clone them before the next call.  The range flag of split-f16 plans is not read (use
        ``overflowed()``)."""
        self._check_input(example_x)
        dev = example_x.device
        static_x = example_x.clone()
        keep = self.overflow_check
        self.overflow_check = "off"
        try:
            with torch.no_grad():
                y = self.forward(static_x)                                # builds the plan, autotunes this batch size
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):                             # warm-up on a non-default stream, as capture requires
                    y = self.forward(static_x)
                    r = post(y) if post is not None else None
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    y = self.forward(static_x)
                    r = post(y) if post is not None else None
        finally:
            self.overflow_check = keep

        def run(x):
            static_x.copy_(x, non_blocking=True)
            graph.replay()
            return y, r
        run.graph = graph
        return run

    def forward_timed(self, x):
        """Forward with a HIP-event pair around every launch; returns (out, ms per launch)."""
        self._check_input(x)
        info = self.prepare(x.size(0) if self._plan_key is None else max(x.size(0), self._plan_key[1]), x.device)
        ms = (C.c_float * info.n_launches)()
        out = self.forward(x, _launch_ms=ms)
        return out, np.frombuffer(ms, dtype=np.float32).copy()

    def read_layer(self, layer: int, batch: int) -> torch.Tensor:
        """Dense NCHW copy of a materialised layer output of the last forward (tests/debug)."""
        lib = _ffi.lib()
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        _ffi.check(lib.rtod_plan_layer_shape(self._plan, layer, C.byref(c), C.byref(h), C.byref(w)))
        dev = torch.device("cuda", self._plan_key[2])
        out = torch.empty((batch, c.value, h.value, w.value), dtype=torch.float32, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _ffi.check(lib.rtod_plan_read_layer(self._plan, layer, batch, C.c_void_p(out.data_ptr()), stream))
        return out

    @contextmanager
    def train_mode(self):
        """Decode without grid offsets / anchors (reference: src/darknet.py:305-314)."""
        try:
            self.TRAIN = True
            yield
        finally:
            self.TRAIN = False
