"""CPU tests of the boundary: librtod.so loads, exports every symbol include/rtod.h declares, and the
host-only plan logic (cfg grammar, shape resolution, buffer planning) agrees with the Python IR and
with the reference-derived fixtures.  No compute call is made (no GPU here)."""
import ctypes as C
import json
import os
import re

import pytest

from realtimeobjectdetection_amd import _ffi, cfgs
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(text, res, max_batch=8):
    lib = _ffi.lib()
    h = C.c_void_p()
    t = text.encode()
    rc = lib.rtod_plan_create(t, len(t), res, res, max_batch, 0, C.byref(h))
    return rc, h


def _describe(h):
    lib = _ffi.lib()
    need = C.c_size_t()
    assert lib.rtod_plan_describe(h, None, 0, C.byref(need)) == 0
    buf = C.create_string_buffer(need.value)
    assert lib.rtod_plan_describe(h, buf, need.value, None) == 0
    return json.loads(buf.value.decode())


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "rtod.h")).read()
    declared = set(re.findall(r"\b(rtod_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = C.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in rtod.h but not exported"
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    assert _ffi.lib().rtod_version() >= 100


@pytest.mark.parametrize("net,res", [("yolov3", 608), ("yolov3", 416), ("yolov3-tiny", 416)])
def test_native_plan_matches_python_ir(net, res):
    gen = {"yolov3": cfgs.yolov3_cfg, "yolov3-tiny": cfgs.yolov3_tiny_cfg}[net]
    rc, h = _plan(gen(), res)
    assert rc == 0, _ffi.last_error()
    d = _describe(h)
    ir = build_ir(parse_cfg_text(gen()), res)
    assert d["total_rows"] == ir.total_rows and d["attrs"] == ir.attrs
    assert d["n_weight_floats"] == ir.n_weights and d["conv_flops"] == ir.conv_flops
    for L, D in zip(ir.layers, d["layers"]):
        assert (L.type, L.cin, L.cout, L.hin, L.win, L.hout, L.wout, L.size, L.stride, L.pad, L.bn, L.leaky) == \
               (D["type"], D["cin"], D["cout"], D["hin"], D["win"], D["hout"], D["wout"], D["size"], D["stride"], D["pad"], D["bn"], D["leaky"])
        assert list(L.srcs) == D["srcs"] and [list(a) for a in L.anchors] == D["anchors"]
        assert (L.rows, L.row_offset) == (D["rows"], D["row_offset"])
    info = _ffi.PlanInfo()
    assert _ffi.lib().rtod_plan_get_info(h, C.byref(info)) == 0
    assert info.total_rows == ir.total_rows and info.conv_flops_per_frame == ir.conv_flops
    # every shortcut and head decode rides a conv epilogue; both concats are zero-copy
    kinds = []
    flops = 0
    for i in range(info.n_launches):
        li = _ffi.LaunchInfo()
        assert _ffi.lib().rtod_plan_get_launch(h, i, C.byref(li)) == 0
        kinds.append(li.kind)
        flops += li.flops_per_frame
    assert flops == ir.conv_flops
    assert 3 not in kinds and 5 not in kinds and 6 not in kinds        # no add / decode / copy launches
    n_conv = sum(1 for L in ir.layers if L.type == "convolutional")
    assert kinds.count(0) + kinds.count(7) == n_conv            # 7 = dedicated stem conv
    _ffi.lib().rtod_plan_destroy(h)


def test_cfg_extensions_parse_identically_on_both_sides():
    """activation=silu, [maxpool] symmetric=1, [upsample] mode=nearest (YOLOv5-style blocks, not reference grammar): the
    native planner and the Python IR agree on shapes and flags; the plan needs no stand-alone add / copy / decode."""
    text = cfgs.v5_style_mini_cfg()
    for res in (128, 224):
        rc, h = _plan(text, res)
        assert rc == 0, _ffi.last_error()
        d = _describe(h)
        ir = build_ir(parse_cfg_text(text), res)
        assert d["n_weight_floats"] == ir.n_weights and d["conv_flops"] == ir.conv_flops and d["total_rows"] == ir.total_rows
        for L, D in zip(ir.layers, d["layers"]):
            assert (L.type, L.cin, L.cout, L.hout, L.wout, L.size, L.stride, L.pad) == (D["type"], D["cin"], D["cout"], D["hout"], D["wout"], D["size"], D["stride"], D["pad"])
            assert (2 if L.silu else 1 if L.leaky else 0, L.nearest, L.pool_pad) == (D["act"], D["nearest"], D["pool_pad"])
        info = _ffi.PlanInfo()
        assert _ffi.lib().rtod_plan_get_info(h, C.byref(info)) == 0
        kinds = []
        for i in range(info.n_launches):
            li = _ffi.LaunchInfo()
            assert _ffi.lib().rtod_plan_get_launch(h, i, C.byref(li)) == 0
            kinds.append(li.kind)
        assert 3 not in kinds and 5 not in kinds and 6 not in kinds
        assert _ffi.lib().rtod_plan_set_precision(h, 1) == 0, _ffi.last_error()      # expressible in the split-f16 format too
        _ffi.lib().rtod_plan_destroy(h)


def test_batch_statistics_bn_is_an_fp32_plan_option():
    """Option bn_batch_stats (the reference's as-run BatchNorm): exact-fp32 plans only, same launch list
    (the statistics / normalisation kernels ride the conv launches), no dedicated stem kernel (layer 0 has BatchNorm)."""
    rc, h = _plan(cfgs.yolov3_cfg(), 416)
    assert rc == 0
    lib = _ffi.lib()
    info0 = _ffi.PlanInfo(); assert lib.rtod_plan_get_info(h, C.byref(info0)) == 0
    assert lib.rtod_plan_set_option(h, b"bn_batch_stats", 1) == 0, _ffi.last_error()
    info1 = _ffi.PlanInfo(); assert lib.rtod_plan_get_info(h, C.byref(info1)) == 0
    assert info1.n_launches == info0.n_launches + 1                      # input pack + generic conv instead of the stem kernel
    assert lib.rtod_plan_set_precision(h, 1) != 0 and "bn_batch_stats" in _ffi.last_error()
    assert lib.rtod_plan_set_precision(h, 0) == 0
    mean = (C.c_double * 32)(); var = (C.c_double * 32)()
    assert lib.rtod_plan_bn_batch_stats(h, 0, mean, var, 32, None) != 0    # no weights loaded / nothing run yet: refused, not a crash
    lib.rtod_plan_destroy(h)


def test_yolov5s_style_graph_matches_published_counts_and_plans_without_fallback_kernels():
    """cfgs.yolov5s_style_cfg restates the PUBLISHED YOLOv5s v6.0 architecture (no source offline): 16.4-16.5 GFLOPs at 640,
    7.2 M parameters, 25 200 output rows — and the native planner agrees with the Python IR, fuses every shortcut / decode
    and places every concat (incl. SPPF's four-way one) without copies."""
    text = cfgs.yolov5s_style_cfg()
    ir = build_ir(parse_cfg_text(text), 640)
    assert ir.total_rows == 25200 and 7.2e6 < ir.n_weights < 7.3e6 and 16.4e9 < ir.conv_flops < 16.5e9
    rc, h = _plan(text, 640)
    assert rc == 0, _ffi.last_error()
    d = _describe(h)
    assert d["conv_flops"] == ir.conv_flops and d["n_weight_floats"] == ir.n_weights and d["total_rows"] == ir.total_rows
    for L, D in zip(ir.layers, d["layers"]):
        assert (L.type, L.cin, L.cout, L.hout, L.wout, L.size, L.stride, L.pad, list(L.srcs)) == (D["type"], D["cin"], D["cout"], D["hout"], D["wout"], D["size"], D["stride"], D["pad"], D["srcs"])
        assert (L.decode_v5, L.nearest, L.pool_pad) == (D["decode_v5"], D["nearest"], D["pool_pad"])
    info = _ffi.PlanInfo()
    assert _ffi.lib().rtod_plan_get_info(h, C.byref(info)) == 0
    kinds = []
    for i in range(info.n_launches):
        li = _ffi.LaunchInfo()
        assert _ffi.lib().rtod_plan_get_launch(h, i, C.byref(li)) == 0
        kinds.append(li.kind)
    assert kinds.count(3) == kinds.count(5) == kinds.count(6) == 0 and kinds.count(0) == 60 and kinds.count(4) == 3 and kinds.count(2) == 2
    _ffi.lib().rtod_plan_destroy(h)


def test_buffer_plan_has_no_live_overlap():
    rc, h = _plan(cfgs.yolov3_cfg(), 608)
    assert rc == 0
    d = _describe(h)
    bufs = d["bufs"]
    for i, a in enumerate(bufs):
        sa = (a["floats_per_frame"] * 8 + 63) // 64 * 64
        for b in bufs[i + 1:]:
            sb = (b["floats_per_frame"] * 8 + 63) // 64 * 64
            live = not (a["last"] < b["first"] or b["last"] < a["first"])
            mem = not (a["offset"] + sa <= b["offset"] or b["offset"] + sb <= a["offset"])
            assert not (live and mem), (a, b)
    assert d["arena_floats"] * 4 < 1.0e9          # 608x608 batch 8 fits well under 1 GB of the 288 GB
    _ffi.lib().rtod_plan_destroy(h)


def test_error_codes_and_messages():
    lib = _ffi.lib()
    rc, h = _plan("[net]\nheight=416\n[convolutional]\nfilters=16\nsize=3\nstride=1\npad=1\nactivation=leaky\n[banana]\nx=1\n", 416)
    assert rc == -3 and "unknown block" in _ffi.last_error().lower()      # reference asserts (darknet.py:524-526)
    rc, h = _plan("[convolutional]\nfilters=1\n", 416)
    assert rc == -3
    rc, h = _plan(cfgs.yolov3_tiny_cfg(), 400)                            # 400/13 grid mismatch is caught
    assert rc in (-3, 0)
    h2 = C.c_void_p()
    t = cfgs.yolov3_tiny_cfg().encode()
    assert lib.rtod_plan_create(t, len(t), 416, 320, 1, 0, C.byref(h2)) == -1     # non-square
    rc, h = _plan(cfgs.yolov3_tiny_cfg(), 416)
    assert rc == 0
    # forward before load_weights is a state error, not a crash (no device touched)
    assert lib.rtod_forward(h, C.c_void_p(16), 1, C.c_void_p(16), None) == -4
    # short weight stream (reference: view_as raises)
    import numpy as np
    w = np.zeros(100, np.float32)
    assert lib.rtod_plan_load_weights(h, w.ctypes.data_as(C.c_void_p), w.size) == -5
    assert "needs" in _ffi.last_error()
    lib.rtod_plan_destroy(h)


def test_darknet_host_class_without_gpu(tmp_path):
    """The nn.Module mirror keeps the reference's surface; forward refuses CPU tensors."""
    import torch
    from realtimeobjectdetection_amd.darknet import Darknet
    from realtimeobjectdetection_amd import synth
    p = cfgs.write_cfg(str(tmp_path / "t.cfg"), cfgs.yolov3_tiny_cfg())
    m = Darknet(p, False)
    assert m.training and m.get_blocks() is m.blocks and m.get_module_list() is m.module_list
    assert m.net_info["height"] == "416" and len(m.module_list) == 24 and len(m.blocks) == 25
    assert m.blocks[17 + 1]["layers"] == ["-4"] and m.blocks[20 + 1]["layers"] == ["-1", " 8"]   # split in place like the reference
    sd = m.state_dict()
    assert len(sd) == 70 and "module_list.15.conv_15.bias" in sd
    ir = build_ir(m.blocks, 416)
    w = synth.synth_weights(ir)
    wp = synth.write_weights_file(str(tmp_path / "w.weights"), w, seen=9)
    m.load_weights(wp)
    assert int(m.seen) == 9 and m.header.tolist() == [0, 2, 0, 9, 0]
    import numpy as np
    assert np.array_equal(m.weight_stream(), w)
    with pytest.raises(RuntimeError):
        m.eval()(torch.zeros(1, 3, 416, 416))


def test_fused_pointwise_accounting_in_split_plans():
    """precision f16s3: YOLOv3's layer 2 (1x1, 64 -> 32) runs in layer 1's epilogue; its FLOPs move to the host launch and its
    own launch entry is empty, so the per-launch table still sums to the network's FLOPs (host-only: no device needed)."""
    rc, h = _plan(cfgs.yolov3_cfg(), 608)
    assert rc == 0
    ir = build_ir(parse_cfg_text(cfgs.yolov3_cfg()), 608)
    lib = _ffi.lib()
    assert lib.rtod_plan_set_precision(h, 1) == 0, _ffi.last_error()
    info = _ffi.PlanInfo()
    assert lib.rtod_plan_get_info(h, C.byref(info)) == 0
    flops, hosts, empty = 0, [], []
    for i in range(info.n_launches):
        li = _ffi.LaunchInfo()
        assert lib.rtod_plan_get_launch(h, i, C.byref(li)) == 0
        flops += li.flops_per_frame
        if li.fused_pointwise:
            hosts.append(li.layer)
        if li.kind == 0 and li.flops_per_frame == 0:
            empty.append(li.layer)
    assert flops == ir.conv_flops
    assert hosts == [1] and empty == [2]
    lib.rtod_plan_destroy(h)


def test_plan_options_are_explicit_state_not_environment(monkeypatch):
    """Fusion / kernel-selection switches are per-plan options (rtod_plan_set_option); environment variables that earlier
    builds honoured are ignored by the product library.  A refused option leaves the plan as it was."""
    lib = _ffi.lib()
    for var in ("RTOD_NO_PW", "RTOD_NO_STEM", "RTOD_NO_BAND", "RTOD_NO_AUTOTUNE", "RTOD_DBG_ZERO", "RTOD_F16S3_VARIANT", "RTOD_CONV_VARIANT"):
        monkeypatch.setenv(var, "1")

    def kinds(h):
        info = _ffi.PlanInfo()
        assert lib.rtod_plan_get_info(h, C.byref(info)) == 0
        out = []
        for i in range(info.n_launches):
            li = _ffi.LaunchInfo()
            assert lib.rtod_plan_get_launch(h, i, C.byref(li)) == 0
            out.append(li)
        return out

    rc, h = _plan(cfgs.yolov3_cfg(classes=3), 416)
    assert rc == 0
    base = kinds(h)
    assert [li.kind for li in base].count(7) == 1 and not ({1, 3, 5, 6} & {li.kind for li in base})    # env ignored: stem kernel, all fused
    assert lib.rtod_plan_set_precision(h, 1) == 0
    assert any(li.fused_pointwise for li in kinds(h))                                                # RTOD_NO_PW ignored
    assert lib.rtod_plan_set_option(h, b"fuse_pointwise", 0) == 0
    assert not any(li.fused_pointwise for li in kinds(h))
    # the split-f16 format has no stand-alone add kernel: refused, plan unchanged
    before = [(li.layer, li.kind) for li in kinds(h)]
    assert lib.rtod_plan_set_option(h, b"fuse_shortcut", 0) == -3 and "add" in _ffi.last_error()
    assert [(li.layer, li.kind) for li in kinds(h)] == before
    assert lib.rtod_plan_set_option(h, b"no_such_option", 1) == -1
    assert lib.rtod_plan_set_precision(h, 0) == 0
    for name in (b"fuse_shortcut", b"fuse_decode", b"zero_copy_concat", b"stem_kernel"):
        assert lib.rtod_plan_set_option(h, name, 0) == 0, _ffi.last_error()
    ks = [li.kind for li in kinds(h)]
    assert ks.count(3) == 23 and ks.count(5) == 3 and ks.count(6) == 4 and ks.count(1) == 1 and 7 not in ks
    ir = build_ir(parse_cfg_text(cfgs.yolov3_cfg(classes=3)), 416)
    assert sum(li.flops_per_frame for li in kinds(h)) == ir.conv_flops
    lib.rtod_plan_destroy(h)
    # the planner picks the stand-alone kernels by itself when the graph forbids fusion
    rc, h = _plan(cfgs.mini_fallback_cfg(), 64)
    assert rc == 0
    assert {3, 4, 5, 6} <= {li.kind for li in kinds(h)}
    assert lib.rtod_plan_set_precision(h, 1) == -3
    lib.rtod_plan_destroy(h)


def test_launch_kernel_names_have_the_format_rocprofv3_prints():
    """rtod_plan_launch_kernel_name feeds every name-keyed join of bench.py / tools (per-kernel roofline, HBM-traffic lookup):
    each name must be the demangled instantiation name rocprofv3 prints for that kernel family.  Checked against the name
    column of the committed profiles/r0*_kernel_stats.csv (numbers masked: same family, same template arity, same argument
    list), for the heuristic YOLOv3 608 plan and with the patch tiles forced (round 2 labelled variant 214, the weights-resident
    patch kernel, as a kernel of another family)."""
    import csv, glob, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mask = lambda n: re.sub(r"\d+", "N", n)
    known = set()
    for f in glob.glob(os.path.join(root, "profiles", "r0*_kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            known.add(mask(row["Name"]))
    assert any(mask("conv_band_f16s3_kernel") in k for k in known) and any(mask("conv_patch_wres_f16s3_kernel") in k for k in known)
    lib = _ffi.lib()
    family = {range(100, 150): "conv_igemm_f16s3_kernel", range(150, 170): "conv_band_f16s3_kernel", range(170, 190): "conv_ring_f16s3_kernel",
              range(210, 214): "conv_patch_f16s3_kernel", range(214, 215): "conv_patch_wres_f16s3_kernel", range(230, 231): "conv_stem2_f16s3_kernel"}
    seen = set()
    for force in (-1, 114, 112, 72):
        rc, h = _plan(cfgs.yolov3_cfg(), 608)
        assert rc == 0
        if force >= 0:
            assert lib.rtod_plan_set_option(h, b"force_f16s3_variant", force) == 0, _ffi.last_error()
        assert lib.rtod_plan_set_precision(h, 1) == 0, _ffi.last_error()
        info = _ffi.PlanInfo()
        assert lib.rtod_plan_get_info(h, C.byref(info)) == 0
        for i in range(info.n_launches):
            li = _ffi.LaunchInfo()
            assert lib.rtod_plan_get_launch(h, i, C.byref(li)) == 0
            buf = C.create_string_buffer(256)
            assert lib.rtod_plan_launch_kernel_name(h, i, buf, 256) == 0, _ffi.last_error()
            name = buf.value.decode()
            if li.kind != 0 or li.flops_per_frame == 0:
                assert name == ""
                continue
            fam = [v for k, v in family.items() if li.variant in k]
            assert fam and ("rtod::" + fam[0] + "<") in name, (li.layer, li.variant, name)
            assert mask(name) in known, (li.layer, li.variant, name)
            seen.add(fam[0])
        lib.rtod_plan_destroy(h)
    assert {"conv_band_f16s3_kernel", "conv_patch_wres_f16s3_kernel", "conv_patch_f16s3_kernel", "conv_ring_f16s3_kernel", "conv_stem2_f16s3_kernel"} <= seen


def test_tile_table_is_validated_per_launch():
    """rtod_plan_set_tiles (bench.py --tiles: profiled passes replay the timing run's kernels) refuses tables that do not fit the
    plan — wrong length, a band tile on a non-band layer, a generic tile on a band layer, a split-K mode on a layer that does not
    split K — and get_tiles returns what set_tiles installed (host-only: nothing is launched)."""
    lib = _ffi.lib()
    rc, h = _plan(cfgs.yolov3_cfg(), 608)
    assert rc == 0 and lib.rtod_plan_set_precision(h, 1) == 0, _ffi.last_error()
    info = _ffi.PlanInfo()
    assert lib.rtod_plan_get_info(h, C.byref(info)) == 0
    n = info.n_launches
    assert lib.rtod_plan_get_tiles(h, 8, None, 0) == -4 and "autotune" in _ffi.last_error()      # RTOD_E_STATE: nothing tuned yet
    infos = []
    for i in range(n):
        li = _ffi.LaunchInfo()
        assert lib.rtod_plan_get_launch(h, i, C.byref(li)) == 0
        infos.append((li.kind, li.variant, li.ksize, li.stride, li.hout, li.flops_per_frame))
    band = [i for i, (k, v, *_r) in enumerate(infos) if k == 0 and 150 <= v < 170]
    ring_or_generic = [i for i, (k, v, ks, st, ho, fl) in enumerate(infos) if k == 0 and fl > 0 and ks == 1 and 100 <= v < 150 or (k == 0 and 170 <= v < 190)]
    assert band and ring_or_generic
    table = [-1] * n
    arr = lambda t: (C.c_int * len(t))(*t)
    assert lib.rtod_plan_set_tiles(h, 8, arr(table[:-1]), n - 1) == -1                           # wrong length
    t = list(table); t[band[0]] = 50 + 3
    t[ring_or_generic[0]] = 70 + 2
    assert lib.rtod_plan_set_tiles(h, 8, arr(t), n) == 0, _ffi.last_error()
    got = (C.c_int * n)()
    assert lib.rtod_plan_get_tiles(h, 8, got, n) == n and list(got) == t
    li = _ffi.LaunchInfo()
    assert lib.rtod_plan_get_launch(h, band[0], C.byref(li)) == 0 and li.variant == 153
    bad = list(table); bad[ring_or_generic[0]] = 50                                              # band tile on a 1x1 layer
    assert lib.rtod_plan_set_tiles(h, 8, arr(bad), n) == -1 and "not a valid tile" in _ffi.last_error()
    bad = list(table); bad[band[0]] = 6                                                          # generic tile on a band layer
    assert lib.rtod_plan_set_tiles(h, 8, arr(bad), n) == -1
    k2 = [i for i in band if infos[i][4] == 19]
    bad = list(table); bad[k2[0]] = 50                                                           # 19x19: split-K layer, mode 0 is not
    assert lib.rtod_plan_set_tiles(h, 8, arr(bad), n) == -1
    bad = list(table); bad[band[0]] = 57                                                         # split-K mode on a 76x76 layer
    assert lib.rtod_plan_set_tiles(h, 8, arr(bad), n) == -1
    assert lib.rtod_plan_set_tiles(h, 9, arr(t), n) == -1                                        # batch beyond max_batch
    lib.rtod_plan_destroy(h)
