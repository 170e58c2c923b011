#!/usr/bin/env python
"""Per-layer differential check of two forced split-f16 tile variants (plan option force_f16s3_variant): the first layer whose output
differs localises a kernel bug.   python tools/diff_layers.py <variant> [baseline variant = 10]"""
import os, sys, tempfile, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from realtimeobjectdetection_amd import cfgs, synth
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
def run(variant):
    cfg_text = cfgs.yolov3_cfg(); ir = build_ir(parse_cfg_text(cfg_text), 416)
    d = tempfile.mkdtemp()
    m = Darknet(cfgs.write_cfg(os.path.join(d, "v3.cfg"), cfg_text), True).eval()
    m.net_info["height"] = 416; m.precision = "f16s3"; m.options["force_f16s3_variant"] = int(variant)
    m.load_weight_stream(synth.synth_weights(ir)); m.keep_all_layers = True
    x = torch.from_numpy(synth.synth_frames(2, 416)).cuda()
    with torch.no_grad(): y = m(x).clone()
    outs = {}
    for L in ir.layers:
        try: outs[L.index] = m.read_layer(L.index, 2).cpu().numpy()
        except Exception: pass
    return y.cpu().numpy(), outs, ir
ya, oa, ir = run(int(sys.argv[2]) if len(sys.argv) > 2 else 10); yb, ob, _ = run(int(sys.argv[1]))
for i in sorted(oa):
    if i in ob:
        d = np.abs(oa[i] - ob[i]).max(); s = np.abs(oa[i]).max()
        if d > 1e-6 * max(s, 1): print("layer", i, ir.layers[i].type, ir.layers[i].size, ir.layers[i].stride, ir.layers[i].cin, ir.layers[i].cout, ir.layers[i].hout, "maxdiff %.3e" % d, "absmax %.3e" % s); 
print("final", np.abs(ya - yb).max())
