#!/usr/bin/env python
"""Per-launch table of YOLOv3-tiny (exact-fp32 kernels):  python tools/exp_tiny_layers.py [batch] [res]"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from realtimeobjectdetection_amd import cfgs, synth, _ffi
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = int(sys.argv[2]) if len(sys.argv) > 2 else 416
text = cfgs.yolov3_tiny_cfg(); ir = build_ir(parse_cfg_text(text), res)
d = tempfile.mkdtemp()
m = Darknet(cfgs.write_cfg(os.path.join(d, "t.cfg"), text), True).eval()
m.net_info["height"] = res
m.load_weight_stream(synth.synth_weights(ir))
x = torch.from_numpy(synth.synth_frames(B, res)).cuda()
with torch.no_grad():
    m(x); m(x)
    tot = None
    for _ in range(20):
        _, ms = m.forward_timed(x)
        tot = ms if tot is None else tot + ms
tot /= 20
for li, t in zip(m.launch_infos(), tot):
    print("L%-3d kind %d k%d s%d %4d->%4d @%3d  %.4f ms  %s" % (li.layer, li.kind, li.ksize, li.stride, li.cin, li.cout, li.hout, t,
          _ffi.lib().rtod_conv_variant_name(li.variant).decode() if li.kind == 0 else ""))
print("sum %.4f ms" % tot.sum())
