#!/usr/bin/env python
"""Does replaying the forward (+ write_results_async) as ONE HIP graph beat enqueuing its ~80 launches one by one when the GPU
is the bottleneck?   python tools/exp_graph.py [res] [batch] [steps]   (single stream, one batch in flight)"""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import torch
from realtimeobjectdetection_amd import cfgs, synth, util
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
res = int(sys.argv[1]) if len(sys.argv) > 1 else 608
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
text = cfgs.yolov3_cfg(); ir = build_ir(parse_cfg_text(text), res)
d = tempfile.mkdtemp()
m = Darknet(cfgs.write_cfg(os.path.join(d, "m.cfg"), text), True).eval()
m.net_info["height"] = res; m.precision = "f16s3"; m.overflow_check = "off"
m.load_weight_stream(synth.synth_weights(ir))
x = torch.from_numpy(synth.synth_frames(B, res)).cuda()
post = lambda y: util.write_results_async(y, 80, 0.6, 0.5, cap=4096)
def timed(fn):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
with torch.no_grad():
    m(x); post(m(x)); torch.cuda.synchronize()
    eager = lambda: post(m(x))
    run = m.make_graphed(x, post)
    graphed = lambda: run(x)
    for rep in range(3):
        e = timed(eager); g = timed(graphed)
        print("rep %d: eager %.4f ms/step (%.0f frames/s)   graphed %.4f ms/step (%.0f frames/s)" % (rep, e, B / e * 1e3, g, B / g * 1e3), flush=True)
