#!/usr/bin/env python
"""YOLOv3 608x608 batch 8, split-f16: eager launch list vs the same forward replayed as one HIP graph (Darknet.make_graphed)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realtimeobjectdetection_amd import cfgs, synth
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
text = cfgs.yolov3_cfg(); ir = build_ir(parse_cfg_text(text), 608)
with tempfile.TemporaryDirectory() as d:
    m = Darknet(cfgs.write_cfg(os.path.join(d, "m.cfg"), text), True).eval()
    m.net_info["height"] = 608; m.precision = "f16s3"; m.overflow_check = "off"
    m.load_weight_stream(synth.synth_weights(ir))
x = torch.from_numpy(synth.synth_frames(8, 608)).cuda()
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    m(x); m(x)
    eager = timeit(lambda: m(x))
    run = m.make_graphed(x)
    graph = timeit(lambda: run(x))
    eager2 = timeit(lambda: m(x))
print({"eager_ms": round(eager, 4), "graph_ms": round(graph, 4), "eager_again_ms": round(eager2, 4)})
