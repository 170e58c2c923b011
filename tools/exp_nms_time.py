#!/usr/bin/env python
"""write_results on its own: the GPU NMS chain (zero, filter, sort + suppress, emit) timed with HIP events on the predictions
of a synthetic-weight network, nothing else on the card.   python tools/exp_nms_time.py [--net yolov3] [--res 608] [--batch 8]
Under `rocprofv3 --kernel-trace --stats` the per-kernel durations are the stand-alone ones (bench.py overlaps them with the
next forward)."""
import argparse, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realtimeobjectdetection_amd import cfgs, synth
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
from realtimeobjectdetection_amd.util import write_results_async

ap = argparse.ArgumentParser()
ap.add_argument("--net", default="yolov3"); ap.add_argument("--res", type=int, default=608); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--conf", type=float, default=0.6); ap.add_argument("--iters", type=int, default=100)
args = ap.parse_args()
cfg_text = {"yolov3": cfgs.yolov3_cfg, "yolov3-tiny": cfgs.yolov3_tiny_cfg}[args.net]()
ir = build_ir(parse_cfg_text(cfg_text), args.res)
with tempfile.TemporaryDirectory() as d:
    m = Darknet(cfgs.write_cfg(os.path.join(d, "t.cfg"), cfg_text), True).eval()
    m.net_info["height"] = args.res
    m.load_weights(synth.write_weights_file(os.path.join(d, "t.weights"), synth.synth_weights(ir)))
x = torch.from_numpy(synth.synth_frames(args.batch, args.res)).cuda()
with torch.no_grad():
    y = m(x).clone()
    for _ in range(10):
        rows, counts = write_results_async(y, 80, args.conf, 0.5, cap=4096)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        rows, counts = write_results_async(y, 80, args.conf, 0.5, cap=4096)
    e1.record(); torch.cuda.synchronize()
c = counts.cpu().tolist()
print({"net": args.net, "res": args.res, "batch": args.batch, "rows": y.shape[1], "conf": args.conf, "write_results_us": round(e0.elapsed_time(e1) / args.iters * 1e3, 2),
       "detections": c[0], "candidates": c[1], "per_image": c[2:2 + args.batch]})
