#!/usr/bin/env python
"""Latency datapoint for SURVEY.md §8(f) row 3: YOLOv3-tiny 416x416 batch 1 (BASELINE configs[0]) on the exact-fp32 kernels,
forward + write_results, one stream, HIP-event timed.   python tools/exp_tiny_latency.py [--batch 1] [--res 416]"""
import argparse, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realtimeobjectdetection_amd import cfgs, synth
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
from realtimeobjectdetection_amd.util import write_results_async

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--res", type=int, default=416)
ap.add_argument("--iters", type=int, default=200); args = ap.parse_args()
cfg_text = cfgs.yolov3_tiny_cfg()
ir = build_ir(parse_cfg_text(cfg_text), args.res)
with tempfile.TemporaryDirectory() as d:
    m = Darknet(cfgs.write_cfg(os.path.join(d, "t.cfg"), cfg_text), True).eval()
    m.net_info["height"] = args.res
    m.load_weights(synth.write_weights_file(os.path.join(d, "t.weights"), synth.synth_weights(ir)))
x = torch.from_numpy(synth.synth_frames(args.batch, args.res)).cuda()
print('stage: eager', flush=True)
with torch.no_grad():
    for _ in range(20):
        write_results_async(m(x), 80, 0.6, 0.5, cap=4096)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        write_results_async(m(x), 80, 0.6, 0.5, cap=4096)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
print('stage: graph', flush=True)
# the same work replayed as one HIP graph (Darknet.make_graphed): forward + write_results_async
run = m.make_graphed(x, post=lambda y: write_results_async(y, 80, 0.6, 0.5, cap=4096))
for _ in range(20):
    run(x)
torch.cuda.synchronize()
e0.record()
for _ in range(args.iters):
    run(x)
e1.record(); torch.cuda.synchronize()
ms_graph = e0.elapsed_time(e1) / args.iters
print('stage: readback', flush=True)
import time
t0 = time.perf_counter()
for _ in range(args.iters):
    y, (rows, counts) = run(x)
    n = int(counts[0].item())                      # one result on the host per frame: the serving latency
lat = (time.perf_counter() - t0) / args.iters * 1e3
print({"graph_ms_per_batch": round(ms_graph, 4), "graph_frames_per_s": round(args.batch * 1000.0 / ms_graph, 1),
       "graph_latency_with_host_readback_ms": round(lat, 4), "launches": m._info.n_launches})
print({"net": "yolov3-tiny", "res": args.res, "batch": args.batch, "precision": m.active_precision, "ms_per_batch": round(ms, 4),
       "frames_per_s": round(args.batch * 1000.0 / ms, 1), "gflop_per_frame": round(ir.conv_flops / 1e9, 3)})
