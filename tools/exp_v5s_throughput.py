#!/usr/bin/env python
"""Datapoint for BASELINE config (5)'s SHAPE: the YOLOv5s-style graph (cfgs.yolov5s_style_cfg: published architecture restated,
synthetic weights, parity unpinned) at 640x640 batch 8 (--precision auto = split-f16 kernels, fp32 = exact-fp32 kernels),
forward + class-offset batched NMS, HIP-event timed.   python tools/exp_v5s_throughput.py [--batch 8] [--res 640]"""
import argparse, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realtimeobjectdetection_amd import cfgs, synth, _ffi
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
from realtimeobjectdetection_amd.util import nms_class_offset

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--res", type=int, default=640)
ap.add_argument("--iters", type=int, default=50); ap.add_argument("--precision", default="auto"); args = ap.parse_args()
text = cfgs.yolov5s_style_cfg()
ir = build_ir(parse_cfg_text(text), args.res)
with tempfile.TemporaryDirectory() as d:
    m = Darknet(cfgs.write_cfg(os.path.join(d, "v5s.cfg"), text), True).eval()
    m.net_info["height"] = args.res
    m.precision = args.precision
    m.load_weight_stream(synth.synth_weights(ir))
x = torch.from_numpy(synth.synth_frames(args.batch, args.res)).cuda()
with torch.no_grad():
    for _ in range(5):
        y = m(x)
    # synthetic weights leave the objectness far below the usual 0.25: take the threshold that ~1 % of the rows pass, so that the
    # post-processing carries a realistic load (~250 candidates per image)
    score = (y[..., 4] * y[..., 5:].max(-1).values).flatten()
    conf = float(torch.quantile(score[:: max(1, score.numel() // 100000)], 0.99))
    det = nms_class_offset(y, 80, conf, 0.45)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        y = m(x)
    e1.record(); torch.cuda.synchronize()
    ms_fwd = e0.elapsed_time(e1) / args.iters
    e0.record()
    for _ in range(args.iters):
        y = m(x)
        det = nms_class_offset(y, 80, conf, 0.45)           # includes its host sync
    e1.record(); torch.cuda.synchronize()
    ms_all = e0.elapsed_time(e1) / args.iters
    _, per = m.forward_timed(x)
rows = sorted(((float(t), li.layer, li.ksize, li.stride, li.cin, li.cout, li.hout) for li, t in zip(m.launch_infos(), per)), reverse=True)[:6]
print({"graph": "yolov5s-style (published architecture restated; synthetic weights; parity unpinned)", "res": args.res, "batch": args.batch,
       "precision": m.active_precision, "gflop_per_frame": round(ir.conv_flops / 1e9, 3), "launches": m._info.n_launches,
       "forward_ms": round(ms_fwd, 4), "forward_frames_per_s": round(args.batch * 1000 / ms_fwd, 1),
       "forward_tflops": round(args.batch * ir.conv_flops / ms_fwd / 1e9, 2),
       "forward_plus_nms_ms": round(ms_all, 4), "frames_per_s": round(args.batch * 1000 / ms_all, 1), "detections": int(det.size(0)), "conf_thres_used": round(conf, 5),
       "slowest_launches_ms_layer_k_s_cin_cout_hout": [(round(r[0], 4),) + tuple(r[1:]) for r in rows]})
