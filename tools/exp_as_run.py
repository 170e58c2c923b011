#!/usr/bin/env python
"""The as-run path (Darknet left in training mode: batch-statistics BatchNorm on the exact-fp32 kernels), YOLOv3 res x res batch B:
    python tools/exp_as_run.py [res] [batch] [forwards]
prints ms per forward + write_results; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, tempfile, time, warnings
sys.path.insert(0, os.getcwd())
import torch
from realtimeobjectdetection_amd import cfgs, synth
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
from realtimeobjectdetection_amd.util import write_results
res = int(sys.argv[1]) if len(sys.argv) > 1 else 608
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
text = cfgs.yolov3_cfg()
w = synth.synth_weights(build_ir(parse_cfg_text(text), res))
d = tempfile.mkdtemp()
m = Darknet(cfgs.write_cfg(os.path.join(d, "m.cfg"), text), True)
m.net_info["height"] = res
m.load_weight_stream(w)
x = torch.from_numpy(synth.synth_frames(B, res)).cuda()
with torch.no_grad(), warnings.catch_warnings():
    warnings.simplefilter("ignore", RuntimeWarning)
    for _ in range(2): write_results(m(x), 80, 0.6, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): write_results(m(x), 80, 0.6, 0.5)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("as-run BN %dx%d b%d: %.3f ms per step, %.1f frames/s" % (res, res, B, 1e3 * dt / n, B * n / dt))
