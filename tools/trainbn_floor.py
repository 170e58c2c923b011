"""Noise floor of the as-run (training-mode BatchNorm) path on the CPU: the oracle's float32 PyTorch ops (bit-identical to the
reference, tests/test_oracle_golden.py) against the SAME graph evaluated in float64, and against itself at another thread count
(another summation order of the reference's own BLAS back end).  Shows how far two correct float32 evaluations of this mode sit
from each other before any GPU kernel is involved.  Usage: python tools/trainbn_floor.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.nn.functional as F
from realtimeobjectdetection_amd import cfgs, synth
from oracle import darknet_ref as O

def rel(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))

def forward64(ref, x, batch_stats):
    """RefDarknet.forward restated in float64 (same graph, same parameters)."""
    saved = ref.params
    ref.params = {i: {k: v.double() for k, v in d.items()} for i, d in saved.items()}
    try:
        return ref.forward(x.double(), keep_layers=True, batch_stats=batch_stats)
    finally:
        ref.params = saved

out = {}
for net, res, B in [("yolov3-tiny", 416, 2), ("yolov3", 416, 2), ("yolov3", 320, 3)]:
    tag = "%s_%d_b%d" % (net, res, B)
    cfg_text = {"yolov3-tiny": cfgs.yolov3_tiny_cfg, "yolov3": cfgs.yolov3_cfg}[net]()
    ref = O.RefDarknet(cfg_text, res)
    w = synth.synth_weights(ref.ir); ref.load_weight_stream(w)
    x = torch.from_numpy(synth.synth_frames(B, res, seed=31))
    row = {}
    for mode, bs in (("eval", False), ("batch_stats", True)):
        with torch.no_grad():
            torch.set_num_threads(8)
            y32, l32 = ref.forward(x, keep_layers=True, batch_stats=bs)
            torch.set_num_threads(1)
            y32b, _ = ref.forward(x, keep_layers=True, batch_stats=bs)
            torch.set_num_threads(8)
            y64, l64 = forward64(ref, x, bs)
        e = rel(y32.numpy(), y64.numpy()); e2 = rel(y32.numpy(), y32b.numpy())
        per = []
        for i in sorted(l32):
            a, b = l32[i].numpy(), l64[i].numpy()
            per.append((i, float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))))
        per.sort(key=lambda t: -t[1])
        row[mode] = {"f32_vs_f64_max": float(e.max()), "f32_vs_f64_p999": float(np.quantile(e, 0.999)),
                     "frac_gt_1e-4": float((e > 1e-4).mean()), "threads8_vs_threads1_max": float(e2.max()),
                     "worst_layers_err_over_absmax": per[:6]}
        print(tag, mode, json.dumps(row[mode]))
    out[tag] = row
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
