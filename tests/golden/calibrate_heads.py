#!/usr/bin/env python
"""Calibrate the synthetic head objectness biases (realtimeobjectdetection_amd/synth.py HEAD_OBJ_BIAS_TABLE).

With random weights on noise frames each objectness channel is ~ constant + small spatial
noise, so a fixed bias of -4 gives either 0 % or 60 % candidates.  This script measures, with the
CPU oracle, the per-(head, anchor) bias that makes ~1.4 % of anchors exceed conf 0.6 at 416x416
(the candidate load SURVEY.md §6 measured on the reference) and prints the table to paste into
synth.py.  It is part of the synthetic-input recipe, not of the product.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from realtimeobjectdetection_amd import cfgs, synth
from oracle.darknet_ref import RefDarknet

TARGET = 0.014
for name, gen in (("yolov3-tiny", cfgs.yolov3_tiny_cfg), ("yolov3", cfgs.yolov3_cfg)):
    m = RefDarknet(gen(), 416)
    m.load_weight_stream(synth.synth_weights(m.ir, obj_bias_table={}))   # flat -4 bias
    x = torch.from_numpy(synth.synth_frames(2, 416))
    with torch.no_grad():
        _, outs = m.forward(x, keep_layers=True)
    table = []
    for L in m.ir.layers:
        if L.type == "yolo":
            t = outs[L.index - 1]
            row = []
            for a in range(len(L.anchors)):
                logit = t[:, a * (5 + L.classes) + 4].reshape(-1).numpy() - synth.HEAD_OBJ_BIAS
                q = np.quantile(logit, 1.0 - TARGET)
                row.append(round(float(np.log(0.6 / 0.4) - q), 4))
            table.append(row)
    print(f'    "{name}": {table},')
