#!/usr/bin/env python
"""Time of the stand-alone 1x1 layers under every tile that can run them, forced one at a time, in ONE process (YOLOv3, per-launch
hipEvent pairs):
    python tools/exp_pw_modes.py out.json [res] [batch] [variant ...]
Variant -1 = autotune.  Per variant: the summed time of the 1x1 launches per grid size (layers the forced tile is not valid for
run their heuristic tile and show up under that tile's name), and whether the network output equals the first variant's bit for bit."""
import json, os, sys, tempfile, collections
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from realtimeobjectdetection_amd import cfgs, synth, _ffi
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
out = sys.argv[1]
res = int(sys.argv[2]) if len(sys.argv) > 2 else 608
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
variants = [int(v) for v in sys.argv[4:]] or [-1, 0, 6, 10, 70, 72, 74, 76] + list(range(90, 101))
text = cfgs.yolov3_cfg(); ir = build_ir(parse_cfg_text(text), res)
d = tempfile.mkdtemp()
cfg_path = cfgs.write_cfg(os.path.join(d, "m.cfg"), text)
w = synth.synth_weights(ir)
x = torch.from_numpy(synth.synth_frames(B, res)).cuda()
ref = None
report = {}
for v in variants:
    m = Darknet(cfg_path, True).eval()
    m.net_info["height"] = res; m.precision = "f16s3"; m.overflow_check = "off"
    m.autotune = v < 0
    if v >= 0: m.options["force_f16s3_variant"] = v
    m.load_weight_stream(w)
    with torch.no_grad():
        y = m(x); m(x)
        tot = None
        for _ in range(10):
            _, ms = m.forward_timed(x)
            tot = ms if tot is None else tot + ms
    tot /= 10
    torch.cuda.synchronize()
    same = None
    if ref is None: ref = y.clone()
    else: same = bool(torch.equal(y, ref))
    g = collections.defaultdict(lambda: [0, 0.0, set()])
    for li, t in zip(m.launch_infos(), tot):
        if li.kind != 0 or li.ksize != 1 or li.flops_per_frame == 0: continue
        name = _ffi.lib().rtod_conv_variant_name(li.variant).decode()
        e = g["%d" % li.hout]; e[0] += 1; e[1] += float(t); e[2].add(name)
    pw = sum(e[1] for e in g.values())
    report[str(v)] = {"equal_to_first": same, "sum_ms": float(tot.sum()), "pointwise_ms": pw,
                      "grids": {k: {"n": e[0], "ms": round(e[1], 4), "us_each": round(1000 * e[1] / e[0], 1), "tiles": sorted(e[2])} for k, e in g.items()}}
    print("variant %3d  forward %.4f ms  1x1 %.4f ms  equal %s  | " % (v, float(tot.sum()), pw, same) +
          "  ".join("%s: %.4f ms /%d (%s)" % (k, e[1], e[0], ",".join(sorted(n.replace("conv_", "").replace("_f16s3", "") for n in e[2]))) for k, e in sorted(g.items(), key=lambda kv: -int(kv[0]))), flush=True)
    del m
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
json.dump(report, open(out, "w"), indent=1)
