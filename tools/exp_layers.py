#!/usr/bin/env python
"""Per-launch timing table of one YOLOv3 forward (hipEvent pair around every launch), optionally with plan options:
    python tools/exp_layers.py out.json [res] [batch] [opt=val ...]
With RTOD_LIB=librtod_diag.so and RTOD_DBG_ZERO=<bits> this is the load / epilogue ablation (results are then garbage).
RTOD_TILES=<file saved by bench.py --tiles> installs that tile table instead of autotuning."""
import json, os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from realtimeobjectdetection_amd import cfgs, synth, _ffi
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
out = sys.argv[1]
res = int(sys.argv[2]) if len(sys.argv) > 2 else 608
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[4:])
text = cfgs.yolov3_cfg(); ir = build_ir(parse_cfg_text(text), res)
d = tempfile.mkdtemp()
m = Darknet(cfgs.write_cfg(os.path.join(d, "m.cfg"), text), True).eval()
m.net_info["height"] = res; m.precision = "f16s3"; m.overflow_check = "off"
m.autotune = opts.pop("autotune", 1) != 0
m.options.update(opts)
m.load_weight_stream(synth.synth_weights(ir))
x = torch.from_numpy(synth.synth_frames(B, res)).cuda()
if os.environ.get("RTOD_TILES"):                      # a tile table saved by bench.py --tiles: no autotune, the same kernels in every process
    table = json.load(open(os.environ["RTOD_TILES"])).get("f16s3_%d_b%d" % (res, B))
    if table is not None:
        m.prepare(B); m.set_tiles(B, table)
with torch.no_grad():
    m(x); m(x)
    tot = None
    for _ in range(10):
        _, ms = m.forward_timed(x)
        tot = ms if tot is None else tot + ms
tot /= 10
rows = []
for li, t in zip(m.launch_infos(), tot):
    rows.append({"layer": li.layer, "kind": li.kind, "variant": li.variant, "k": li.ksize, "s": li.stride, "cin": li.cin, "cout": li.cout,
                 "hout": li.hout, "ms": round(float(t), 5), "name": _ffi.lib().rtod_conv_variant_name(li.variant).decode() if li.kind == 0 else ""})
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
json.dump({"per_launch": rows, "sum_ms": float(tot.sum()), "opts": opts, "dbg": os.environ.get("RTOD_DBG_ZERO", "")}, open(out, "w"), indent=1)
print(out, "sum %.4f ms" % float(tot.sum()))
