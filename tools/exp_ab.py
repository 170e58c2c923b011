#!/usr/bin/env python
"""A/B of plan options in ONE process (cdna guide 5.4 rule 24): one Darknet per setting (the autotune memo is process-wide, so
every instance runs the same tiles), interleaved rounds of per-launch hipEvent timings, group sums per kernel class.
    python tools/exp_ab.py out.json [res] [batch] name:opt=val,opt=val ...      (name 'base:' = no options)"""
import collections, json, os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from realtimeobjectdetection_amd import cfgs, synth, _ffi
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet
out = sys.argv[1]; res = int(sys.argv[2]); B = int(sys.argv[3])
settings = []
for s in sys.argv[4:]:
    name, _, kv = s.partition(":")
    settings.append((name, dict((p.split("=")[0], int(p.split("=")[1])) for p in kv.split(",") if p)))
text = cfgs.yolov3_cfg(); ir = build_ir(parse_cfg_text(text), res)
w = synth.synth_weights(ir)
x = torch.from_numpy(synth.synth_frames(B, res)).cuda()
d = tempfile.mkdtemp()
models = []
for name, opts in settings:
    m = Darknet(cfgs.write_cfg(os.path.join(d, "m.cfg"), text), True).eval()
    m.net_info["height"] = res; m.precision = "f16s3"; m.overflow_check = "off"
    m.autotune = opts.pop("autotune", 1) != 0
    m.options.update(opts); m.load_weight_stream(w)
    with torch.no_grad():
        m(x); y = m(x)
    torch.cuda.synchronize()
    if models and not torch.equal(y, models[0][2]): print("!! output of", name, "differs from", models[0][0], "max abs", float((y - models[0][2]).abs().max()))
    models.append((name, m, y.clone()))
def cls(li, nm):
    if li.kind != 0: return "other"
    if li.ksize == 1: return "1x1_%d" % li.hout
    if li.stride == 2: return "3x3s2"
    if "band" in nm: return "band%d" % li.hout
    return "3x3s1gen"
acc = {name: None for name, _, _ in models}
ROUNDS = 6
with torch.no_grad():
    for r in range(ROUNDS):
        for name, m, _y in models:
            for _ in range(3):
                _, ms = m.forward_timed(x)
                acc[name] = ms if acc[name] is None else acc[name] + ms
tabs = {}
for name, m, _y in models:
    t = acc[name] / (3 * ROUNDS); g = collections.defaultdict(float)
    for li, v in zip(m.launch_infos(), t):
        nm = _ffi.lib().rtod_conv_variant_name(li.variant).decode() if li.kind == 0 else ""
        g[cls(li, nm)] += float(v)
    g["TOTAL"] = float(t.sum()); tabs[name] = dict(g)
keys = sorted({k for g in tabs.values() for k in g})
print("%-10s" % "", " ".join("%10s" % n for n, _, _ in models))
for k in keys: print("%-10s" % k, " ".join("%10.4f" % tabs[n].get(k, 0.0) for n, _, _ in models))
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
json.dump(tabs, open(out, "w"), indent=1)
