#!/usr/bin/env python
"""Phase times of the NMS workgroup of image 0: shader-clock stamps after prefix / gather / sort / pair tests / greedy replay /
compaction (each phase carries ~1.5-2k cycles of the stamp's own store).  Needs a diagnostic library:
  cd realtimeobjectdetection_amd/csrc && hipcc $(CXXFLAGS of nms.o) -DRTOD_NMS_STAMPS -c nms.hip -o nms_stamps.o &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o ../librtod_nmsdiag.so <all objects but nms.o> nms_stamps.o
  RTOD_LIB=$PWD/realtimeobjectdetection_amd/librtod_nmsdiag.so python tools/exp_nms_stamps.py [net res batch]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realtimeobjectdetection_amd import cfgs, synth, util
from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
from realtimeobjectdetection_amd.darknet import Darknet

net, res, batch = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("yolov3", 608, 8)
cfg_text = {"yolov3": cfgs.yolov3_cfg, "yolov3-tiny": cfgs.yolov3_tiny_cfg}[net]()
ir = build_ir(parse_cfg_text(cfg_text), res)
with tempfile.TemporaryDirectory() as d:
    m = Darknet(cfgs.write_cfg(os.path.join(d, "t.cfg"), cfg_text), True).eval()
    m.net_info["height"] = res
    m.load_weights(synth.write_weights_file(os.path.join(d, "t.weights"), synth.synth_weights(ir)))
x = torch.from_numpy(synth.synth_frames(batch, res)).cuda()
with torch.no_grad():
    y = m(x).clone()
n = y.shape[1]
for _ in range(5):
    util.write_results_async(y, 80, 0.6, 0.5, cap=4096)
torch.cuda.synchronize()
ws = util._nms_buffers(y.device, batch, n, 4096)[0]
a256 = lambda v: (v + 255) & ~255
P = 1
while P < n: P <<= 1
G = (n + 255) // 256
off = a256(4 * (3 * batch + 4 + 2 * batch * G)) + a256(8 * batch * P)
st = ws.view(torch.int64)[off // 8: off // 8 + 7].cpu().tolist()
names = ["prefix", "gather", "sort", "pair tests", "greedy replay", "compaction"]
print({nm: st[i + 1] - st[i] for i, nm in enumerate(names)}, "total cycles", st[6] - st[0])
