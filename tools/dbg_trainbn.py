import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, warnings
warnings.simplefilter("ignore")
from realtimeobjectdetection_amd import cfgs, synth
from realtimeobjectdetection_amd.darknet import Darknet
from oracle import darknet_ref as O
g = np.load("tests/golden/trainbn.npz")
for net, res, B in [("yolov3-tiny", 416, 2), ("yolov3", 416, 2), ("yolov3", 320, 3)]:
    tag = "%s_%d_b%d" % (net, res, B)
    cfg_text = {"yolov3-tiny": cfgs.yolov3_tiny_cfg, "yolov3": cfgs.yolov3_cfg}[net]()
    d = tempfile.mkdtemp()
    m = Darknet(cfgs.write_cfg(os.path.join(d, "n.cfg"), cfg_text), True)
    m.net_info["height"] = res
    ref = O.RefDarknet(cfg_text, res); w = synth.synth_weights(ref.ir); m.load_weight_stream(w); ref.load_weight_stream(w)
    x = torch.from_numpy(synth.synth_frames(B, res, seed=31))
    m.keep_all_layers = True
    with torch.no_grad():
        y = m(x.cuda()).cpu().numpy()
        want, outs = ref.forward(x, keep_layers=True, batch_stats=True)
    stride = int(g["stride_" + tag]); got = y[:, ::stride]; gw = g["rows_" + tag]
    e = np.abs(got - gw) / np.maximum(1.0, np.abs(gw))
    print(tag, "max %.2e p99.9 %.2e p99 %.2e frac>1e-4 %.2e" % (e.max(), np.quantile(e, 0.999), np.quantile(e, 0.99), (e > 1e-4).mean()), "argmax col", np.unravel_index(e.argmax(), e.shape)[2])
    worst = []
    for D in m.plan_description()["layers"]:
        i = D["index"]
        if D["type"] == "yolo" or (D["type"] == "convolutional" and D["fused_into"] >= 0): continue
        gg = m.read_layer(i, B).cpu().numpy(); wv = outs[i].numpy()
        worst.append((float(np.abs(gg - wv).max() / max(1.0, np.abs(wv).max())), i, D["type"]))
    worst.sort(reverse=True); print("   worst layers (err/absmax):", [(("%.1e" % a), i, t) for a, i, t in worst[:5]])
