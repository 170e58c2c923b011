"""Per-layer error of the GPU path against the oracle, eval-mode BatchNorm vs batch-statistics BatchNorm (the mode the reference's
callers run, detect.py:185-194), for the three cases of tests/golden/trainbn.npz; end-to-end error against the reference rows.
    python tools/dbg_trainbn.py [out.json]        (GPU box; profiles/r03_trainbn_layers.json is its committed output)"""
import json, os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, warnings
warnings.simplefilter("ignore")
from realtimeobjectdetection_amd import cfgs, synth
from realtimeobjectdetection_amd.darknet import Darknet
from oracle import darknet_ref as O
g = np.load("tests/golden/trainbn.npz")
out = {}
for net, res, B in [("yolov3-tiny", 416, 2), ("yolov3", 416, 2), ("yolov3", 320, 3)]:
    tag = "%s_%d_b%d" % (net, res, B)
    cfg_text = {"yolov3-tiny": cfgs.yolov3_tiny_cfg, "yolov3": cfgs.yolov3_cfg}[net]()
    ref = O.RefDarknet(cfg_text, res); w = synth.synth_weights(ref.ir); ref.load_weight_stream(w)
    x = torch.from_numpy(synth.synth_frames(B, res, seed=31))
    row = {}
    for mode in ("eval", "batch_stats"):
        d = tempfile.mkdtemp()
        m = Darknet(cfgs.write_cfg(os.path.join(d, "n.cfg"), cfg_text), True)
        if mode == "eval": m.eval()
        m.net_info["height"] = res; m.precision = "fp32"; m.load_weight_stream(w)
        m.keep_all_layers = True
        with torch.no_grad():
            y = m(x.cuda()).cpu().numpy()
            want, outs = ref.forward(x, keep_layers=True, batch_stats=(mode == "batch_stats"))
        e = np.abs(y - want.numpy()) / np.maximum(1.0, np.abs(want.numpy()))
        layers = []
        for D in m.plan_description()["layers"]:
            i = D["index"]
            if D["type"] == "yolo" or (D["type"] == "convolutional" and D["fused_into"] >= 0): continue
            gg = m.read_layer(i, B).cpu().numpy(); wv = outs[i].numpy()
            layers.append({"layer": i, "type": D["type"], "err_over_absmax": float(np.abs(gg - wv).max() / max(1.0, np.abs(wv).max()))})
        row[mode] = {"gpu_vs_oracle_max": float(e.max()), "gpu_vs_oracle_p999": float(np.quantile(e, 0.999)), "frac_gt_1e-4": float((e > 1e-4).mean()), "layers": layers}
        if mode == "batch_stats":
            stride = int(g["stride_" + tag]); gw = g["rows_" + tag]
            er = np.abs(y[:, ::stride] - gw) / np.maximum(1.0, np.abs(gw))
            row[mode]["gpu_vs_reference_rows_max"] = float(er.max()); row[mode]["gpu_vs_reference_rows_p999"] = float(np.quantile(er, 0.999))
        worst = sorted(layers, key=lambda r: -r["err_over_absmax"])[:4]
        print(tag, mode, "max %.2e p99.9 %.2e frac>1e-4 %.1e" % (e.max(), np.quantile(e, 0.999), (e > 1e-4).mean()),
              "worst layers:", [(r["layer"], "%.1e" % r["err_over_absmax"]) for r in worst])
        del m
    out[tag] = row
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
