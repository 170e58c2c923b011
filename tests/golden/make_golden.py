#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by importing the REAL reference.

Runs only in the build container, where /root/reference is mounted (read-only).  The reference
never travels to the GPU box; the fixtures written here (data: inputs are regenerated from
seeds, expected outputs are stored) do.  Nothing from /root/reference is copied.

    python tests/golden/make_golden.py

Harness per SURVEY.md App. C: a stub ``cv2`` module (src/util.py:4 imports cv2 at top level but
the hot path never touches it), ``sys.dont_write_bytecode`` so the read-only tree is not written.
Canonical mode: ``Darknet(cfg, False).eval()``.
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def import_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.darknet import Darknet                     # noqa
    from src.util import predict_transform, write_results, bbox_iou, confidence_mask   # noqa
    return Darknet, predict_transform, write_results, bbox_iou, confidence_mask


def ref_model(Darknet, cfg_name, res, weights):
    from realtimeobjectdetection_amd.synth import write_weights_file
    m = Darknet(os.path.join(REF, "cfg", cfg_name), False).eval()
    m.net_info["height"] = res
    with tempfile.NamedTemporaryFile(suffix=".weights") as f:
        write_weights_file(f.name, weights, seen=32013312)
        m.load_weights(f.name)
    return m


def module_ir(m):
    """Layer description read off the reference's module_list / blocks."""
    out = []
    for i, (blk, mod) in enumerate(zip(m.blocks[1:], m.module_list)):
        d = {"index": i, "type": blk["type"]}
        if blk["type"] == "convolutional":
            conv = mod[0]
            d.update(cin=conv.in_channels, cout=conv.out_channels, size=conv.kernel_size[0],
                     stride=conv.stride[0], pad=conv.padding[0], bias=conv.bias is not None,
                     bn=any("batch_norm" in n for n, _ in mod.named_children()),
                     leaky=any("leaky" in n for n, _ in mod.named_children()))
        elif blk["type"] == "yolo":
            d.update(anchors=[list(a) for a in mod[0].anchors], classes=int(blk["classes"]))
        elif blk["type"] == "route":
            d.update(layers=[int(a) for a in blk["layers"]])
        elif blk["type"] == "shortcut":
            d.update(frm=int(blk["from"]))
        elif blk["type"] == "maxpool":
            d.update(size=int(blk["size"]), stride=int(blk["stride"]))
        out.append(d)
    return out


def state_sha(m):
    h = hashlib.sha256()
    sd = m.state_dict()
    for k in sorted(sd.keys()):
        if k.endswith("num_batches_tracked"):
            continue
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    return h.hexdigest(), len(sd)


PROBE_IDX_SEED = 99


def layer_probes(outputs, n_layers):
    """(mean, absmax, 32 fixed-index samples) per layer output tensor (NCHW)."""
    rng = np.random.Generator(np.random.PCG64(PROBE_IDX_SEED))
    means, amax, samples, idxs = [], [], [], []
    for i in range(n_layers):
        t = outputs[i].numpy().astype(np.float32)
        flat = t.reshape(-1)
        idx = rng.integers(0, flat.size, 32)
        means.append(float(flat.astype(np.float64).mean()))
        amax.append(float(np.abs(flat).max()))
        samples.append(flat[idx])
        idxs.append(idx)
    return (np.array(means), np.array(amax), np.stack(samples).astype(np.float32),
            np.stack(idxs).astype(np.int64))


def main():
    import torch
    torch.manual_seed(0)
    Darknet, predict_transform, write_results, bbox_iou, confidence_mask = import_reference()
    from realtimeobjectdetection_amd import cfgs, synth
    from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir

    meta = {"torch": torch.__version__, "numpy": np.__version__}

    # ---- G1/G2: cfg -> IR, weight loader ------------------------------------------------
    for name, gen in (("yolov3-tiny", cfgs.yolov3_tiny_cfg), ("yolov3", cfgs.yolov3_cfg)):
        ir = build_ir(parse_cfg_text(gen()), 416)
        w = synth.synth_weights(ir)
        m = ref_model(Darknet, name + ".cfg", 416, w)
        sha, nkeys = state_sha(m)
        n_params = sum(p.numel() for p in m.parameters())
        n_stream = 0
        for mod in m.module_list:
            for sub in mod.children():
                if isinstance(sub, torch.nn.Conv2d):
                    n_stream += sub.weight.numel() + (sub.bias.numel() if sub.bias is not None else 0)
                elif isinstance(sub, torch.nn.BatchNorm2d):
                    n_stream += 4 * sub.num_features
        g = {"layers": module_ir(m), "n_weight_floats": n_stream, "state_sha256": sha,
             "state_keys": nkeys, "n_params": n_params, "header": m.header.tolist(),
             "seen": int(m.seen)}
        with open(os.path.join(HERE, f"ir_{name}.json"), "w") as f:
            json.dump(g, f, indent=0)
        print(name, "weights floats", n_stream, "sha", sha[:12])

    # ---- G3/G4: forward outputs + per-layer probes --------------------------------------
    cases = [("yolov3-tiny", cfgs.yolov3_tiny_cfg, 416, 1, 1),
             ("yolov3-tiny", cfgs.yolov3_tiny_cfg, 608, 2, 7),
             ("yolov3", cfgs.yolov3_cfg, 416, 2, 97),
             ("yolov3", cfgs.yolov3_cfg, 608, 1, 97)]
    for name, gen, res, B, row_stride in cases:
        ir = build_ir(parse_cfg_text(gen()), res)
        w = synth.synth_weights(ir)
        m = ref_model(Darknet, name + ".cfg", res, w)
        x = torch.from_numpy(synth.synth_frames(B, res))
        outputs = {}
        hooks = []
        for i, mod in enumerate(m.module_list):
            hooks.append(mod.register_forward_hook(lambda mod_, inp, out, i=i: outputs.__setitem__(i, out.detach())))
        with torch.no_grad():
            y = m(x)
        for h in hooks:
            h.remove()
        # route/shortcut/yolo modules are placeholders never called: take their tensors from a
        # manual replay of the reference's own forward bookkeeping
        with torch.no_grad():
            mods = m.blocks[1:]
            outs = {}
            xx = x
            for i in range(len(mods)):
                t = mods[i]["type"]
                if t in ("convolutional", "upsample", "maxpool"):
                    xx = m.module_list[i](xx); outs[i] = xx
                elif t == "route":
                    xx = m.pass_through_route(i, mods, outs, xx)
                elif t == "shortcut":
                    xx = m.pass_through_shortcut(i, mods, outs, xx)
                elif t == "yolo":
                    outs[i] = outs[i - 1]
        means, amax, samples, idxs = layer_probes(outs, len(mods))
        y = y.numpy()
        cand = float((y[..., 4] > 0.6).mean())
        np.savez(os.path.join(HERE, f"fwd_{name}_{res}_b{B}.npz"),
                 rows=y[:, ::row_stride, :].astype(np.float32), row_stride=row_stride,
                 n_rows=y.shape[1], layer_mean=means, layer_absmax=amax,
                 layer_samples=samples, layer_sample_idx=idxs, cand_frac=cand)
        print(name, res, B, "out", y.shape, "cand frac %.4f" % cand, "absmax last", amax[-2])
        # write_results on the real forward output (end-to-end golden)
        with torch.no_grad():
            det = write_results(torch.from_numpy(y.copy()), 80, 0.6, 0.5)
        det = np.zeros((0, 8), np.float32) if isinstance(det, int) else det.numpy()
        np.savez(os.path.join(HERE, f"det_{name}_{res}_b{B}.npz"), det=det.astype(np.float32))
        print("   detections", det.shape)

    # ---- G5: head decode ----------------------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(555))
    anchors_sets = {13: [(116, 90), (156, 198), (373, 326)], 26: [(30, 61), (62, 45), (59, 119)],
                    52: [(10, 13), (16, 30), (33, 23)], 19: [(116, 90), (156, 198), (373, 326)],
                    38: [(30, 61), (62, 45), (59, 119)], 76: [(10, 13), (16, 30), (33, 23)]}
    hd = {}
    for G, anchors in anchors_sets.items():
        res = 416 if G in (13, 26, 52) else 608
        raw = (rng.standard_normal((2, 255, G, G), dtype=np.float32) * np.float32(1.5))
        with torch.no_grad():
            dec = predict_transform(torch.from_numpy(raw.copy()), res, anchors, 80, False).numpy()
            dec_tr = predict_transform(torch.from_numpy(raw.copy()), res, anchors, 80, False, TRAIN=True).numpy()
        step = max(1, dec.shape[1] // 120)
        hd[f"dec_{G}"] = dec[:, ::step, :].astype(np.float32)
        hd[f"dectrain_{G}"] = dec_tr[:, ::step, :].astype(np.float32)
        hd[f"step_{G}"] = step
    np.savez(os.path.join(HERE, "head_decode.npz"), **hd)
    print("head decode fixtures", {k: v.shape for k, v in hd.items() if hasattr(v, "shape") and v.ndim})

    # ---- G6: write_results ---------------------------------------------------------------
    nms = {}

    def run_wr(tag, pred, conf=0.6, thr=0.5, ncls=80):
        with torch.no_grad():
            r = write_results(torch.from_numpy(pred.copy()), ncls, conf, thr)
        if isinstance(r, int):
            nms[f"{tag}_isint"] = np.array(1)
            nms[f"{tag}_out"] = np.zeros((0, 8), np.float32)
        else:
            nms[f"{tag}_isint"] = np.array(0)
            nms[f"{tag}_out"] = r.numpy().astype(np.float32)
        nms[f"{tag}_args"] = np.array([conf, thr, ncls], dtype=np.float64)
        print("  nms", tag, "->", "int 0" if isinstance(r, int) else tuple(r.shape))

    # synthetic recipe cases (inputs regenerated from seeds by the tests)
    run_wr("synth_b2_n2535", synth.synth_predictions(2, 2535, 80, 416, seed=2024))
    run_wr("synth_b8_n10647", synth.synth_predictions(8, 10647, 80, 416, seed=2025))
    run_wr("synth_b3_n22743", synth.synth_predictions(3, 22743, 80, 608, seed=2026), 0.6, 0.4)
    run_wr("synth_dense", synth.synth_predictions(2, 3000, 80, 416, seed=2027, obj_mu=0.5, obj_sigma=1.0), 0.5, 0.45)
    run_wr("synth_c20", synth.synth_predictions(2, 2000, 20, 416, seed=2028, obj_mu=-2.0), 0.3, 0.5, 20)
    # edge cases: explicit small inputs stored with the fixture
    def blank(B, N, C=80):
        p = np.zeros((B, N, 5 + C), np.float32); p[..., 2:4] = 10.0; return p
    e = blank(2, 16)                                   # nothing above conf -> int 0
    e[..., 4] = 0.3
    nms["edge_none_in"] = e; run_wr("edge_none", e)
    e = blank(1, 8); e[0, 3, :5] = [100, 120, 40, 60, 0.9]; e[0, 3, 5 + 17] = 0.8      # single candidate
    nms["edge_single_in"] = e; run_wr("edge_single", e)
    e = blank(1, 8); e[0, :, 4] = 0.6; e[0, 2, 4] = np.nextafter(np.float32(0.6), np.float32(1))   # obj == conf dropped (strict >)
    e[0, :, 0] = np.arange(8) * 50; e[0, :, 1] = 30; e[0, :, 5] = 0.5
    nms["edge_eqconf_in"] = e; run_wr("edge_eqconf", e)
    e = blank(1, 6); e[0, :, :5] = [200, 200, 50, 80, 0.0]; e[0, :, 4] = [0.95, 0.9, 0.85, 0.8, 0.75, 0.7]
    e[0, :, 5 + 3] = 0.9                                # identical boxes -> IoU 1 -> only the top one stays
    nms["edge_identical_in"] = e; run_wr("edge_identical", e)
    e = blank(1, 6); e[0, :, :5] = [200, 200, 50, 80, 0.9]; e[0, :, 0] = [50, 51, 52, 300, 301, 302]
    e[0, :, 4] = [0.91, 0.92, 0.93, 0.94, 0.95, 0.96]  # class score underflows to exactly 0 on rows 0,3
    e[0, :, 5 + 7] = [0.0, 0.7, 0.6, 0.0, 0.8, 0.5]
    nms["edge_zeroscore_in"] = e; run_wr("edge_zeroscore", e)
    e = blank(1, 4); e[0, :, :5] = [200, 200, 50, 80, 0.9]; e[0, :, 4] = [0.91, 0.92, 0.93, 0.94]
    nms["edge_allzeroscore_in"] = e; run_wr("edge_allzeroscore", e)    # every class score 0 -> empty [0,8]
    e = blank(2, 12); rr = np.random.Generator(np.random.PCG64(7))
    e[..., 0:2] = rr.uniform(50, 350, (2, 12, 2)); e[..., 2:4] = rr.uniform(30, 90, (2, 12, 2))
    e[..., 4] = rr.uniform(0.61, 0.99, (2, 12)); e[0, :, 4] = 0.1     # image 0 empty, image 1 full; one class only
    e[..., 5 + 11] = rr.uniform(0.3, 0.9, (2, 12))
    nms["edge_oneclass_in"] = e; run_wr("edge_oneclass", e)
    np.savez(os.path.join(HERE, "nms.npz"), **nms)

    # ---- bbox_iou / confidence_mask spot vectors ------------------------------------------
    rr = np.random.Generator(np.random.PCG64(11))
    xy = rr.uniform(0, 400, (257, 2)).astype(np.float32)
    wh = rr.uniform(1, 200, (257, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh, rr.uniform(0, 1, (257, 3)).astype(np.float32)], 1).astype(np.float32)
    with torch.no_grad():
        iou = bbox_iou(torch.from_numpy(boxes[:1]), torch.from_numpy(boxes[1:])).numpy()
        t = torch.from_numpy(synth.synth_predictions(1, 64, 80, 416, seed=5))
        cm = confidence_mask(t, 0.02).numpy()
    np.savez(os.path.join(HERE, "iou.npz"), boxes=boxes, iou=iou.astype(np.float32),
             cm_sum=cm.astype(np.float64).sum(), cm_nnz_rows=int((cm[0, :, 4] != 0).sum()))

    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f)
    print("done")


if __name__ == "__main__":
    main()
