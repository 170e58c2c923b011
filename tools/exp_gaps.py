#!/usr/bin/env python
"""Plain forwards (no per-launch events) for a rocprofv3 --kernel-trace run, and the analysis of that trace: per-dispatch
duration and the gap to the next dispatch of the stream.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/exp_gaps.py run [res] [batch] [iters]
    python tools/exp_gaps.py report OUT"""
import csv, glob, os, sys, collections
if sys.argv[1] == "run":
    import tempfile
    sys.path.insert(0, os.getcwd())
    import torch
    from realtimeobjectdetection_amd import cfgs, synth
    from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
    from realtimeobjectdetection_amd.darknet import Darknet
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 608
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    text = cfgs.yolov3_cfg(); ir = build_ir(parse_cfg_text(text), res)
    d = tempfile.mkdtemp()
    m = Darknet(cfgs.write_cfg(os.path.join(d, "m.cfg"), text), True).eval()
    m.net_info["height"] = res; m.precision = "f16s3"; m.overflow_check = "off"
    m.load_weight_stream(synth.synth_weights(ir))
    x = torch.from_numpy(synth.synth_frames(B, res)).cuda()
    with torch.no_grad():
        m(x); m(x)
        torch.cuda.synchronize()
        for _ in range(iters):
            y = m(x)
        torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    n_per = 77
    # the last `iters` forwards: take the trailing dispatches, group into forwards of equal length by finding the stem kernel
    starts = [i for i, r in enumerate(rows) if "conv_stem2" in r["Kernel_Name"]]
    starts = starts[-8:]                                  # last 8 forwards
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    tot_d = tot_g = 0.0; nf = 0
    for a, b in zip(starts[:-1], starts[1:]):
        fw = rows[a:b]
        nf += 1
        for r, nx in zip(fw, fw[1:] + [rows[b]]):
            k = r["Kernel_Name"].split("<")[0].replace("void rtod::", "")
            d_ = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            g_ = (int(nx["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3
            dur[k].append(d_); gap[k].append(g_); tot_d += d_; tot_g += g_
    print("forwards %d: mean kernel time %.1f us, mean gaps %.1f us, launches per forward %d" % (nf, tot_d / nf, tot_g / nf, (starts[1] - starts[0])))
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        print("%-34s n %4d  dur mean %7.2f us  gap-after mean %5.2f us (min %5.2f max %5.2f)" % (k, len(dur[k]) // nf, sum(dur[k]) / len(dur[k]), sum(gap[k]) / len(gap[k]), min(gap[k]), max(gap[k])))
