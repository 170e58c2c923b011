"""Experiment: does splitting the batch over two HIP streams (two plans) hide the per-layer tail rounds?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from realtimeobjectdetection_amd import synth
from realtimeobjectdetection_amd.util import write_results_async
dev = torch.device("cuda", 0)
R, B = 608, 8
x = torch.from_numpy(synth.synth_frames(B, R)).to(dev)
for nsplit in (1, 2, 4):
    models = [bench.build_model(R, dev, B // nsplit, "f16s3")[0] for _ in range(nsplit)]
    streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    xs = [x[i * (B // nsplit):(i + 1) * (B // nsplit)].contiguous() for i in range(nsplit)]
    def step():
        outs = []
        for m, s, xi in zip(models, streams, xs):
            with torch.cuda.stream(s), torch.no_grad():
                outs.append(m(xi))
        return outs
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("splits", nsplit, "forward-only fps %.1f  ms/step %.3f" % (B * 20 / dt, dt / 20 * 1e3), flush=True)
    del models
