import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from realtimeobjectdetection_amd import synth
from realtimeobjectdetection_amd.util import write_results_async
dev = torch.device("cuda", 0)
m = bench.build_model(608, dev, 8, "f16s3")[0]
x = torch.from_numpy(synth.synth_frames(8, 608)).to(dev)
with torch.no_grad(): y = m(x)
torch.cuda.synchronize()
for _ in range(3): write_results_async(y, 80, 0.6, 0.5, cap=4096)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): rows, counts = write_results_async(y, 80, 0.6, 0.5, cap=4096)
e1.record(); torch.cuda.synchronize()
print("nms gpu ms per call %.4f" % (e0.elapsed_time(e1) / 50), "counts", counts[:2].tolist())
t0 = time.perf_counter()
for _ in range(50): rows, counts = write_results_async(y, 80, 0.6, 0.5, cap=4096)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("nms host issue ms per call %.4f" % ((t1 - t0) / 50 * 1e3))
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(20): y = m(x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("forward host issue ms %.3f, total ms %.3f" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
