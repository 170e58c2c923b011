#!/usr/bin/env python
"""bench.py — frames/s of the YOLOv3 hot path (Darknet.forward + write_results) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--res 608] [--batch 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both launch forms work: started plainly with ``--gpus N > 1`` (no WORLD_SIZE in the environment) this process starts N
child ranks itself — before it makes any GPU call — relays rank 0's JSON line and exits non-zero if a rank fails.

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
``Darknet.forward`` (one launch list on one HIP stream) + ``write_results`` (filter, sort, NMS; 3 launches) and, for
N > 1, the fixed-capacity RCCL all-gather of the detections (shard.FixedGather).  One process per GPU; frames are
sharded (weak scaling: every rank runs its own ``--batch`` frames), no collective on the data path except that final
gather.  Rank 0 prints ONE JSON line.

What the numbers in the line are (all measured in this run, same plans, same inputs):
  value         K steps between barrier + device-synchronise pairs, ``--inflight`` batches in flight (default 2: steps
                alternate over two plans on two HIP streams), ``write_results_async`` (device-side counts, no host sync)
                on a third stream.  The throughput-serving schedule; the metric string says so.
  single_stream one batch in flight: forward i+1 starts after forward i (write_results_async still on its own stream).
  dropin_api    the reference's calling sequence verbatim (detect.py:62-70): ``y = model(x)``;
                ``write_results(y, 80, conf, nms)`` returning a new [D,8] tensor — one plan, one stream, the host
                synchronisation of write_results inside every step.
  roofline      dominant kernel = the conv instantiation with the largest total time.  achieved = algorithmic conv FLOPs
                of its launches / their summed duration, from a hipEvent pair around every launch on the launch stream
                (rtod_forward_timed) in an instrumented single-stream replay right after the timed region.
  cpu_baseline  the oracle (oracle/darknet_ref.py: the reference's PyTorch-CPU op sequence + NMS) timed on
                this box's host cores on a bounded sample (rank 0, N = 1 only): all cores and one thread, forward and
                NMS separately.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0        # dense f16/bf16 MFMA; the split kernel issues 3 products per algorithmic MAC
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--res", type=int, default=608)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--conf", type=float, default=0.6)
    ap.add_argument("--nms", type=float, default=0.5)
    ap.add_argument("--precision", default="f16s3", choices=["fp32", "f16s3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-nms", action="store_true", help="run write_results on the forward's stream (no overlap with the next batch)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight: steps alternate over N plans (own activation arena) on N HIP streams, so one batch's "
                         "partial rounds, prologues and epilogues overlap the other's kernels; 1 = strictly sequential forwards")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT",
                    help="plan option (rtod_plan_set_option) for A/B runs, e.g. --opt k_slices=0; recorded in config.plan_options")
    ap.add_argument("--layers-out", default="", help="write the per-launch table (JSON) here")
    ap.add_argument("--tiles", default="", metavar="FILE",
                    help="tile table (JSON): if FILE holds a table for this resolution / batch it is installed (rtod_plan_set_tiles: no autotune "
                         "launches, the same kernels in every process — profiled passes replay the timing pass); otherwise autotune runs and "
                         "its table is written there")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra workloads reported beside the headline (BASELINE configs[1]: YOLOv3 416x416 batch 8; configs[4]'s "
                         "shape: the YOLOv5s-style graph 640x640 batch 8, parity unpinned)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ self-launch (N > 1)
def self_launch(args):
    """Start one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, rendezvous on 127.0.0.1).
    Runs BEFORE this process touches the GPU: it imports nothing that initialises HIP and never becomes a rank itself."""
    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # this pool's host driver only supports dmabuf IPC: without it RCCL's (and torch's) cross-process device-memory handles
        # fail with "hipIpcGetMemHandle: invalid argument".  The image exports it already; kept for a caller with a scrubbed environment.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        # rank 0 owns stdout (the one JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, c))
                for q in alive:
                    procs[q].terminate()                    # exact PIDs this process started
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------ model / baselines
def build_model(res, device, max_batch, precision, options=None):
    from realtimeobjectdetection_amd import cfgs, synth
    from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
    from realtimeobjectdetection_amd.darknet import Darknet
    cfg_text = cfgs.yolov3_cfg()
    ir = build_ir(parse_cfg_text(cfg_text), res)
    w = synth.synth_weights(ir)
    with tempfile.TemporaryDirectory() as d:
        cfg_path = cfgs.write_cfg(os.path.join(d, "yolov3.cfg"), cfg_text)
        wpath = synth.write_weights_file(os.path.join(d, "yolov3.weights"), w)
        m = Darknet(cfg_path, True).eval()
        m.net_info["height"] = res
        m.precision = precision
        m.options.update(options or {})
        m.load_weights(wpath)
    m.prepare(max_batch, device)
    return m, ir, w, cfg_text


def host_cpu_share():
    """Threads this job may use: affinity mask, cgroup quota, and the pool's 16-core share per GPU."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("RTOD_CPU_THREADS", "16"))))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(cfg_text, w, res, batch, conf, nms, warmup=3, iters=10):
    """Oracle forward and write_results on the host cores, BASELINE.md §4's protocol: 3 warm-up + 10 timed iterations of the
    bench batch on this job's share of the host cores (forward and write_results timed separately), then one thread on a
    single frame (1 warm-up + 3 timed).  About 25-30 s of CPU work on a 16-core share."""
    import torch
    from realtimeobjectdetection_amd import synth
    from oracle import darknet_ref as O
    cores = host_cpu_share()
    ref = O.RefDarknet(cfg_text, res)
    ref.load_weight_stream(w)
    x = torch.from_numpy(synth.synth_frames(batch, res))
    with torch.no_grad():
        torch.set_num_threads(cores)
        for _ in range(warmup):
            O.write_results(ref.forward(x), 80, conf, nms)
        tf = tn = 0.0
        for _ in range(iters):
            t0 = time.perf_counter()
            y = ref.forward(x)
            t1 = time.perf_counter()
            O.write_results(y, 80, conf, nms)
            t2 = time.perf_counter()
            tf += t1 - t0; tn += t2 - t1
        out = {"value": round(batch * iters / (tf + tn), 3), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": "%d warm-up + %d timed x (yolov3 %dx%d batch %d forward + write_results), torch %s CPU ops, %d threads = this job's "
                         "share of the host's cores (affinity / cgroup quota / the pool's 16 cores per GPU; %s visible)"
                         % (warmup, iters, res, res, batch, torch.__version__, cores, os.cpu_count()),
               "forward_s_per_batch": round(tf / iters, 4), "write_results_s_per_batch": round(tn / iters, 4),
               "cpu_model": cpu_model_name(), "host_cores_visible": os.cpu_count()}
        torch.set_num_threads(1)
        x1 = x[:1].contiguous()
        O.write_results(ref.forward(x1), 80, conf, nms)             # warm-up
        tf = tn = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            y1 = ref.forward(x1)
            t1 = time.perf_counter()
            O.write_results(y1, 80, conf, nms)
            t2 = time.perf_counter()
            tf += t1 - t0; tn += t2 - t1
        out["one_thread"] = {"value": round(3.0 / (tf + tn), 4), "unit": "frames/s", "forward_s": round(tf / 3, 4),
                             "write_results_s": round(tn / 3, 4), "sample": "1 warm-up + 3 timed x (batch 1 forward + write_results), 1 thread"}
        torch.set_num_threads(cores)
    return out


def rocprof_kernel_name(model, index):
    """Exact kernel instantiation name rocprofv3 prints for launch `index` (built next to the launch switch in csrc)."""
    import ctypes as C
    from realtimeobjectdetection_amd import _ffi
    buf = C.create_string_buffer(256)
    _ffi.check(_ffi.lib().rtod_plan_launch_kernel_name(model._plan, int(index), buf, 256))
    return buf.value.decode()


def roofline_from_launches(model, x, steps):
    """Instrumented replay: hipEvent pair around every launch; group conv launches by kernel instantiation."""
    import numpy as np
    from realtimeobjectdetection_amd import _ffi
    infos = model.launch_infos()
    B = x.size(0)
    runs = []
    for _ in range(steps):
        _, ms = model.forward_timed(x)
        runs.append(np.asarray(ms, dtype=np.float64))
    runs = np.stack(runs)
    tot = runs.mean(axis=0)                              # the roofline figures are averages, as the contract asks
    med = np.median(runs, axis=0)                        # per-layer table: also the median (one stalled replay moves a 20-sample mean by 2x)
    lib = _ffi.lib()
    groups = {}
    for idx, (li, ms) in enumerate(zip(infos, tot)):
        if li.kind != 0 or li.flops_per_frame == 0:      # non-conv launches; 1x1 convs fused into the previous conv's epilogue
            continue
        # group by the exact kernel instantiation rocprofv3 reports (tile variant + epilogue)
        name = lib.rtod_conv_variant_name(li.variant).decode()
        kname = rocprof_kernel_name(model, idx)
        g = groups.setdefault(kname, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "tile": name})
        g["ms"] += float(ms); g["flops"] += float(li.flops_per_frame) * B; g["launches"] += 1
        g["bytes"] += float(li.bytes_per_frame) * B + float(li.weight_bytes)
    dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
    name, g = dom
    achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
    conv_ms = sum(v["ms"] for v in groups.values())
    per_layer = [{"layer": li.layer, "kind": li.kind, "variant": li.variant, "k": li.ksize, "s": li.stride, "cin": li.cin,
                  "cout": li.cout, "hout": li.hout, "ms": round(float(ms), 5), "ms_median": round(float(md), 5),
                  "tflops": round(float(li.flops_per_frame) * B / (float(ms) * 1e-3) / 1e12, 2) if li.kind == 0 and ms > 0 else None,
                  "gbs": round((float(li.bytes_per_frame) * B + li.weight_bytes) / (float(ms) * 1e-3) / 1e9, 1) if ms > 0 else None}
                 for li, ms, md in zip(infos, tot, med)]
    # the pointwise (1x1, stride 1) group: HBM-bound by algorithmic bytes (north star: achieved GB/s on the 1x1 convs)
    pw = [(li, ms) for li, ms in zip(infos, tot) if li.kind == 0 and li.ksize == 1 and li.flops_per_frame > 0]
    pw_ms = sum(float(ms) for _, ms in pw)
    pw_bytes = sum(float(li.bytes_per_frame) * B + float(li.weight_bytes) for li, _ in pw)
    split = "f16s3" in name
    peak = round(PEAK_F16_MFMA_TFLOPS / 3.0, 1) if split else PEAK_FP32_MFMA_TFLOPS
    roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "peak_note": ("algorithmic FLOPs; each MAC issues 3 f16 MFMA products, peak = 2500/3" if split
                          else "exact-fp32 MFMA v_mfma_f32_32x32x2_f32"),
            "launches_per_step": g["launches"], "avg_launch_ms": round(g["ms"] / g["launches"], 5),
            "flops_per_launch": g["flops"] / g["launches"],
            "all_conv_tflops": round(sum(v["flops"] for v in groups.values()) / (conv_ms * 1e-3) / 1e12, 2),
            "forward_launch_ms_sum": round(float(tot.sum()), 4),
            "pointwise_convs": {"bound": "hbm", "launches": len(pw), "ms": round(pw_ms, 4),
                                "achieved": round(pw_bytes / (pw_ms * 1e-3) / 1e9, 1) if pw_ms > 0 else None,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(pw_bytes / (pw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pw_ms > 0 else None}}
    return roof, per_layer, groups


# ------------------------------------------------------------------------------------------ extra workloads (N = 1)
def _throughput(models, x, post, steps, warmup):
    """frames/s of forward + post over `steps` steps, batches alternating over `models` (one HIP stream each), `post` on a side
    stream behind an event — the schedule of the headline figure; device-synchronise on both sides of the timed region."""
    import torch
    dev = x.device
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in models[1:]]
    side = torch.cuda.Stream(device=dev)

    def step(i):
        k = i % len(models)
        with torch.no_grad(), torch.cuda.stream(streams[k]):
            y = models[k](x)
            side.wait_stream(streams[k])
            with torch.cuda.stream(side):
                post(y)
            y.record_stream(side)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return round(x.size(0) * steps / dt, 2), round(dt / steps * 1e3, 4)


def extra_workloads(dev, args):
    """The other single-GPU configurations BASELINE.json lists, measured in this run beside the headline: configs[1] (YOLOv3
    416x416 batch 8) and configs[4]'s SHAPE (YOLOv5s-style graph 640x640 batch 8 reusing the conv / NMS kernels; the reference
    fetches that model from the network, detect.py:255-285 — here: published architecture restated, synthetic weights, PARITY
    UNPINNED).  Same schedule as the headline (two batches in flight) and the strictly sequential figure."""
    import torch
    from realtimeobjectdetection_amd import cfgs, synth
    from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
    from realtimeobjectdetection_amd.darknet import Darknet
    from realtimeobjectdetection_amd.util import write_results_async, nms_class_offset_async
    out = {}
    B = 8
    # -- configs[1]
    ms_ = [build_model(416, dev, B, args.precision)[0] for _ in range(2)]
    for m_ in ms_:
        m_.overflow_check = "off"
    x = torch.from_numpy(synth.synth_frames(B, 416)).to(dev)
    with torch.no_grad():
        for m_ in ms_:
            m_(x)
    post = lambda y: write_results_async(y, 80, args.conf, args.nms, cap=4096)
    v2, t2 = _throughput(ms_, x, post, args.steps, args.warmup)
    v1, t1 = _throughput(ms_[:1], x, post, args.steps, args.warmup)
    ir = build_ir(parse_cfg_text(cfgs.yolov3_cfg()), 416)
    out["yolov3_416_b8"] = {"workload": "YOLOv3 cfg 416x416 batch=8, 1 GPU (BASELINE configs[1])", "value": v2, "unit": "frames/s", "ms_per_step": t2,
                            "single_stream": {"value": v1, "ms_per_step": t1}, "precision": args.precision,
                            "conv_gflop_per_frame": round(ir.conv_flops / 1e9, 3), "whole_path_tflops": round(v2 * ir.conv_flops / 1e12, 2),
                            "f16_range_overflow": bool(any(m_.overflowed() for m_ in ms_))}
    del ms_
    # -- configs[4]'s shape
    text = cfgs.yolov5s_style_cfg()
    ir5 = build_ir(parse_cfg_text(text), 640)
    w5 = synth.synth_weights(ir5)
    ms_ = []
    with tempfile.TemporaryDirectory() as d:
        for _ in range(2):
            m_ = Darknet(cfgs.write_cfg(os.path.join(d, "v5s.cfg"), text), True).eval()
            m_.net_info["height"] = 640
            m_.precision = "auto" if args.precision == "f16s3" else "fp32"
            m_.overflow_check = "off"
            m_.load_weight_stream(w5)
            ms_.append(m_)
    x = torch.from_numpy(synth.synth_frames(B, 640)).to(dev)
    with torch.no_grad():
        for m_ in ms_:
            y = m_(x)
    # synthetic weights leave obj * cls far below the usual 0.25: threshold = what ~1 % of the rows pass (~250 candidates per image)
    score = (y[..., 4] * y[..., 5:].max(-1).values).flatten()
    conf5 = float(torch.quantile(score[:: max(1, score.numel() // 100000)], 0.99))
    post = lambda y: nms_class_offset_async(y, 80, conf5, 0.45, cap=4096)
    v2, t2 = _throughput(ms_, x, post, args.steps, args.warmup)
    v1, t1 = _throughput(ms_[:1], x, post, args.steps, args.warmup)
    out["yolov5s_style_640_b8"] = {"workload": "YOLOv5s-SHAPED graph 640x640 batch=8 (BASELINE configs[4]'s shape): published v6.0 architecture restated in the "
                                               "cfg grammar, synthetic weights, class-offset batched NMS; PARITY UNPINNED (the reference fetches its model "
                                               "from torch.hub, detect.py:255-285: nothing to pin offline) - a workload of that shape for the kernels, not the reference's model",
                                   "value": v2, "unit": "frames/s", "ms_per_step": t2, "single_stream": {"value": v1, "ms_per_step": t1},
                                   "precision": ms_[0].active_precision, "conv_gflop_per_frame": round(ir5.conv_flops / 1e9, 3),
                                   "whole_path_tflops": round(v2 * ir5.conv_flops / 1e12, 2), "conf_threshold_used": round(conf5, 5),
                                   "f16_range_overflow": bool(any(m_.overflowed() for m_ in ms_))}
    return out


def as_run_bn_workload(dev, args):
    """What the two-import-line drop-in of INTEGRATION.md section 1 runs when the caller, like the reference's detect.py:185-194,
    never calls .eval(): BatchNorm on the statistics of the batch (src/darknet.py:493-495) — exact-fp32 kernels, raw conv ->
    bn_stats -> bn_apply per layer, running statistics updated on the device (one launch) — through the reference's calling
    sequence.  YOLOv3 at the headline resolution and batch; a slow parity path, reported so that its cost is a number."""
    import warnings
    import torch
    from realtimeobjectdetection_amd import cfgs, synth
    from realtimeobjectdetection_amd.cfg import parse_cfg_text, build_ir
    from realtimeobjectdetection_amd.darknet import Darknet
    from realtimeobjectdetection_amd.util import write_results
    R, B = args.res, args.batch
    text = cfgs.yolov3_cfg()
    w = synth.synth_weights(build_ir(parse_cfg_text(text), R))
    with tempfile.TemporaryDirectory() as d:
        m = Darknet(cfgs.write_cfg(os.path.join(d, "yolov3.cfg"), text), True)      # no .eval(): as detect.py builds it
        m.net_info["height"] = R
        m.load_weight_stream(w)
    x = torch.from_numpy(synth.synth_frames(B, R)).to(dev)
    steps, warm = max(1, min(args.steps, 10)), 2
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for _ in range(warm):
            det = write_results(m(x), 80, args.conf, args.nms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            det = write_results(m(x), 80, args.conf, args.nms)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"workload": "YOLOv3 cfg %dx%d batch=%d, Darknet(cfg, True) left in TRAINING mode (batch-statistics BatchNorm, what detect.py:185-194 runs), "
                        "reference calling sequence y = model(x); write_results(y, 80, conf, nms)" % (R, R, B),
            "value": round(B * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warm,
            "precision": m.active_precision, "launches_per_forward": int(m._info.n_launches),
            "detections_last_step": 0 if isinstance(det, int) else int(det.size(0)),
            "note": "parity path (exact-fp32 kernels; frames of a batch depend on each other; tolerance of this mode: DESIGN.md section 1); "
                    "call .eval() for the headline path"}


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import numpy as np  # noqa: F401
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  RTOD_BENCH_BACKEND=gloo lets two ranks share one GPU for a dry run of this path.
        backend = os.environ.get("RTOD_BENCH_BACKEND", "nccl")
        # (no device_id: the communicator is created lazily by the first collective on the device set above — the long-standing
        #  path; barriers name the device explicitly)
        dist.init_process_group(backend, rank=rank, world_size=world)
    barrier_kw = {"device_ids": [dev.index]} if world > 1 and os.environ.get("RTOD_BENCH_BACKEND", "nccl") == "nccl" else {}

    from realtimeobjectdetection_amd import synth
    from realtimeobjectdetection_amd.shard import FixedGather
    from realtimeobjectdetection_amd.util import write_results, write_results_async
    B, R = args.batch, args.res
    plan_opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in args.opt)
    model, ir, w, cfg_text = build_model(R, dev, B, args.precision, plan_opts)
    extra = [build_model(R, dev, B, args.precision, plan_opts)[0] for _ in range(max(0, args.inflight - 1))]
    models = [model] + extra
    for m_ in models:
        m_.overflow_check = "off"                       # the timed loops never read the range flag; checked once below
    fstreams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in extra]
    step_no = [0]
    # this rank's frame shard: frames [rank*B, (rank+1)*B) of the global synthetic stream
    x = torch.from_numpy(synth.synth_frames(B, R, seed=synth.FRAME_SEED + rank)).to(dev)
    CAP = 4096                                          # rows gathered per rank (fixed-capacity, no host sync)
    gather = FixedGather(CAP, dev) if world > 1 else None

    # write_results (3 small latency-bound launches, ~0.04 ms stand-alone) and the detection gather run on a second stream behind an
    # event, so that batch i's NMS overlaps batch i+1's first convolutions: the two batches are independent, every
    # step's work is enqueued inside the timed region and drained by the final device synchronize (--serial-nms: one stream)
    side = None if args.serial_nms else torch.cuda.Stream(device=dev)

    def post(y):
        rows, counts = write_results_async(y, 80, args.conf, args.nms, cap=CAP)
        if gather is not None:
            gather.gather(rows, counts, rank * B)       # global image index (detect.py:101-102) + RCCL all-gather
        return rows, counts

    def step(nm):
        i = step_no[0] % nm; step_no[0] += 1
        with torch.no_grad(), torch.cuda.stream(fstreams[i]):
            y = models[i](x)
            if side is None:
                rows, counts = post(y)
            else:
                side.wait_stream(fstreams[i])
                with torch.cuda.stream(side):
                    rows, counts = post(y)
                y.record_stream(side)
        return y, rows, counts

    def step_dropin(_nm):
        """The reference's calling sequence (detect.py:62-70): forward, then write_results with its host sync."""
        with torch.no_grad():
            y = model(x)
            det = write_results(y, 80, args.conf, args.nms)
        return y, det, None

    def timed_run(fn, nm):
        """W untimed + K timed steps; returns (seconds, last step's tensors)."""
        step_no[0] = 0
        for _ in range(args.warmup):
            fn(nm)
        if world > 1:
            dist.barrier(**barrier_kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn(nm)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(**barrier_kw)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    tiles_key = "%s_%d_b%d" % (args.precision, R, B)
    tiles_src = "autotune (this run)"
    if args.tiles and os.path.exists(args.tiles) and args.precision == "f16s3":
        table = json.load(open(args.tiles)).get(tiles_key)
        if table is not None:
            for m_ in models:
                m_.set_tiles(B, table)                  # no autotune launches: every process runs exactly these kernels
            tiles_src = "installed from " + os.path.basename(args.tiles)
    with torch.no_grad():                               # set-up, not a step: every plan's first forward autotunes its tiles
        for m_ in models:
            m_(x)
    torch.cuda.synchronize()
    if args.tiles and rank == 0 and args.precision == "f16s3" and tiles_src.startswith("autotune"):
        tabs = json.load(open(args.tiles)) if os.path.exists(args.tiles) else {}
        tabs[tiles_key] = model.get_tiles(B)
        os.makedirs(os.path.dirname(os.path.abspath(args.tiles)), exist_ok=True)
        json.dump(tabs, open(args.tiles, "w"))
    dt, (y, rows, counts) = timed_run(step, len(models))
    n_det, n_cand = [int(v) for v in counts[:2].tolist()]
    gathered = None
    if gather is not None:
        g = gather.compact()                            # content check of the last step's gather (host side, untimed)
        gathered = 0 if isinstance(g, int) else int(g.size(0))
    overflow = any(m_.overflowed() for m_ in models)    # split-f16 range guard over everything run so far

    def rate(dt_):
        return round(world * B * args.steps / dt_, 2), round(dt_ / args.steps * 1e3, 4)

    single = None
    if len(models) > 1:                                 # the strictly sequential figure beside the pipelined one
        dt1, _ = timed_run(step, 1)
        v, ms = rate(dt1)
        single = {"value": v, "unit": "frames/s", "ms_per_step": ms,
                  "note": "one batch in flight (forward i+1 starts after forward i); write_results_async on a second stream"}
    model.overflow_check = "write_results"
    dtd, (_, det, _) = timed_run(step_dropin, 1)
    model.overflow_check = "off"
    v, ms = rate(dtd)
    dropin = {"value": v, "unit": "frames/s", "ms_per_step": ms,
              "detections_last_step": 0 if isinstance(det, int) else int(det.size(0)),
              "note": "reference calling sequence: y = model(x); write_results(y, 80, conf, nms) -> new [D,8] tensor; one plan, "
                      "one stream, write_results' host sync (and the split-f16 range check) inside every step"}
    fps, ms_step = rate(dt)

    roof = None
    if rank == 0 and not args.no_roofline:
        roof, per_layer, groups = roofline_from_launches(model, x, max(1, min(args.steps, 10)))
        # HBM bytes per launch of that kernel: PMC counters need rocprofv3 around the process, so this run cannot measure them;
        # the figure comes from the committed passes of this same command with the tile table frozen (profiles/traffic.json,
        # tools/profile_round.sh + summarize_profiles.py: 2*FETCH_SIZE + WRITE_SIZE per launch, equal launch counts in both
        # passes); null for a kernel / workload that was not profiled
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path) and R == 608 and B == 8:
            try:
                tr = json.load(open(tr_path))
                roof["traffic"] = tr.get("kernels", tr).get(roof["kernel"])
                if roof["traffic"] is not None:
                    roof["traffic_source"] = tr.get("source", "profiles/traffic.json (rocprofv3 --pmc passes, not this run)")
            except Exception:
                pass
        if args.layers_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.layers_out)), exist_ok=True)
            json.dump({"per_launch": per_layer, "groups": groups}, open(args.layers_out, "w"), indent=1)
    n_inflight = len(models)
    other = None
    if rank == 0 and world == 1 and not args.no_extras and R == 608 and B == 8:
        del extra, models[1:]                           # their arenas are not needed any more
        other = extra_workloads(dev, args)
    as_run = None
    if rank == 0 and world == 1 and not args.no_extras and R == 608 and B == 8:
        as_run = as_run_bn_workload(dev, args)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg_text, w, R, B, args.conf, args.nms)

    if rank == 0:
        sched = ("%d batches in flight, write_results_async on %s" % (n_inflight, "the forward's stream" if args.serial_nms else "a second stream")
                 if n_inflight > 1 else "one batch in flight, write_results_async")
        line = {
            "metric": "frames/sec YOLOv3 %dx%d bs=%d (Darknet.forward + write_results; %s)" % (R, R, B, sched),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f16x2-split (3 MFMA products, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "YOLOv3 cfg %dx%d batch=%d per GPU, %s MFMA conv + fused head + GPU NMS (BASELINE configs[%d])"
                                   % (R, R, B, "exact-fp32" if args.precision == "fp32" else "split-f16", 2 if R == 608 else 1),
                       "precision": args.precision, "plan_options": plan_opts, "tiles": tiles_src,
                       "frames_per_step": world * B, "parallelism": "frame-shard x%d" % world,
                       "in_flight_batches": n_inflight, "write_results": "async (device-side counts, capacity %d rows)" % CAP,
                       "write_results_stream": "same" if args.serial_nms else "second stream",
                       "conf": args.conf, "nms": args.nms, "detections_last_step": n_det, "candidates_last_step": n_cand,
                       "f16_range_overflow": bool(overflow),
                       "conv_gflop_per_frame": round(ir.conv_flops / 1e9, 3),
                       "whole_path_tflops": round(fps * ir.conv_flops / 1e12, 2),
                       "whole_path_frac_fp32_mfma_peak": round(fps * ir.conv_flops / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4)},
        }
        if gathered is not None:
            line["config"]["gathered_detections_last_step"] = gathered
        if single is not None:
            line["single_stream"] = single
        line["dropin_api"] = dropin
        if roof is not None:
            line["roofline"] = roof
        if other is not None:
            line["other_configs"] = other
        if as_run is not None:
            line["as_run_bn"] = as_run
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))                     # no GPU call has been made in this process
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        args.gpus = int(os.environ["WORLD_SIZE"])
    run_rank(args)


if __name__ == "__main__":
    main()
