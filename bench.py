#!/usr/bin/env python
"""bench.py — frames/s of the YOLOv3 hot path (Darknet.forward + write_results) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--res 608] [--batch 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
``Darknet.forward`` (78 HIP launches) + ``write_results`` (filter, sort, NMS; 3 launches) and, for
N > 1, the fixed-capacity RCCL all-gather of the detections.  One process per GPU; frames are
sharded (weak scaling: every rank runs its own ``--batch`` frames), no collective on the data path
except that final gather.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      dominant kernel = the conv implicit-GEMM instantiation with the largest total time.
                achieved = algorithmic conv FLOPs of its launches / their summed duration, measured
                with a hipEvent pair around every launch on the launch stream (rtod_forward_timed)
                in an instrumented replay of the timed steps right after the timed region.
                peak = 157.3 TFLOP/s (exact-fp32 MFMA, MI355X_MICROARCH.md).
  cpu_baseline  the oracle (oracle/darknet_ref.py: the reference's PyTorch-CPU op sequence + NMS)
                timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0        # dense f16/bf16 MFMA; the split kernel issues 3 products per algorithmic MAC
HBM_PEAK_GBS = 8000.0


def build_model(res, device, max_batch, precision):
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


def cpu_baseline(cfg_text, w, res, batch, conf, nms, budget_s=25.0):
    """Oracle forward + write_results on the host cores, bounded sample."""
    from realtimeobjectdetection_amd import synth
    from oracle import darknet_ref as O
    cores = host_cpu_share()
    torch.set_num_threads(cores)
    ref = O.RefDarknet(cfg_text, res)
    ref.load_weight_stream(w)
    x = torch.from_numpy(synth.synth_frames(batch, res))
    with torch.no_grad():
        t0 = time.perf_counter()
        y = ref.forward(x)                      # warm-up (also sizes the sample)
        warm = time.perf_counter() - t0
        iters = int(max(1, min(4, budget_s // max(warm, 1e-3) - 1)))
        t0 = time.perf_counter()
        for _ in range(iters):
            y = ref.forward(x)
            O.write_results(y, 80, conf, nms)
        dt = time.perf_counter() - t0
    return {"value": round(batch * iters / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d x (yolov3 %dx%d batch %d forward + write_results), torch %s CPU ops, %d threads"
                      % (iters, res, res, batch, torch.__version__, cores)}


def rocprof_kernel_name(variant, epi):
    """Exact kernel instantiation name rocprofv3 prints for a conv launch (built next to the launch switch in csrc)."""
    import ctypes as C
    from realtimeobjectdetection_amd import _ffi
    buf = C.create_string_buffer(256)
    _ffi.check(_ffi.lib().rtod_conv_kernel_name(int(variant), int(epi), buf, 256))
    return buf.value.decode()


def roofline_from_launches(model, x, steps):
    """Instrumented replay: hipEvent pair around every launch; group conv launches by tile variant."""
    import ctypes as C
    from realtimeobjectdetection_amd import _ffi
    infos = model.launch_infos()
    B = x.size(0)
    tot = np.zeros(len(infos), dtype=np.float64)
    for _ in range(steps):
        _, ms = model.forward_timed(x)
        tot += ms
    tot /= steps
    lib = _ffi.lib()
    groups = {}
    for li, ms in zip(infos, tot):
        if li.kind != 0 or li.flops_per_frame == 0:      # non-conv launches; 1x1 convs fused into the previous conv's epilogue
            continue
        # group by the exact kernel instantiation rocprofv3 reports (tile variant + epilogue)
        name = lib.rtod_conv_variant_name(li.variant).decode()
        epi = 2 if li.fused_decode else ((4 if li.fused_residual else 3) if li.fused_pointwise else (1 if li.fused_residual else 0))
        kname = rocprof_kernel_name(li.variant, epi)
        g = groups.setdefault(kname, {"ms": 0.0, "flops": 0.0, "launches": 0, "tile": name})
        g["ms"] += float(ms); g["flops"] += float(li.flops_per_frame) * B; g["launches"] += 1
    dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
    name, g = dom
    achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
    conv_ms = sum(v["ms"] for v in groups.values())
    per_layer = [{"layer": li.layer, "kind": li.kind, "variant": li.variant, "k": li.ksize, "s": li.stride, "cin": li.cin,
                  "cout": li.cout, "hout": li.hout, "ms": round(float(ms), 5),
                  "tflops": round(float(li.flops_per_frame) * B / (float(ms) * 1e-3) / 1e12, 2) if li.kind == 0 and ms > 0 else None,
                  "gbs": round((float(li.bytes_per_frame) * B + li.weight_bytes) / (float(ms) * 1e-3) / 1e9, 1) if ms > 0 else None}
                 for li, ms in zip(infos, tot)]
    split = "f16s3" in name
    peak = round(PEAK_F16_MFMA_TFLOPS / 3.0, 1) if split else PEAK_FP32_MFMA_TFLOPS
    roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "peak_note": ("algorithmic FLOPs; each MAC issues 3 f16 MFMA products, peak = 2500/3" if split
                          else "exact-fp32 MFMA v_mfma_f32_32x32x2_f32"),
            "launches_per_step": g["launches"], "avg_launch_ms": round(g["ms"] / g["launches"], 5),
            "flops_per_launch": g["flops"] / g["launches"],
            "all_conv_tflops": round(sum(v["flops"] for v in groups.values()) / (conv_ms * 1e-3) / 1e12, 2),
            "forward_launch_ms_sum": round(float(tot.sum()), 4)}
    return roof, per_layer, groups


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--res", type=int, default=608)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--conf", type=float, default=0.6)
    ap.add_argument("--nms", type=float, default=0.5)
    ap.add_argument("--precision", default=os.environ.get("RTOD_PRECISION", "f16s3"), choices=["fp32", "f16s3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-nms", action="store_true", help="run write_results on the forward's stream (no overlap with the next batch)")
    ap.add_argument("--skew-frames", type=int, default=4, help="frames of the untimed forward that offsets the second stream (0: none)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight: steps alternate over N plans (own activation arena) on N HIP streams, so one batch's "
                         "partial rounds, prologues and epilogues overlap the other's kernels; 1 = strictly sequential forwards")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--layers-out", default="", help="write the per-launch table (JSON) here")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
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
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from realtimeobjectdetection_amd import synth
    from realtimeobjectdetection_amd.util import write_results_async
    B, R = args.batch, args.res
    model, ir, w, cfg_text = build_model(R, dev, B, args.precision)
    extra = [build_model(R, dev, B, args.precision)[0] for _ in range(max(0, args.inflight - 1))]
    models = [model] + extra
    fstreams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in extra]
    step_no = [0]
    # this rank's frame shard: frames [rank*B, (rank+1)*B) of the global synthetic stream
    x = torch.from_numpy(synth.synth_frames(B, R, seed=synth.FRAME_SEED + rank)).to(dev)
    CAP = 4096                                          # rows gathered per rank (fixed-capacity, no host sync)
    if world > 1:
        g_rows = torch.empty((world * CAP, 8), dtype=torch.float32, device=dev)
        g_counts = torch.empty((world * 2,), dtype=torch.int32, device=dev)

    # write_results (3 small latency-bound launches, ~0.1 ms) and the detection gather run on a second stream behind an
    # event, so that batch i's NMS overlaps batch i+1's first convolutions: the two batches are independent, every
    # step's work is enqueued inside the timed region and drained by the final device synchronize (--serial-nms: one stream)
    side = None if args.serial_nms else torch.cuda.Stream(device=dev)

    def post(y):
        rows, counts = write_results_async(y, 80, args.conf, args.nms, cap=CAP)
        if world > 1:
            rows[:, 0].add_(float(rank * B))        # global image index (detect.py:101-102)
            dist.all_gather_into_tensor(g_rows, rows)
            dist.all_gather_into_tensor(g_counts, counts[:2].contiguous())
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

    def timed_run(nm):
        """W untimed + K timed steps with `nm` batches in flight; returns (seconds, last step's tensors)."""
        step_no[0] = 0
        for _ in range(args.warmup):
            step(nm)
        if nm == 2 and args.skew_frames > 0:
            # untimed: one forward of half a batch on the second stream puts it about half a forward behind the first, so the
            # two batches sit in different parts of the network (one in the HBM-bound early layers while the other is in the
            # MFMA-bound deep ones) instead of running the same layer side by side: +1.2 % (1984 vs 1961 frames/s, same box)
            with torch.no_grad(), torch.cuda.stream(fstreams[1]):
                models[1](x[:min(args.skew_frames, B)].contiguous())
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(nm)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    with torch.no_grad():                               # set-up, not a step: every plan's first forward autotunes its tiles
        for m_ in models:
            m_(x)
    torch.cuda.synchronize()
    dt, (y, rows, counts) = timed_run(len(models))
    single = None
    if len(models) > 1:                                 # the strictly sequential figure beside the pipelined one
        dt1, _ = timed_run(1)
        single = {"value": round(world * B * args.steps / dt1, 2), "unit": "frames/s", "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                  "note": "one batch in flight (forward i+1 starts after forward i; write_results still overlapped)"}
    fps = world * B * args.steps / dt
    n_det, n_cand = [int(v) for v in counts[:2].tolist()]

    roof = None
    if rank == 0 and not args.no_roofline:
        roof, per_layer, groups = roofline_from_launches(model, x, max(1, min(args.steps, 10)))
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (profiles/traffic.json,
        # written by tools/summarize_profiles.py: 2*FETCH_SIZE + WRITE_SIZE); null if it was not profiled
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path) and R == 608 and B == 8:
            try:
                roof["traffic"] = json.load(open(tr_path)).get(roof["kernel"])
            except Exception:
                pass
        if args.layers_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.layers_out)), exist_ok=True)
            json.dump({"per_launch": per_layer, "groups": groups}, open(args.layers_out, "w"), indent=1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg_text, w, R, B, args.conf, args.nms)

    if rank == 0:
        line = {
            "metric": "frames/sec YOLOv3 %dx%d bs=%d (Darknet.forward + write_results)" % (R, R, B),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f16x2-split (3 MFMA products, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "YOLOv3 cfg %dx%d batch=%d per GPU, %s MFMA conv + fused head + GPU NMS (BASELINE configs[%d])"
                                   % (R, R, B, "exact-fp32" if args.precision == "fp32" else "split-f16", 2 if R == 608 else 1),
                       "precision": args.precision,
                       "frames_per_step": world * B, "parallelism": "frame-shard x%d" % world,
                       "in_flight_batches": len(models), "write_results_stream": "same" if args.serial_nms else "second stream",
                       "conf": args.conf, "nms": args.nms, "detections_last_step": n_det, "candidates_last_step": n_cand,
                       "conv_gflop_per_frame": round(ir.conv_flops / 1e9, 3),
                       "whole_path_tflops": round(fps * ir.conv_flops / 1e12, 2),
                       "whole_path_frac_fp32_mfma_peak": round(fps * ir.conv_flops / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4)},
        }
        if single is not None:
            line["single_stream"] = single
        if roof is not None:
            line["roofline"] = roof
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
