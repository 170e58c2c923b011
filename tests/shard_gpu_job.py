"""Two-rank frame-shard job on DEVICE tensors (both ranks on cuda:0, gloo), started by tests/conftest.py BEFORE the pytest
process makes its first GPU call (on the pool an exec from a GPU-initialised process is refused), checked later by
tests/test_shard_gpu.py.  This launcher never touches the GPU itself.

    python tests/shard_gpu_job.py <outdir>

Stage 1: two `tests/shard_gpu_job.py --rank` workers: YOLOv3 416x416, 4 frames per rank, 3 steps of forward +
write_results_async + shard.FixedGather.gather (device tensors), then FixedGather.compact; each rank saves what it gathered,
its own write_results rows and the time its steps took.
Stage 2: `bench.py --gpus 2` itself (its self-launch, RTOD_BENCH_BACKEND=gloo) on the same frames: the JSON line.
Writes <outdir>/done.json last."""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES, B, STEPS, CAP = 416, 4, 3, 1024


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_main(out):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    from realtimeobjectdetection_amd import synth
    from realtimeobjectdetection_amd.shard import FixedGather
    from realtimeobjectdetection_amd.util import write_results, write_results_async
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = bench.build_model(RES, dev, B, "f16s3")[0]
        x = torch.from_numpy(synth.synth_frames(B, RES, seed=synth.FRAME_SEED + rank)).to(dev)
        gather = FixedGather(CAP, dev)
        t_fwd, t_gather = [], []
        with torch.no_grad():
            model(x)                                                # autotune
            for _ in range(STEPS):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                y = model(x)
                rows, counts = write_results_async(y, 80, 0.6, 0.5, cap=CAP)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                g_rows, g_meta = gather.gather(rows, counts, rank * B)
                assert g_rows.is_cuda and g_meta.is_cuda
                torch.cuda.synchronize(); t2 = time.perf_counter()
                t_fwd.append(t1 - t0); t_gather.append(t2 - t1)
            got = gather.compact()
            local = write_results(y, 80, 0.6, 0.5)
        np.savez(os.path.join(out, "rank%d.npz" % rank),
                 gathered=np.zeros((0, 8), np.float32) if isinstance(got, int) else got.cpu().numpy(), gathered_is_zero=int(isinstance(got, int)),
                 local=np.zeros((0, 8), np.float32) if isinstance(local, int) else local.cpu().numpy(), local_is_zero=int(isinstance(local, int)),
                 meta=gather.meta.cpu().numpy(), t_forward_ms=1e3 * np.asarray(t_fwd), t_gather_ms=1e3 * np.asarray(t_gather))
    finally:
        dist.destroy_process_group()


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    status = {"stage1_rc": None, "stage2_rc": None}
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        log = open(os.path.join(out, "rank%d.log" % r), "w")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", out], env=env, stdout=log, stderr=subprocess.STDOUT))
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=600))
        except subprocess.TimeoutExpired:
            p.kill(); rcs.append(-9)                        # exact PID this process started
    status["stage1_rc"] = rcs
    if all(rc == 0 for rc in rcs):
        env = dict(base, RTOD_BENCH_BACKEND="gloo")
        with open(os.path.join(out, "bench.json"), "w") as fo, open(os.path.join(out, "bench.log"), "w") as fe:
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--res", str(RES), "--batch", str(B), "--steps", str(STEPS),
                                    "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-roofline"], env=env, stdout=fo, stderr=fe, timeout=900)
                status["stage2_rc"] = r.returncode
            except subprocess.TimeoutExpired:
                status["stage2_rc"] = -9
    json.dump(status, open(os.path.join(out, "done.json"), "w"))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--rank":
        rank_main(sys.argv[2])
    else:
        main()
