"""The N > 1 path on DEVICE tensors (SURVEY.md section 8 e; reference: detect.py:101-102 shifts the image column per batch,
detect.py:177-183 is its DataParallel wrapper): two ranks on one MI355X (both on cuda:0, gloo — one GPU per lease, so RCCL
itself cannot run here), FixedGather on GPU tensors, and bench.py's own rank code through its self-launch.

The job (tests/shard_gpu_job.py) is started by conftest.py at the end of collection, before this process makes a GPU call."""
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _wait(job, timeout=1500):
    t0 = time.time()
    done = os.path.join(job["out"], "done.json")
    while not os.path.exists(done):
        if job["proc"].poll() is not None and not os.path.exists(done):
            break
        if time.time() - t0 > timeout:
            job["proc"].kill()                                   # exact PID conftest started
            pytest.fail("two-rank job did not finish in %d s" % timeout)
        time.sleep(1.0)
    assert os.path.exists(done), "two-rank job exited without a result: rc %s" % job["proc"].poll()
    return json.load(open(done))


def test_two_ranks_on_device_tensors(shard_gpu_job):
    if shard_gpu_job is None:
        pytest.skip("the two-rank job is started at collection time on a GPU box only")
    out = shard_gpu_job["out"]
    status = _wait(shard_gpu_job)
    logs = "".join(open(os.path.join(out, f)).read()[-1500:] for f in ("rank0.log", "rank1.log") if os.path.exists(os.path.join(out, f)))
    assert status["stage1_rc"] == [0, 0], logs
    r0, r1 = (np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(2))
    B = 4
    # every rank sees the same gathered rows: rank 0's write_results rows, then rank 1's with the image column shifted by its first frame
    assert not int(r0["local_is_zero"]) and not int(r1["local_is_zero"]) and len(r0["local"]) > 0 and len(r1["local"]) > 0
    shifted = r1["local"].copy(); shifted[:, 0] += B
    want = np.concatenate([r0["local"], shifted], 0)
    for r in (r0, r1):
        assert not int(r["gathered_is_zero"])
        assert np.array_equal(r["gathered"], want)
        assert r["meta"].reshape(2, 2)[:, 0].tolist() == [len(r0["local"]), len(r1["local"])]
    assert set(np.unique(want[:, 0]).astype(int)) <= set(range(2 * B)) and want[:, 0].max() >= B
    # bench.py's own N = 2 code path on the same frames: one JSON line from rank 0, the same number of gathered rows
    assert status["stage2_rc"] == 0, open(os.path.join(out, "bench.log")).read()[-3000:]
    line = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frames_per_step"] == 2 * B
    assert line["config"]["gathered_detections_last_step"] == len(want)
    assert line["value"] > 0 and line["steps"] == 3
    print("two ranks on one GPU (gloo): forward+write_results %.2f ms, gather + sync %.2f ms per step; bench.py --gpus 2: %.1f frames/s, %.2f ms per step"
          % (float(np.median(r0["t_forward_ms"])), float(np.median(r0["t_gather_ms"])), line["value"], line["ms_per_step"]))
