"""Frame sharding across the GPUs of one node (SURVEY.md §8 e; reference: none — its
``nn.DataParallel`` wrapper is broken for detection, detect.py:177-183).

Frames are independent units (eval-mode BN), so rank r of R runs ``Darknet.forward`` +
``write_results`` on its own contiguous frame range with replicated weights and no collective on
the data path.  The only exchange is the final detection gather: per-rank counts, then the rows
padded to the maximum count, both as ``all_gather`` (RCCL over xGMI when the backend is "nccl";
payload is tens of KB, latency-bound).  Concatenation by rank preserves the single-GPU output
order (image-major) after the image column is shifted by the rank's first frame, exactly what
detect.py:101-102 does per batch.
"""
import torch
import torch.distributed as dist


def frame_range(n_frames: int, rank: int, world: int):
    """Contiguous split; the first ``n_frames % world`` ranks take one extra frame."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_detections(local_rows, frame_offset: int, device=None, group=None):
    """All-gather per-rank detection rows into the global ``[D,8]`` tensor (or int 0).

    ``local_rows`` is what ``write_results`` returned on this rank: ``[d,8]`` tensor or int 0.
    Works with any initialised backend (gloo on CPU tensors in tests, nccl/RCCL on GPU).
    """
    world = dist.get_world_size(group)
    if isinstance(local_rows, int):
        if device is None:
            device = torch.device("cpu")
        rows = torch.zeros((0, 8), dtype=torch.float32, device=device)
        had = 0
    else:
        rows = local_rows.clone()
        device = rows.device
        rows[:, 0] += float(frame_offset)
        had = 1
    meta = torch.tensor([rows.size(0), had], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    any_out = any(int(m[1]) for m in metas)
    cap = max(counts) if counts else 0
    if cap == 0:
        return torch.zeros((0, 8), dtype=torch.float32, device=device) if any_out else 0
    padded = torch.zeros((cap, 8), dtype=torch.float32, device=device)
    padded[:rows.size(0)] = rows
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


class FixedGather:
    """Fixed-capacity detection gather without a host synchronisation (what ``bench.py`` runs every step).

    ``gather_detections`` sizes its second all-gather from the first one's counts: one host round trip per batch.  A
    detector that keeps several batches in flight instead gathers a fixed ``cap`` rows per rank plus the counts and
    compacts on the host only when it consumes the result.  Buffers are allocated once; ``gather`` enqueues two
    ``all_gather_into_tensor`` calls (RCCL over xGMI with the nccl backend: ``cap*32`` bytes per rank, latency-bound) on
    the current stream and returns device tensors; ``compact`` (host side, after a sync) yields exactly what
    ``gather_detections`` returns: rows in single-process order (image-major) or the int 0.
    """

    def __init__(self, cap: int, device, group=None):
        self.cap, self.group = int(cap), group
        self.world = dist.get_world_size(group)
        self.rows = torch.empty((self.world * self.cap, 8), dtype=torch.float32, device=device)
        self.meta = torch.empty((self.world * 2,), dtype=torch.int32, device=device)

    def gather(self, local_rows: torch.Tensor, local_counts: torch.Tensor, frame_offset: int):
        """``local_rows`` ``[cap,8]`` and ``local_counts`` (``[0]`` = valid rows, ``[1]`` = candidates) as returned by
        ``util.write_results_async(..., cap=cap)``.  The image column is shifted in place by this rank's first frame
        (detect.py:101-102 does the same per batch)."""
        assert local_rows.shape == (self.cap, 8)
        local_rows[:, 0].add_(float(frame_offset))
        dist.all_gather_into_tensor(self.rows, local_rows, group=self.group)
        dist.all_gather_into_tensor(self.meta, local_counts[:2].contiguous(), group=self.group)
        return self.rows, self.meta

    def compact(self):
        """Host side: ``[D,8]`` rows in rank (= image) order, an empty tensor when candidates existed but none survived,
        or 0 when no rank had a candidate (the reference's conventions, src/util.py:343-346)."""
        meta = self.meta.view(self.world, 2).tolist()
        if any(int(d) > self.cap for d, _ in meta):
            raise RuntimeError("FixedGather: a rank produced %d detections, capacity is %d" % (max(int(d) for d, _ in meta), self.cap))
        if not any(int(c) for _, c in meta):
            return 0
        parts = [self.rows[r * self.cap: r * self.cap + int(meta[r][0])] for r in range(self.world)]
        return torch.cat(parts, 0)
