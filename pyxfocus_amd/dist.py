"""Multi-GPU layer: one process per MI355X, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

Rays are independent, so the trace shards trivially by GLOBAL ray index and needs NO collective
(SURVEY 8e).  The only exchanges are at the focal plane:

* scalar metrics (centroid, rmsCentroid, analytic best focus, steer) -- the twelve partial sums of
  ``pxf_moments`` are all-reduced (96 bytes) by ``analyses.*(..., group=True)``: no ray moves;
* ``all_gather_focal`` -- the north-star's "final focal-plane all-gather feeding analyses.py" for
  analyses that need every ray (hpd, images).  Shards have different lengths after ``vignette``:
  counts are exchanged first, columns are padded to the longest shard into ONE (ncols x maxlen)
  buffer so a single ``all_gather_into_tensor`` moves all columns (fewer, larger collectives: xGMI
  is 7 point-to-point links per GPU, per-call latency dominates small transfers), then trimmed.

Everything here also runs on CPU tensors with the gloo backend (tests/test_dist_cpu.py).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank=None, world=None):
    """Contiguous global-index range [lo, hi) owned by `rank`: sizes differ by at most one."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    base, rem = divmod(int(total), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_focal(cols, group=None):
    """cols: list of equal-length 1-D tensors (this rank's surviving rays).  Returns a list of the
    same columns concatenated over ranks in rank order (i.e. global ray order)."""
    world = dist.get_world_size(group)
    n = cols[0].shape[0]
    dev, dt = cols[0].device, cols[0].dtype
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts = counts.tolist()
    maxlen = max(counts)
    send = torch.zeros((len(cols), maxlen), dtype=dt, device=dev)
    for i, c in enumerate(cols):
        send[i, :n] = c
    recv = torch.empty((world, len(cols), maxlen), dtype=dt, device=dev)
    # flat views: RCCL and gloo both accept (world*k,) <- (k,)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    return [torch.cat([recv[r, i, :counts[r]] for r in range(world)]) for i in range(len(cols))]


def all_reduce_sums(t, group=None):
    """In-place SUM all-reduce of a small tensor of partial sums."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
