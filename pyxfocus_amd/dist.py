"""Multi-GPU layer: one process per MI355X, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

Rays are independent, so the trace shards trivially by GLOBAL ray index and needs NO collective
(SURVEY 8e).  The only exchanges are at the focal plane:

* scalar metrics (centroid, rmsCentroid, analytic best focus, steer) -- the twelve partial sums of
  ``pxf_moments`` are all-reduced (96 bytes) by ``analyses.*(..., group=True)``: no ray moves;
* ``all_gather_focal`` -- the north-star's "final focal-plane all-gather feeding analyses.py" for
  analyses that need every ray (hpd, images).  Shards have different lengths after ``vignette``:
  counts are exchanged first; equal shards are gathered column by column straight into the final
  tensors, uneven ones through a bounded staging buffer into preallocated outputs (peak extra
  memory = the gathered columns + 1 GiB; a config-5 column is 64 GB gathered);
* ``surfaces.focus*(group=)``, ``transformations.steerX/Y(group=)`` -- the frame moves by the
  GLOBAL best-focus shift / mean tilt, identical on every rank.

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


GATHER_SCRATCH_BYTES = 1 << 30      # bound on the staging memory of an uneven gather (send + world x recv)


def all_gather_focal(cols, group=None, scratch_bytes=None, staged=None):
    """cols: list of equal-length 1-D tensors (this rank's surviving rays).  Returns a list of the
    same columns concatenated over ranks in rank order (i.e. global ray order).

    Memory: the gathered columns themselves plus at most `scratch_bytes` (default 1 GiB) -- at
    config-5 size a gathered fp64 column is 64 GB (8 ranks x 1e9 survivors), so nothing may be held
    twice.  Counts are exchanged first (world x int64).  Equal counts (no vignetting, or a
    sharded bundle that divides evenly): each column is gathered by ONE collective straight into
    its final tensor, no padding, no copy.  Uneven counts: every output column is allocated at its
    final size and filled by per-rank offsets through a fixed staging buffer of `scratch_bytes`
    ((world+1) equal chunks: one send slot, world receive slots); a rank whose slice fills the
    chunk sends from its column directly.  `staged=True` takes the staged path for equal shards too (the
    bounded-memory route regardless of the count table; also how a one-rank run exercises it)."""
    world = dist.get_world_size(group)
    n = int(cols[0].shape[0])
    dev, dt = cols[0].device, cols[0].dtype
    for c in cols:
        if c.dim() != 1 or int(c.shape[0]) != n or c.dtype != dt or c.device != dev:
            raise ValueError("all_gather_focal: columns must be 1-D, equal length, one dtype, one device")
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts = [int(v) for v in counts.tolist()]
    total, maxlen = sum(counts), max(counts)
    out = [torch.empty(total, dtype=dt, device=dev) for _ in cols]
    if total == 0:
        return out
    if min(counts) == maxlen and not staged:                      # equal shards: straight into place
        for o, c in zip(out, cols):
            dist.all_gather_into_tensor(o, c.contiguous(), group=group)
        return out
    offs = [0] * world
    for r in range(1, world):
        offs[r] = offs[r - 1] + counts[r - 1]
    esz = out[0].element_size()
    budget = GATHER_SCRATCH_BYTES if scratch_bytes is None else int(scratch_bytes)
    chunk = max(1, min(maxlen, budget // (esz * (world + 1))))
    stage = torch.empty((world + 1) * chunk, dtype=dt, device=dev)
    slot, recv_all = stage[:chunk], stage[chunk:]
    for o, c in zip(out, cols):
        c = c.contiguous()
        for k0 in range(0, maxlen, chunk):
            L = min(chunk, maxlen - k0)
            mine = min(max(n - k0, 0), L)
            if mine == L:
                send = c[k0:k0 + L]
            else:                                                    # short or empty slice: pad (never read back)
                send = slot[:L]
                if mine:
                    send[:mine].copy_(c[k0:k0 + mine])
            recv = recv_all[:world * L]
            dist.all_gather_into_tensor(recv, send, group=group)
            for r in range(world):
                m = min(max(counts[r] - k0, 0), L)
                if m:
                    o[offs[r] + k0:offs[r] + k0 + m].copy_(recv[r * L:r * L + m])
    return out


def all_reduce_sums(t, group=None):
    """In-place SUM all-reduce of a small tensor of partial sums."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
