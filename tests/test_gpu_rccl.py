"""Every collective site of the package through the `nccl` backend (= RCCL) on device tensors.

One MI355X per test box, and RCCL refuses two ranks on one device, so this is a ONE-rank communicator:
`init_process_group("nccl", world_size=1, device_id=cuda:0)` creates a real RCCL communicator and every
`all_reduce` / `all_gather_into_tensor` below is an RCCL call on RCCL's own stream, ordered against the
stream the kernels run on.  What it proves: the backend initialises, accepts the int64 / fp64 views of the
uint8 workspaces the radix selects all-reduce, the stream hand-over between pxf kernels and RCCL is right
(a missing wait would leave histograms unreduced or half-written), and `group=` results equal the
un-grouped ones.  What it cannot prove: anything about a second rank (tests/test_dist_cpu.py and
tests/test_gpu_dist.py cover world 2 / 3 over gloo).  The child is a spawned process: the communicator
is created before anything else touches the GPU in it."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _trace(n, seed):
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import conicsolve as con
    r0, z0, L = 220., 8400., 100.
    rays = sources.annulus(float(con.primrad(z0, r0, z0)), float(con.primrad(z0 + L, r0, z0)), n,
                           rng="philox", seed=seed)
    tran.transform(rays, 0, 0, -(z0 + 2 * L), 0, 1e-4, 0)
    surf.wolterprimary(rays, r0, z0)
    tran.reflect(rays)
    surf.woltersecondary(rays, r0, z0)
    tran.reflect(rays)
    tran.transform(rays, 0, 0, 3., 0, 0, 0)
    surf.flat(rays)
    return rays


def _child(port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    res = {"backend": dist.get_backend()}
    try:
        import pyxfocus_amd.analyses as anal
        import pyxfocus_amd.surfaces as surf
        import pyxfocus_amd.transformations as tran
        from pyxfocus_amd import dist as pdist
        rays = _trace(n, 91)
        w = 1. + .5 * torch.sin(torch.arange(n, dtype=torch.float64, device=dev) * .37)
        for name, fn in (("centroid", anal.centroid), ("rmsCentroid", anal.rmsCentroid), ("rmsX", anal.rmsX),
                         ("rmsY", anal.rmsY), ("hpd", anal.hpd), ("analyticImagePlane", anal.analyticImagePlane),
                         ("analyticXPlane", anal.analyticXPlane), ("analyticYPlane", anal.analyticYPlane)):
            for wt, tag in ((None, ""), (w, "_w")):
                res[name + tag] = (fn(rays, weights=wt), fn(rays, weights=wt, group=True))
        # a data-dependent vignette, then the config-5 tail: steer, focus (frame moves), gathers
        keep = rays[1] > rays[1].median()               # about half survive
        pair = []
        for grp in (None, True):
            r2 = tran.vignette(rays, keep)
            coords = tran.newCoords()
            kw = {} if grp is None else {"group": grp}
            tran.steerX(r2, coords=coords, **kw)
            tran.steerY(r2, coords=coords, **kw)
            dzy = surf.focusY(r2, weights=w[keep], coords=coords, **kw)
            dzx = surf.focusX(r2, coords=coords, **kw)
            dzi = surf.focusI(r2, coords=coords, **kw)
            pair.append((dzy, dzx, dzi, np.array(coords), r2[1].cpu().numpy(), r2[2].cpu().numpy()))
        res["tail"] = pair
        x, y, wk = r2[1], r2[2], w[keep].contiguous()
        torch.cuda.synchronize()
        g_eq = pdist.all_gather_focal([x, y, wk])                                     # one collective per column
        g_st = pdist.all_gather_focal([x, y, wk], scratch_bytes=1 << 16, staged=True)   # chunked through the staging buffer
        res["gather"] = [[c.cpu().numpy() for c in g] for g in (g_eq, g_st)] + [[c.cpu().numpy() for c in (x, y, wk)]]
        # int64 counts path with an EMPTY shard
        e = torch.empty(0, dtype=torch.float64, device=dev)
        res["gather_empty"] = [int(c.shape[0]) for c in pdist.all_gather_focal([e, e])]
        t = torch.arange(5, dtype=torch.float64, device=dev)
        res["sums"] = pdist.all_reduce_sums(t.clone()).cpu().numpy()                # world 1: returns without a call
        tt = t.clone()
        dist.all_reduce(tt)
        dist.barrier()
        res["raw_allreduce"] = tt.cpu().numpy()
        q.put(res)
    except Exception as e:                                                              # pragma: no cover
        import traceback
        q.put({"error": traceback.format_exc()})
    finally:
        dist.destroy_process_group()


def test_every_collective_site_over_rccl():
    import torch.multiprocessing as mp
    n = 300_001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_child, args=(_free_port(), n, q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert "error" not in res, res.get("error")
    assert p.exitcode == 0
    assert res["backend"] == "nccl"
    for k, v in res.items():
        if isinstance(v, tuple) and len(v) == 2 and k not in ("tail",):
            a, b = np.asarray(v[0], dtype=float), np.asarray(v[1], dtype=float)
            # one rank: the all-reduce is the identity, so group= must return the very same numbers
            # (hpd takes the one-pass bracket when un-grouped and the six-pass select when grouped: same element)
            assert np.array_equal(a, b), (k, v)
    (a, b) = res["tail"]
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], (a[:3], b[:3])
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    g_eq, g_st, want = res["gather"]
    for u, v, c in zip(g_eq, g_st, want):
        assert np.array_equal(u, c) and np.array_equal(v, c)
    assert len(want[0]) not in (0, n)
    assert res["gather_empty"] == [0, 0]
    assert np.array_equal(res["sums"], np.arange(5.)) and np.array_equal(res["raw_allreduce"], np.arange(5.))
