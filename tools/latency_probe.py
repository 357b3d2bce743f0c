#!/usr/bin/env python3
"""Dev tool: host-side cost per API call and chain latency for small bundles (eager / fused / graph)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pyxfocus_amd as pxf
import pyxfocus_amd.sources as sources
import pyxfocus_amd.surfaces as surf
import pyxfocus_amd.transformations as tran
from pyxfocus_amd import conicsolve as con

R0, Z0, L = 220., 8400., 100.
rin, rout = float(con.primrad(Z0, R0, Z0)), float(con.primrad(Z0 + L, R0, Z0))


def chain(r):
    tran.transform(r, 0, 0, -(Z0 + 2 * L), 0, 0, 0)
    surf.wolterprimary(r, R0, Z0)
    tran.reflect(r)
    surf.woltersecondary(r, R0, Z0)
    tran.reflect(r)
    surf.flat(r)


for n in (1000, 10_000, 100_000, 1_000_000, 10_000_000):
    src = sources.annulus(rin, rout, n, rng="philox", seed=1)
    rays = [c.clone() for c in src]
    reps = 200 if n <= 1_000_000 else 20

    def reset():
        for a, b in zip(rays, src):
            a.copy_(b)

    out = []
    for mode in ("eager", "fused", "graph"):
        reset()
        if mode == "graph":
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    with pxf.fused():
                        chain(rays)
            torch.cuda.current_stream().wait_stream(s)
            run = g.replay
        elif mode == "fused":
            def run():
                with pxf.fused():
                    chain(rays)
        else:
            run = lambda: chain(rays)
        run(); torch.cuda.synchronize()
        # every repetition must start from source rays (re-tracing traced rays is garbage geometry
        # that runs the Newton loops into their cap); time the resets alone and subtract
        t0 = time.perf_counter()
        for _ in range(reps):
            reset()
        torch.cuda.synchronize()
        t_reset = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            reset()
            t1 = time.perf_counter()
            run()
        t_issue_last = time.perf_counter() - t1
        torch.cuda.synchronize()
        t_total = (time.perf_counter() - t0) / reps - t_reset
        out.append("%s issue %.1f us total %.1f us" % (mode, t_issue_last * 1e6, t_total * 1e6))
    print("n=%-9d %s" % (n, " | ".join(out)), flush=True)
