#!/usr/bin/env python3
"""Dev tool: host cost of recording + flushing a six-call fused program on a tiny bundle."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyxfocus_amd as pxf  # noqa: E402
import pyxfocus_amd.sources as sources  # noqa: E402
import pyxfocus_amd.surfaces as surf  # noqa: E402
import pyxfocus_amd.transformations as tran  # noqa: E402

rays = sources.circularbeam(10., 1000, rng="philox", seed=1)


def chain():
    with pxf.fused():
        tran.transform(rays, 0, 0, 1e-9, 0, 0, 0)
        surf.flat(rays)
        tran.reflect(rays)
        surf.flat(rays)
        tran.reflect(rays)
        surf.flat(rays)


chain()
torch.cuda.synchronize()
N = 3000
t0 = time.perf_counter()
for _ in range(N):
    chain()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("fused six-call chain: %.1f us per pass (host)" % ((t1 - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    chain()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
