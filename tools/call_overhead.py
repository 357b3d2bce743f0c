#!/usr/bin/env python3
"""Dev tool: where the host time of one API call goes (tiny bundle, so the GPU is never the limit)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyxfocus_amd.sources as sources  # noqa: E402
import pyxfocus_amd.surfaces as surf  # noqa: E402
import pyxfocus_amd.transformations as tran  # noqa: E402

rays = sources.circularbeam(10., 1000, rng="philox", seed=1)
surf.flat(rays)
torch.cuda.synchronize()
N = 5000
for name, fn in (("reflect", lambda: tran.reflect(rays)), ("flat", lambda: surf.flat(rays)),
                 ("transform", lambda: tran.transform(rays, 0, 0, 1e-9, 0, 0, 0))):
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-10s %.2f us per call (host issue rate)" % (name, (t1 - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tran.transform(rays, 0, 0, 1e-9, 0, 0, 0)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
