#!/usr/bin/env python3
"""Dev tool: zernsurf time per 1e8 rays as a function of the highest radial order (one instantiation each)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyxfocus_amd.sources as sources  # noqa: E402
import pyxfocus_amd.surfaces as surf  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
base = sources.circularbeam(50., N, rng="philox", seed=1)
base[3].fill_(10.)
base[4].fill_(0.01)
base[6].fill_(-float(np.sqrt(1 - 1e-4)))
work = [c.clone() for c in base]
for nmax in range(1, 16):
    nterms = (nmax + 1) * (nmax + 2) // 2
    ro, ao = surf.zmodes(nterms)
    coeff = np.random.default_rng(nmax).normal(0, 1e-3, nterms) / (1 + ro)
    coeff[0] = 0.
    ms = []
    for rep in range(4):
        for w, b in zip(work, base):
            w.copy_(b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        surf.zernsurf(work, coeff, 50., rorder=ro, aorder=ao)
        e1.record()
        torch.cuda.synchronize()
        if rep:
            ms.append(e0.elapsed_time(e1))
    print("nmax %2d  %3d terms  %.2f ms  %.2e rays/s" % (nmax, nterms, np.mean(ms), N / (np.mean(ms) * 1e-3)), flush=True)
