#!/usr/bin/env python3
"""A/B of program-kernel builds on ONE box: times the fused programs of configs 2 / 3 / 5 (exact and
tolerance mode) at 1e8 rays with the library named by $PXF_LIB (default: the in-tree one).  Prints one line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import pyxfocus_amd as pxf  # noqa: E402
import pyxfocus_amd.sources as sources  # noqa: E402
import pyxfocus_amd.surfaces as surf  # noqa: E402
import pyxfocus_amd.transformations as tran  # noqa: E402
from pyxfocus_amd import _lib, conicsolve as con  # noqa: E402
from tools import workloads as wl  # noqa: E402


def timeit(fn, reps=5):
    ms = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rays = fn(None)
        a.record()
        fn(rays)
        b.record()
        torch.cuda.synchronize()
        if r:
            ms.append(a.elapsed_time(b))
        del rays
    return round(float(np.median(ms)), 3)


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
    rin, rout = float(con.primrad(wl.Z0, wl.R0, wl.Z0)), float(con.primrad(wl.Z0 + wl.LMIR, wl.R0, wl.Z0))
    out = {"lib": os.path.relpath(_lib.LIB_PATH, ROOT)}
    for name, chain, src in (("c2", wl.config2_chain, lambda: sources.annulus(rin, rout, n, rng="philox", seed=1)),
                             ("c3", wl.config3_chain, lambda: sources.subannulus(rin, rout, .5, n, zhat=-1., rng="philox", seed=2))):
        for fast in (False, True):
            def fn(rays, fast=fast, chain=chain, src=src):
                if rays is None:
                    return src()
                with pxf.fused(fast=fast):
                    chain(rays, tran, surf)
            out["%s_%s" % (name, "fast" if fast else "exact")] = timeit(fn)
    import arcus_sector as A
    shells = A.petal_layout()
    per = max(172, (n // len(shells)) // 2 * 2)
    for fast in (False, True):
        res = A.defineSPOaperture_segmented(per, source="philox", seed=3, fast=fast)
        prog, rays = res[5], res[0]

        def fn(r, prog=prog, rays=rays):
            if r is None:
                sources.segments("subannulus", [(sh[0], sh[0] + A.PORE, sh[3], -1.) for sh in shells], per, out=rays, seed=4)
                return rays
            prog.run()
        out["spo_seg_%s" % ("fast" if fast else "exact")] = timeit(fn)
        del res, prog, rays
    print(json.dumps(out))


if __name__ == "__main__":
    main()
