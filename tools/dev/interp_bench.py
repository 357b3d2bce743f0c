#!/usr/bin/env python3
"""Timing of analyses.interpolateVec(on_device=True) against the host path (scipy griddata) on one focal plane.
usage: interp_bench.py [rays] [grid]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pyxfocus_amd.analyses as anal  # noqa: E402
import pyxfocus_amd.sources as sources  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    g = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rays = sources.circularbeam(50., n, rng="philox", seed=5)
    rays[4].copy_(torch.sin(rays[1] * .1) * 1e-3)
    out = {"rays": n, "grid": [g, g]}
    for meth in ("linear", "nearest"):
        anal.interpolateVec(rays, 4, g, g, method=meth, on_device=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = anal.interpolateVec(rays, 4, g, g, method=meth, on_device=True)[0]
        torch.cuda.synchronize()
        out["device_%s_ms" % meth] = (time.perf_counter() - t0) * 1e3
        if meth == "linear":
            out["inside_fraction"] = float((~torch.isnan(res)).double().mean())
        sys.stderr.write("%s done\n" % meth)
    if n <= 2_000_000:
        t0 = time.perf_counter()
        want = anal.interpolateVec(rays, 4, g, g, method="linear")[0]
        out["host_scipy_linear_ms"] = (time.perf_counter() - t0) * 1e3
        got = anal.interpolateVec(rays, 4, g, g, method="linear", on_device=True)[0].cpu().numpy()
        ok = ~np.isnan(want) & ~np.isnan(got)
        out["max_abs_diff_vs_scipy"] = float(np.abs(got[ok] - want[ok]).max())
        out["nan_pattern_differs_on"] = int((np.isnan(want) != np.isnan(got)).sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
