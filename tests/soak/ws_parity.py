#!/usr/bin/env python3
"""Position parity of the tolerance-parity mirrors (Wolter-Schwarzschild, Legendre-Legendre, woltersine)
against the oracle, stated the way the north-star states it: RMS ray-position error in mm, plus the
share of rays that sit more than 1e-10 / 1e-9 mm apart (rays whose linear-convergence iteration
stopped one step apart in the two implementations).  Prints one JSON object."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import api, cases  # noqa: E402
from util import call_gpu, to_dev, to_host  # noqa: E402


def main():
    num = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
    be = api.load_ref() if api.ref_available() else api.load_port()
    out = {}
    for name, fn, cols, args in cases.cases(num=num, seed=4321):
        if not any(t in fn for t in ("ws", "ll", "woltersine")):
            continue
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(be, fn, want, args)
        dev = to_dev(cols)
        call_gpu(fn, dev, args)
        got = to_host(dev)
        moved = (want["x"] != cols["x"]) | (want["y"] != cols["y"]) | (want["z"] != cols["z"])
        f = np.isfinite(want["x"]) & np.isfinite(want["y"]) & np.isfinite(want["z"]) & moved
        d = np.sqrt(sum((got[k][f] - want[k][f]) ** 2 for k in "xyz"))
        du = np.sqrt(sum((got[k][f] - want[k][f]) ** 2 for k in ("ux", "uy", "uz")))
        out[name] = {"fn": fn, "rays": int(f.sum()), "rms_pos_err_mm": float(np.sqrt((d ** 2).mean())),
                     "max_pos_err_mm": float(d.max()), "median_pos_err_mm": float(np.median(d)),
                     "frac_gt_1e-10": float((d > 1e-10).mean()), "frac_gt_1e-9": float((d > 1e-9).mean()),
                     "max_normal_err": float(du.max()),
                     "same_missed_rays": bool(np.array_equal(moved, (got["x"] != cols["x"]) | (got["y"] != cols["y"]) |
                                                             (got["z"] != cols["z"])))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
