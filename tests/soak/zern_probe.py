#!/usr/bin/env python3
"""Which rays does a Zernike instantiation get wrong, and how?  For nmax in the given range: tracezern
on the GPU (library = $PXF_LIB or the in-tree one) vs the oracle; per order: max |dz|, number of wrong
rays, their distribution over lane-in-wave (ray index mod 64) and over Newton iteration count (from the
oracle's path length), and the first few offenders.  usage: zern_probe.py [lo hi [n]]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import api, cases  # noqa: E402
from pyxfocus_amd import _lib, native  # noqa: E402
from util import ALLCOLS  # noqa: E402


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9, 15)
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    port = api.load_port()
    out = {"lib": _lib.LIB_PATH}
    for nmax in range(lo, hi + 1):
        rng = np.random.default_rng(40 + nmax)
        nterms = (nmax + 1) * (nmax + 2) // 2
        ro, ao = cases.noll_orders(nterms)
        ro, ao = np.asarray(ro, np.int32), np.asarray(ao, np.int32)
        coeff = rng.normal(0, 1e-3, nterms) / (1 + ro)
        coeff[0] = 0.
        c = {k: np.zeros(n) for k in ALLCOLS}
        rho, th = 1. + 44. * np.sqrt(rng.random(n)), rng.random(n) * 2 * np.pi
        c["x"], c["y"], c["z"] = rho * np.cos(th), rho * np.sin(th), np.full(n, 10.)
        c["l"], c["m"] = np.full(n, .01), rng.normal(0, 1e-3, n)
        c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
        nine = ALLCOLS[1:]
        want = {k: v.copy() for k, v in c.items()}
        port.zernsurf.tracezern(*[want[k] for k in nine], coeff, ro, ao, 50.)
        t = {k: torch.from_numpy(v.copy()).cuda() for k, v in c.items()}
        native.zernsurf.tracezern(*[t[k] for k in nine], coeff, ro, ao, 50.)
        got = {k: v.cpu().numpy() for k, v in t.items()}
        err = {k: np.abs(got[k] - want[k]) for k in nine}
        bad = np.zeros(n, bool)
        for k in nine:
            bad |= ~(err[k] <= 1e-11 * (1 + np.abs(want[k])))
        ent = {"max_err": {k: float(np.nanmax(err[k])) for k in nine}, "wrong_rays": int(bad.sum())}
        if bad.any():
            idx = np.nonzero(bad)[0]
            ent["lanes"] = np.bincount(idx % 64, minlength=64).tolist()
            ent["waves"] = sorted(set((idx // 64).tolist()))[:20]
            ent["first"] = [{"i": int(i), **{k: [float(got[k][i]), float(want[k][i])] for k in ("x", "z", "ux")}} for i in idx[:4]]
        out[str(nmax)] = ent
        print(nmax, json.dumps(ent)[:600], flush=True)


if __name__ == "__main__":
    main()
