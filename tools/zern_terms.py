#!/usr/bin/env python3
"""Zernike miscompile hunt, step 2: which TERM does an instantiation get wrong?  For one nmax, run
tracezern once per term with only that coefficient non-zero (plus, last, the dense set) and save z and
the Newton path for every ray to an .npz; two runs with two libraries ($PXF_LIB) are then diffed
offline.  usage: zern_terms.py nmax out.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("PXF_PKG_ROOT", ROOT))
from pyxfocus_amd import native  # noqa: E402


def noll(nterms):
    ro, ao, n, j = [], [], 0, 1
    while len(ro) < nterms:
        for mm in [m for m in range(n + 1) if (n - m) % 2 == 0]:
            if mm == 0:
                ro.append(n); ao.append(0); j += 1
            else:
                for _ in range(2):
                    ro.append(n); ao.append(mm if j % 2 == 0 else -mm); j += 1
        n += 1
    return np.array(ro[:nterms], np.int32), np.array(ao[:nterms], np.int32)


def main():
    nmax, out = int(sys.argv[1]), sys.argv[2]
    n = 1024
    nterms = (nmax + 1) * (nmax + 2) // 2
    ro, ao = noll(nterms)
    rng = np.random.default_rng(40 + nmax)
    dense = rng.normal(0, 1e-3, nterms) / (1 + ro)
    dense[0] = 0.
    rho, th = 1. + 44. * np.sqrt(rng.random(n)), rng.random(n) * 2 * np.pi
    base = {"x": rho * np.cos(th), "y": rho * np.sin(th), "z": np.full(n, 10.), "l": np.full(n, .01),
            "m": rng.normal(0, 1e-3, n)}
    base["n"] = -np.sqrt(1 - base["l"] ** 2 - base["m"] ** 2)
    res = {"ro": ro, "ao": ao, "rho": rho}
    sets = [(i, np.eye(nterms)[i] * 1e-3) for i in range(1, nterms)] + [(-1, dense)]
    for i, coeff in sets:
        t = {k: torch.from_numpy(v.copy()).cuda() for k, v in base.items()}
        u = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
        native.zernsurf.tracezern(t["x"], t["y"], t["z"], t["l"], t["m"], t["n"], *u, coeff, ro, ao, 50.)
        res["z%d" % i] = t["z"].cpu().numpy()
        res["x%d" % i] = t["x"].cpu().numpy()
        res["ux%d" % i] = u[0].cpu().numpy()
    np.savez_compressed(out, **res)
    print("saved", out, len(sets), "sets")


if __name__ == "__main__":
    main()
