#!/usr/bin/env python3
"""One bundle sized for the whole HBM of an MI355X (288 GB): N rays x 80 B resident, the config-2
two-bounce chain eager and fused, focal metrics.  usage: python tools/max_bundle.py [--rays 3e9]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyxfocus_amd as pxf  # noqa: E402
import pyxfocus_amd.analyses as anal  # noqa: E402
import pyxfocus_amd.sources as sources  # noqa: E402
import pyxfocus_amd.surfaces as surf  # noqa: E402
import pyxfocus_amd.transformations as tran  # noqa: E402
from pyxfocus_amd import conicsolve as con  # noqa: E402

R0, Z0, L = 220., 8400., 100.


def chain(rays):
    tran.transform(rays, 0, 0, -(Z0 + 2 * L), 0, 0, 0)
    surf.wolterprimary(rays, R0, Z0)
    tran.reflect(rays)
    surf.woltersecondary(rays, R0, Z0)
    tran.reflect(rays)
    surf.flat(rays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=float, default=3e9)
    a = ap.parse_args()
    N = int(a.rays)
    free, total = torch.cuda.mem_get_info()
    out = {"rays": N, "bundle_GB": 80. * N / 1e9, "hbm_total_GB": total / 1e9, "hbm_free_GB": free / 1e9}
    rin, rout = float(con.primrad(Z0, R0, Z0)), float(con.primrad(Z0 + L, R0, Z0))
    rays = sources.annulus(rin, rout, N, rng="philox", seed=5)
    torch.cuda.synchronize()
    for tag, ctx in (("eager", None), ("fused", pxf.fused)):
        if tag == "fused":
            sources.annulus(rin, rout, N, rng="philox", seed=5, out=rays)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if ctx:
            with ctx():
                chain(rays)
        else:
            chain(rays)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[tag] = {"ms": dt * 1e3, "rays_per_s": N / dt, "rms_centroid_mm": anal.rmsCentroid(rays),
                    "hpd_mm": anal.hpd(rays), "centroid": list(anal.centroid(rays))}
    out["eager_equals_fused_metrics"] = out["eager"]["rms_centroid_mm"] == out["fused"]["rms_centroid_mm"] and \
        out["eager"]["hpd_mm"] == out["fused"]["hpd_mm"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
