#!/usr/bin/env python3
"""How much of a wave's work in the fanned-grating search is real?  Every lane walks the fan until ITS ray meets a
grating; a wave runs until its slowest lane has.  Lane utilisation = sum(rounds) / (64 * sum over waves of max rounds),
measured on the config-5 bundle (examples/arcus_sector.py) from the it_hit column pxf_fan_search returns.
usage: fan_divergence.py [rays]"""
import ctypes as C
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import arcus_sector as A  # noqa: E402
from pyxfocus_amd import _lib, native  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    shells = A.petal_layout()
    per = max(172, (n // len(shells)) // 2 * 2)
    api = A.Api.amd()
    rays, w, minfoc, lmax, coords, _ = A.defineSPOaperture_segmented(per, source="philox", seed=300)
    focp = np.dot(coords[3], [0., 0., 0., 1.])[:3]
    hubdist = float(np.sqrt(A.OUTERRAD ** 2 + focp[2] ** 2))
    angle = float(np.arctan(A.OUTERRAD / (minfoc - (lmax + A.GAP + A.GRATLEN))))
    tran, pi = api.tran, math.pi
    with api.fused():
        tran.transform(rays, A.OUTERRAD, 0, 0, 0, 0, 0)
        tran.transform(rays, 0, 0, 0, 0, 0, -pi / 2)
        tran.transform(rays, 0, 0, 0, -pi / 2 - angle + A.INC, 0, 0)
        tran.transform(rays, 0, 0, 0, 0, 0, A.YAW)
        tran.transform(rays, 0, hubdist, 0, 0, 0, 0)
    ang = A.GRATLEN * math.sin(A.INC) / hubdist * .95
    cols = list(rays[1:10])
    num, dev = native._check_cols(cols, "fan")
    it_hit = torch.empty(num, dtype=torch.int32, device=dev)
    scal = torch.zeros(4, dtype=torch.int32, device=dev)
    ptrs = (C.c_void_p * 10)(None, *[t.data_ptr() for t in cols])
    h = _lib.lib()
    keep = [t.clone() for t in cols]
    best = [1e30, 1e30]
    for rep in range(3):                              # (the first repetition carries the code-object load)
        for t, k in zip(cols, keep):
            t.copy_(k)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        _lib.check(h.pxf_fan_search(ptrs, num, hubdist, A.GRATLEN, ang, .001, 100000, it_hit.data_ptr(), scal.data_ptr(),
                                    scal[2:].data_ptr(), native._stream(dev)), "pxf_fan_search")
        ev[1].record()
        torch.cuda.synchronize()
        imax_ = int(scal[0])
        _lib.check(h.pxf_fan_finish(ptrs, num, ang, 160. / hubdist, -3., 2.4, it_hit.data_ptr(), imax_,
                                    native._stream(dev)), "pxf_fan_finish")
        ev[2].record()
        torch.cuda.synchronize()
        best = [min(best[0], ev[0].elapsed_time(ev[1])), min(best[1], ev[1].elapsed_time(ev[2]))]
    del keep
    it = it_hit.cpu().numpy().astype(np.int64)
    imax = int(scal[0])
    full = it[:num // 64 * 64].reshape(-1, 64)
    search_rounds = full + 1                       # round 0 counts
    finish_rounds = imax - full
    out = {"rays": num, "imax": imax, "search_ms": best[0], "finish_ms": best[1],
           "mean_rounds_search": float(search_rounds.mean()), "mean_of_wave_max_search": float(search_rounds.max(1).mean()),
           "lane_utilisation_search": float(search_rounds.sum() / (64 * search_rounds.max(1).sum())),
           "lane_utilisation_finish": float(finish_rounds.sum() / max(1, 64 * finish_rounds.max(1).sum())),
           "hist_it_deciles": [int(v) for v in np.percentile(it, [0, 10, 25, 50, 75, 90, 100])]}
    # the same bundle with rays sorted by hit round: what a perfectly binned launch would reach
    srt = np.sort(it)[:num // 64 * 64].reshape(-1, 64) + 1
    out["lane_utilisation_if_sorted"] = float(srt.sum() / (64 * srt.max(1).sum()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
