#!/usr/bin/env python3
"""Measure BASELINE configs 3 and 4 on one MI355X (not the headline bench line; see bench.py).

config3: two-bounce Wolter-I (sub-aperture) + one radial grating, SURVEY 8d:
         config-2 chain ... reflect, then transform -> flat -> reflect -> radgrat -> itransform -> flat
         (1080 algorithmic B/ray with one launch per call: tools/workloads.py)
config4: 36-term Zernike sag surface (rad 50): zernsurf -> reflect -> flat (264 B/ray); fp64-VALU-bound.

Per-launch HIP-event timing; eager (one launch per call) and fused (one launch) variants.
Prints one JSON object per config.
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pyxfocus_amd as pxf  # noqa: E402
import pyxfocus_amd.analyses as anal  # noqa: E402
import pyxfocus_amd.sources as sources  # noqa: E402
import pyxfocus_amd.surfaces as surf  # noqa: E402
import pyxfocus_amd.transformations as tran  # noqa: E402
from pyxfocus_amd import conicsolve as con  # noqa: E402

R0, Z0, PSI, L = 220., 8400., 1., 100.
GRAZE, ZG = 1.5 * math.pi / 180, 8000.


class Timer:
    def __init__(self):
        self.rows = []

    def run(self, name, nbytes, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.rows.append((name, nbytes, a, b))

    def table(self, n):
        out = []
        for name, nbytes, a, b in self.rows:
            ms = a.elapsed_time(b)
            out.append({"kernel": name, "ms": ms, "bytes_per_ray": nbytes,
                        "GBps": nbytes * n / (ms * 1e-3) / 1e9 if nbytes else None})
        return out


def config3_steps(rays, t=None):
    run = t.run if t else (lambda name, nb, fn: fn())
    run("transform", 144, lambda: tran.transform(rays, 0, 0, -(Z0 + 2 * L), 0, 0, 0))
    run("wolterprimary", 96, lambda: surf.wolterprimary(rays, R0, Z0, PSI))
    run("reflect", 72, lambda: tran.reflect(rays))
    run("woltersecondary", 96, lambda: surf.woltersecondary(rays, R0, Z0, PSI))
    run("reflect", 72, lambda: tran.reflect(rays))
    run("flat", 96, lambda: surf.flat(rays))
    # grating stage (frame: origin = focus = grating hub)
    ty, rx = -ZG * math.sin(GRAZE), math.pi / 2 - GRAZE
    run("transform(g)", 144, lambda: tran.transform(rays, 0, ty, 0, rx, 0, 0))
    run("flat(g)", 96, lambda: surf.flat(rays))
    run("reflect(g)", 72, lambda: tran.reflect(rays))
    run("radgrat", 64, lambda: tran.radgrat(rays, 160. / ZG, 1., 2.4))
    run("itransform(g)", 144, lambda: tran.itransform(rays, 0, ty, 0, rx, 0, 0))
    run("flat(focal)", 96, lambda: surf.flat(rays))


def config3(n, reps):
    rin, rout = float(con.primrad(Z0, R0, Z0)), float(con.primrad(Z0 + L, R0, Z0))
    res = {"config": 3, "rays": n, "algorithmic_bytes_per_ray": 1080}
    best = None
    for r in range(reps + 1):
        rays = sources.subannulus(rin, rout, 0.5, n, zhat=-1., rng="philox", seed=30 + r)
        t = Timer()
        config3_steps(rays, t)
        torch.cuda.synchronize()
        tab = t.table(n)
        tot = sum(k["ms"] for k in tab)
        if r > 0 and (best is None or tot < best[0]):
            best = (tot, tab)
        if r == reps:
            res["valid_fraction"] = float(torch.isfinite(rays[6]).double().mean())
            res["focal_rms_x_mm"], res["focal_rms_y_mm"] = anal.rmsX(rays), anal.rmsY(rays)
            chk = [c.clone() for c in rays]
        del rays
    res["eager"] = {"ms": best[0], "rays_per_s": n / (best[0] * 1e-3), "GBps": 1080 * n / (best[0] * 1e-3) / 1e9,
                    "kernels": best[1]}
    fbest = None
    for r in range(reps + 1):
        rays = sources.subannulus(rin, rout, 0.5, n, zhat=-1., rng="philox", seed=30 + reps if r == reps else 90 + r)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        with pxf.fused():
            config3_steps(rays)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        if r > 0 and (fbest is None or ms < fbest):
            fbest = ms
        if r == reps:
            res["fused_bit_identical_to_eager"] = bool(all(
                torch.equal(torch.nan_to_num(u), torch.nan_to_num(v)) and torch.equal(u.isnan(), v.isnan())
                for u, v in zip(rays, chk)))
        del rays
    res["fused"] = {"ms": fbest, "rays_per_s": n / (fbest * 1e-3)}
    return res


def config4(n, reps, nterms=36):
    rng = np.random.default_rng(4)
    coeff = rng.normal(0., 1e-3, nterms)
    coeff[0] = 0.
    ro, ao = surf.zmodes(nterms)
    res = {"config": 4, "rays": n, "terms": nterms, "nmax": int(ro.max()), "algorithmic_bytes_per_ray": 264}
    best = None
    for r in range(reps + 1):
        rays = sources.circularbeam(50., n, rng="philox", seed=40 + r)
        rays[3].fill_(10.)
        rays[4].fill_(0.01)
        rays[6].fill_(-math.sqrt(1 - 1e-4))
        t = Timer()
        t.run("zernsurf", 96, lambda: surf.zernsurf(rays, coeff, 50., rorder=ro, aorder=ao))
        t.run("reflect", 72, lambda: tran.reflect(rays))
        t.run("flat", 96, lambda: surf.flat(rays))
        torch.cuda.synchronize()
        tab = t.table(n)
        tot = sum(k["ms"] for k in tab)
        if r > 0 and (best is None or tot < best[0]):
            best = (tot, tab)
        if r == reps:
            res["valid_fraction"] = float(torch.isfinite(rays[1]).double().mean())
        del rays
    z = best[1][0]
    res["eager"] = {"ms": best[0], "rays_per_s": n / (best[0] * 1e-3), "kernels": best[1]}
    res["zernsurf_rays_per_s"] = n / (z["ms"] * 1e-3)
    res["zernsurf_hbm_frac_of_8TBps"] = z["GBps"] / 8000.
    # the same three calls as ONE launch (program kernel around the Zernike trace; 120 B/ray)
    fbest = None
    for r in range(reps + 1):
        rays = sources.circularbeam(50., n, rng="philox", seed=40 + r)
        rays[3].fill_(10.)
        rays[4].fill_(0.01)
        rays[6].fill_(-math.sqrt(1 - 1e-4))
        chk = None
        if r == reps:
            chk = [c.clone() for c in rays]
            surf.zernsurf(chk, coeff, 50., rorder=ro, aorder=ao)
            tran.reflect(chk)
            surf.flat(chk)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        with pxf.fused():
            surf.zernsurf(rays, coeff, 50., rorder=ro, aorder=ao)
            tran.reflect(rays)
            surf.flat(rays)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        if r > 0 and (fbest is None or ms < fbest):
            fbest = ms
        if chk is not None:
            res["fused_max_abs_diff_vs_eager"] = max(float(torch.nan_to_num(u - v).abs().max()) for u, v in zip(rays, chk))
        del rays
    res["fused"] = {"ms": fbest, "rays_per_s": n / (fbest * 1e-3), "launches": 1 if int(ro.max()) <= 7 else 3}
    return res


# ---- config 5: Arcus-style multi-shell Wolter-I + grating -----------------------
ARCUS_ANG = [[-27.653, -16.592, -5.531, 5.531, 16.592, 27.653], [-23.174, -13.904, -4.635, 4.635, 13.904, 23.174],
             [-19.942, -11.965, -3.988, 3.988, 11.965, 19.942], [-16.065, -5.355, 5.355, 16.065],
             [-14.317, -4.772, 4.772, 14.317], [-12.972, -4.324, 4.324, 12.972], [-11.756, -3.919, 3.919, 11.756],
             [-10.791, -3.597, 3.597, 10.791]]                               # module azimuths, sector.py:402-410
ARCUS_RIN = [320.443, 382.638, 444.833, 507.027, 569.222, 631.417, 693.612, 755.807]   # sector.py:413
ARCUS_ROUT = [376.243, 438.438, 500.633, 562.827, 625.022, 681.217, 749.412, 811.607]  # sector.py:414
ARCUS_SPAN = [50.159, 49.839, 49.614, 89.363, 82.476, 77.572, 86.892, 82.053]          # sector.py:417


def arcus_shells(F=12e3, pitch=.775):
    """(r0, z0, module angle [rad], azimuthal width [rad]) for every shell of every module."""
    out = []
    for i in range(8):
        for a in ARCUS_ANG[i]:
            for r0 in np.arange(ARCUS_RIN[i], ARCUS_ROUT[i], pitch):
                out.append((float(r0), math.sqrt(F * F - r0 * r0), a * math.pi / 180, ARCUS_SPAN[i] / r0))
    return out


def config5(n, reps, lmir=50.):
    """One GPU's share of config 5: `n` rays spread evenly over all 2.7k shells of the Arcus layout,
    each shell with its own Wolter-I prescription (r0, z0=sqrt(F^2-r0^2)).  Per shell: device source
    into the shell's slice of ONE big bundle, then the two-bounce chain as a fused program on that
    slice (1 launch); then the grating stage + focal plane as one program over the whole bundle.
    The ~5.5k launches are captured once in a HIP graph and replayed (launch-bound otherwise)."""
    shells = arcus_shells()
    ns = len(shells)
    per = max(2, (n // ns) // 2 * 2)                # even -> every slice stays 16-byte aligned
    n = per * ns
    dev = torch.device("cuda", 0)
    buf = torch.empty((10, n), dtype=torch.float64, device=dev)
    rays = [buf[i] for i in range(10)]
    geo = [(r0, z0, a, w, float(con.primrad(z0, r0, z0)), float(con.primrad(z0 + lmir, r0, z0)))
           for (r0, z0, a, w) in shells]
    ty, rx = -ZG * math.sin(GRAZE), math.pi / 2 - GRAZE

    def step(seed):
        for k, (r0, z0, a, w, rin, rout) in enumerate(geo):
            sl = [c[k * per:(k + 1) * per] for c in rays]
            sources.subannulus(rin, rout, w, per, zhat=-1., rng="philox", seed=seed, first_index=k * per, out=sl)
            with pxf.fused():
                tran.transform(sl, 0, 0, -(z0 + 2 * lmir), 0, 0, a)      # to the shell's focus frame, module azimuth
                surf.wolterprimary(sl, r0, z0)
                tran.reflect(sl)
                surf.woltersecondary(sl, r0, z0)
                tran.reflect(sl)
        with pxf.fused():                                                  # grating stage on the whole bundle
            tran.transform(rays, 0, ty, 0, rx, 0, 0)
            surf.flat(rays)
            tran.reflect(rays)
            tran.radgrat(rays, 160. / ZG, 1., 2.4)
            tran.itransform(rays, 0, ty, 0, rx, 0, 0)
            surf.flat(rays)

    res = {"config": 5, "rays": n, "shells": ns, "rays_per_shell": per}
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step(50)                                                               # warm-up, eager
    torch.cuda.synchronize()
    import time
    best = None
    for r in range(reps):
        t0 = time.perf_counter()
        a.record(); step(51 + r); b.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        best = wall if best is None else min(best, wall)
    res["eager_loop"] = {"ms_wall": best, "rays_per_s": n / (best * 1e-3), "launches": 2 * ns + 1}
    res["valid_fraction"] = float(torch.isfinite(rays[6]).double().mean())
    res["focal_rms_x_mm"], res["focal_rms_y_mm"] = anal.rmsX(rays), anal.rmsY(rays)
    ref_x = rays[1].clone()
    # the same step as one captured HIP graph
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            step(51 + reps - 1)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gbest = None
    for r in range(reps + 1):
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        if r > 0:
            gbest = ms if gbest is None else min(gbest, ms)
    res["hip_graph"] = {"ms": gbest, "rays_per_s": n / (gbest * 1e-3), "nodes": 2 * ns + 1}
    res["graph_equals_eager"] = bool(torch.equal(torch.nan_to_num(rays[1]), torch.nan_to_num(ref_x)))
    del g
    # the same step as THREE launches: segmented source, segmented two-bounce program (per-shell
    # parameter blocks staged in LDS), grating program
    def shell(sl, r0, z0, a_):
        tran.transform(sl, 0, 0, -(z0 + 2 * lmir), 0, 0, a_)
        surf.wolterprimary(sl, r0, z0)
        tran.reflect(sl)
        surf.woltersecondary(sl, r0, z0)
        tran.reflect(sl)

    import time as _t
    t0 = _t.perf_counter()
    prog = pxf.compile_segments(rays, per, shell, [(g_[0], g_[1], g_[2]) for g_ in geo])
    res["segmented_compile_s"] = _t.perf_counter() - t0
    src_par = [(g_[4], g_[5], g_[3], -1.) for g_ in geo]

    def step3(seed):
        sources.segments("subannulus", src_par, per, out=rays, seed=seed)
        prog.run()
        with pxf.fused():
            tran.transform(rays, 0, ty, 0, rx, 0, 0)
            surf.flat(rays)
            tran.reflect(rays)
            tran.radgrat(rays, 160. / ZG, 1., 2.4)
            tran.itransform(rays, 0, ty, 0, rx, 0, 0)
            surf.flat(rays)

    sbest = None
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for r in range(reps + 1):
        evs[0].record()
        sources.segments("subannulus", src_par, per, out=rays, seed=51 + reps - 1)
        evs[1].record()
        prog.run()
        evs[2].record()
        with pxf.fused():
            tran.transform(rays, 0, ty, 0, rx, 0, 0)
            surf.flat(rays)
            tran.reflect(rays)
            tran.radgrat(rays, 160. / ZG, 1., 2.4)
            tran.itransform(rays, 0, ty, 0, rx, 0, 0)
            surf.flat(rays)
        evs[3].record()
        torch.cuda.synchronize()
        ms = evs[0].elapsed_time(evs[3])
        if r > 0 and (sbest is None or ms < sbest[0]):
            sbest = (ms, evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2]), evs[2].elapsed_time(evs[3]))
    res["segmented"] = {"ms": sbest[0], "rays_per_s": n / (sbest[0] * 1e-3), "launches": 3,
                        "source_ms": sbest[1], "two_bounce_ms": sbest[2], "grating_ms": sbest[3]}
    res["segmented_equals_eager"] = bool(torch.equal(torch.nan_to_num(rays[1]), torch.nan_to_num(ref_x)))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=float, default=1e8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--configs", default="3,4")
    args = ap.parse_args()
    n = int(args.rays)
    torch.cuda.set_device(0)
    for c in args.configs.split(","):
        res = {"3": config3, "4": config4, "5": config5}[c](n, args.reps)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
