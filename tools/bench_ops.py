#!/usr/bin/env python3
"""Per-kernel measurement of every surface / transform / analysis entry (the SURVEY §8f rows as
well as a-e): N rays resident in HBM, each op launched REPS times on a freshly restored bundle,
HIP events around each launch.  Prints one JSON object; algorithmic bytes per ray are the Op's
RD/WR column sets x 8 B (pyxfocus_amd/csrc/pxf_ops_*.h; columns a kernel writes only for rays that
miss -- the zeroed direction cosines of conic/sphere/cyl/spoCone -- are not counted).

usage: python tools/bench_ops.py [--rays 1e8] [--reps 5] > profiles/rNN_ops.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyxfocus_amd.analyses as anal  # noqa: E402
import pyxfocus_amd.sources as sources  # noqa: E402
import pyxfocus_amd.surfaces as surf  # noqa: E402
import pyxfocus_amd.transformations as tran  # noqa: E402
from pyxfocus_amd import conicsolve as con, native  # noqa: E402

R0, Z0, PSI, L = 220., 8400., 1., 100.
HBM = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=float, default=1e8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="", help="comma list of op-name prefixes (default: every op)")
    a = ap.parse_args()
    N = int(a.rays)
    rin, rout = float(con.primrad(Z0, R0, Z0)), float(con.primrad(Z0 + L, R0, Z0))

    def wolter_in():                       # annulus above a Wolter-I primary, travelling -z
        r = sources.annulus(rin, rout, N, rng="philox", seed=1)
        tran.transform(r, 0, 0, -(Z0 + 2 * L), 0, 0, 0)
        return r

    def secondary_in():                    # after primary + reflect
        r = wolter_in()
        surf.wolterprimary(r, R0, Z0)
        tran.reflect(r)
        return r

    def ws_secondary_in():                 # after a W-S primary + reflect
        r = wolter_in()
        surf.wsPrimary(r, R0, Z0, PSI)
        tran.reflect(r)
        return r

    def beam(rad=20., z=50., tilt=0.01):   # circular beam at z, travelling -z with a small tilt
        r = sources.circularbeam(rad, N, rng="philox", seed=2)
        r[3].fill_(z)
        r[4].fill_(tilt)
        r[6].copy_(-torch.sqrt(1 - r[4] ** 2 - r[5] ** 2))
        return r

    def ybeam():                           # beam travelling -y toward a y = f(x) cylinder (cylconic)
        r = sources.circularbeam(20., N, rng="philox", seed=3)
        r[2].add_(50.)
        r[4].fill_(0.01)
        r[5].fill_(-float(np.sqrt(1 - 1e-4)))
        r[6].fill_(0.)
        return r

    def normal_beam():                     # beam on a flat with normals set (for reflect/refract/angles)
        r = beam()
        surf.flat(r)
        return r

    rng = np.random.default_rng(0)
    zc = rng.normal(0, 1e-3, 36)
    zc[0] = 0.
    nn, mm = [], []
    n_ = 0
    while len(nn) < 36:                    # standard ordering up to n=7
        for m_ in range(-n_, n_ + 1, 2):
            nn.append(n_)
            mm.append(m_)
        n_ += 1
    nn, mm = np.array(nn[:36]), np.array(mm[:36])

    def zern_full(order):                  # every (n, m) up to radial order `order`: 66 terms at 10, 136 at 15
        n2, m2 = [], []
        for n__ in range(order + 1):
            for m__ in range(-n__, n__ + 1, 2):
                n2.append(n__)
                m2.append(m__)
        c2 = np.random.default_rng(order).normal(0, 1e-3, len(n2)) / (1. + np.array(n2))
        c2[0] = 0.
        return c2, np.array(n2), np.array(m2)
    z10, z15 = zern_full(10), zern_full(15)
    llc = np.array([1e-4, 5e-5, 2e-5])
    lla = np.array([2, 3, 4])
    llz = np.array([0, 1, 2])
    a11, z11 = np.meshgrid(np.arange(11), np.arange(11), indexing="ij")
    ll11c, ll11a, ll11z = rng.normal(0, 1e-5, 121) / (1 + a11.flatten() + z11.flatten()), a11.flatten(), z11.flatten()
    a8, z8 = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    ll8c, ll8a, ll8z = rng.normal(0, 1e-5, 64) / (1 + a8.flatten() + z8.flatten()), a8.flatten(), z8.flatten()
    a17, z17 = np.meshgrid(np.arange(6), np.arange(17), indexing="ij")
    ll17c, ll17a, ll17z = rng.normal(0, 1e-5, 102) / (1 + a17.flatten() + z17.flatten()), a17.flatten(), z17.flatten()

    import contextlib
    import pyxfocus_amd as pxf

    def ll_pair(r, fuse):
        with (pxf.fused() if fuse else contextlib.nullcontext()):
            surf.primaryLL(r, R0, Z0, Z0 + L, Z0, 2 * np.pi, ll11c, ll11a, ll11z)
            tran.reflect(r)
            tran.transform(r, 0, 0, 1e-3, 0, 1e-6, 0)
            surf.secondaryLL(r, R0, Z0, PSI, Z0, Z0 - L, 2 * np.pi, ll11c, ll11a, ll11z)
            tran.reflect(r)
            tran.itransform(r, 0, 0, 1e-3, 0, 1e-6, 0)
            surf.flat(r)

    def ws_pair(r, fuse):
        with (pxf.fused() if fuse else contextlib.nullcontext()):
            surf.wsPrimary(r, R0, Z0, PSI)
            tran.reflect(r)
            surf.wsSecondary(r, R0, Z0, PSI)
            tran.reflect(r)
            tran.transform(r, 0, 0, -1e-3, 0, 0, 0)
            surf.flat(r)

    ops = [
        # name, input builder, call, algorithmic B/ray
        # (a general frame change of a bundle whose normals are set: all nine columns change and are rewritten; a
        #  pure translation rewrites one -- TransformOp stores only what changed)
        ("transform", normal_beam, lambda r: tran.transform(r, .1, -.2, 3., 1e-3, -2e-3, .5), 144),
        ("itransform", normal_beam, lambda r: tran.itransform(r, .1, -.2, 3., 1e-3, -2e-3, .5), 144),
        ("transform (pure translation along z)", normal_beam, lambda r: tran.transform(r, 0, 0, -100., 0, 0, 0), 80),
        ("flat", beam, lambda r: surf.flat(r), 96),
        ("flatopd", beam, lambda r: surf.flat(r, nr=1.), 112),
        ("reflect", normal_beam, lambda r: tran.reflect(r), 72),
        ("refract", normal_beam, lambda r: tran.refract(r, 1., 1.5), 96),
        ("wolterprimary", wolter_in, lambda r: surf.wolterprimary(r, R0, Z0), 96),
        ("wolterprimaryopd", wolter_in, lambda r: surf.wolterprimary(r, R0, Z0, nr=1.), 112),
        ("woltersecondary", secondary_in, lambda r: surf.woltersecondary(r, R0, Z0), 96),
        ("woltersine", wolter_in, lambda r: surf.woltersine(r, R0, Z0, 1e-4, .1), 96),
        ("primaryLL", wolter_in, lambda r: surf.primaryLL(r, R0, Z0, Z0 + L, Z0, 2 * np.pi, llc, lla, llz), 96),
        ("primaryLL 11x11 coeffs (orders <= 10)", wolter_in,
         lambda r: surf.primaryLL(r, R0, Z0, Z0 + L, Z0, 2 * np.pi, ll11c, ll11a, ll11z), 96),
        ("primaryLL 8x8 coeffs (orders <= 7)", wolter_in,
         lambda r: surf.primaryLL(r, R0, Z0, Z0 + L, Z0, 2 * np.pi, ll8c, ll8a, ll8z), 96),
        ("primaryLL 6x17 coeffs (az order 16: generic kernel)", wolter_in,
         lambda r: surf.primaryLL(r, R0, Z0, Z0 + L, Z0, 2 * np.pi, ll17c, ll17a, ll17z), 96),
        ("secondaryLL", secondary_in,
         lambda r: surf.secondaryLL(r, R0, Z0, PSI, Z0, Z0 - L, 2 * np.pi, llc, lla, llz), 96),
        # a mirror PAIR with the frame changes and reflections around it: one launch per call vs ONE launch
        # (pyxfocus_amd.fused(): pxf_program_mirror.hip); algorithmic bytes of the one-launch form
        ("LL pair 11x11 per call (7 launches)", wolter_in, lambda r: ll_pair(r, False), 120),
        ("LL pair 11x11 fused (1 launch)", wolter_in, lambda r: ll_pair(r, True), 120),
        ("W-S pair per call (6 launches)", wolter_in, lambda r: ws_pair(r, False), 120),
        ("W-S pair fused (1 launch)", wolter_in, lambda r: ws_pair(r, True), 120),
        ("wsPrimary", wolter_in, lambda r: surf.wsPrimary(r, R0, Z0, PSI), 96),
        ("wsPrimaryB", wolter_in, lambda r: surf.wsPrimaryB(r, R0, Z0, PSI, .4), 96),
        ("wsSecondary", ws_secondary_in, lambda r: surf.wsSecondary(r, R0, Z0, PSI), 96),
        ("wsSecondaryB", ws_secondary_in, lambda r: surf.wsSecondaryB(r, R0, Z0, PSI, .4), 96),
        ("spoCone", wolter_in, lambda r: surf.spoPrimary(r, R0, Z0), 96),
        ("conic", lambda: beam(z=50.), lambda r: surf.conic(r, 2000., -1.), 96),
        ("conicopd", lambda: beam(z=50.), lambda r: surf.conic(r, 2000., -1., nr=1.), 112),
        ("sphere", lambda: beam(z=50.), lambda r: surf.sphere(r, 2000.), 96),
        ("cyl", lambda: beam(z=50.), lambda r: surf.cyl(r, 2000.), 96),
        ("cylconic", ybeam, lambda r: surf.cylconic(r, 1. / 2000., -1.), 96),
        ("conicplus", lambda: beam(z=50.), lambda r: surf.conicplus(r, 2000., -1., [1e-9, 1e-13]), 96),
        ("torus", lambda: beam(rad=5., z=50.), lambda r: surf.torus(r, 300., 2000.), 96),
        ("paraxial", normal_beam, lambda r: surf.paraxial(r, 1000.), 48),
        ("legSurf", normal_beam,
         lambda r: surf.legSurf(r, 25., 25., 1, np.array([1e-4, 2e-5, 1e-5]), np.array([1, 2, 0]),
                                np.array([0, 1, 2])), 64),
        ("zernsurf36", lambda: beam(rad=50., z=10.), lambda r: surf.zernsurf(r, zc, 50., rorder=nn, aorder=mm), 96),
        ("zernsurfN10 (66 terms, radial order 10)", lambda: beam(rad=50., z=10.),
         lambda r: surf.zernsurf(r, z10[0], 50., rorder=z10[1], aorder=z10[2]), 96),
        ("zernsurfN15 (136 terms, radial order 15)", lambda: beam(rad=50., z=10.),
         lambda r: surf.zernsurf(r, z15[0], 50., rorder=z15[1], aorder=z15[2]), 96),
        ("zernconic36", lambda: beam(rad=50., z=10.),
         lambda r: surf.zernconic(r, 5000., -1., zc, 50., rorder=nn, aorder=mm), 96),
        ("zernphase36", normal_beam, lambda r: surf.zernphase(r, zc, 50., 1e-3, rorder=nn, aorder=mm), 80),
        ("radgrat", normal_beam, lambda r: tran.radgrat(r, 160. / 8000., 1, 2.4), 64),
        ("grat", normal_beam, lambda r: tran.grat(r, 160., 1, 2.4e-6), 48),
        ("pointTo", normal_beam, lambda r: tran.pointTo(r, 0., 0., -8000.), 48),
        ("applyT", wolter_in, lambda r: tran.applyT(r, _coords(), inverse=False), 144),
        ("vignette(all kept)", normal_beam, lambda r: tran.vignette(r), 168),
        ("centroid+rmsCentroid", normal_beam, lambda r: anal.rmsCentroid(r), 32),
        ("hpd (radix select)", normal_beam, lambda r: anal.hpd(r), None),
        ("hpd weighted (weighted radix select)", normal_beam, lambda r: anal.hpd(r, weights=r[9]), None),
        ("indAngle", normal_beam, lambda r: anal.indAngle(r), 56),
        ("southwellbin 48x48 (1 LDS tile)", normal_beam,
         lambda r: native.southwellbin(r[1], r[2], r[4], r[5], 1.0, 48, 48), 32),
        ("southwellbin 128x128 (3 LDS tiles)", normal_beam,
         lambda r: native.southwellbin(r[1], r[2], r[4], r[5], 0.4, 128, 128), 32),
    ]

    def _coords():
        c = tran.newCoords()
        tran.transform([torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(10)], .1, -.2, 3.,
                       1e-3, -2e-3, .5, coords=c)
        return c

    out = {"rays": N, "reps": a.reps, "hbm_peak_GBps": HBM, "device": torch.cuda.get_device_name(0), "ops": []}
    only = [t for t in a.only.split(",") if t]
    for name, build, call, bpr in ops:
        if only and not any(name == t or name.startswith(t + " ") for t in only):
            continue
        try:
            base = build()
            torch.cuda.synchronize()
            work = [c.clone() for c in base]
            ms = []
            for rep in range(a.reps + 1):
                for w, b in zip(work, base):
                    w.copy_(b)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                res = call(work)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    ms.append(e0.elapsed_time(e1))
                del res
            avg = float(np.mean(ms))
            ent = {"op": name, "avg_ms": round(avg, 4), "min_ms": round(float(np.min(ms)), 4),
                   "rays_per_s": N / (avg * 1e-3), "bytes_per_ray": bpr}
            if bpr:
                ent["GBps"] = bpr * N / (avg * 1e-3) / 1e9
                ent["frac_of_hbm_peak"] = ent["GBps"] / HBM
            out["ops"].append(ent)
            del base, work
            torch.cuda.empty_cache()
        except Exception as e:  # keep going: one bad geometry must not lose the table
            out["ops"].append({"op": name, "error": repr(e)[:200]})
        print(json.dumps(out["ops"][-1]), file=sys.stderr, flush=True)
    # reconstruct: one workgroup, latency-bound -- report time per sweep for two grid sizes
    for dim in (() if only else (64, 128, 200)):
        try:
            g = torch.Generator(device="cpu").manual_seed(dim)
            yy, xx = torch.meshgrid(torch.linspace(-1, 1, dim, dtype=torch.float64),
                                    torch.linspace(-1, 1, dim, dtype=torch.float64), indexing="xy")
            sx, sy = 1e-4 * torch.sin(3 * xx), 2e-4 * torch.cos(2 * yy)
            inval = (xx ** 2 + yy ** 2) > 0.9
            ph = torch.zeros_like(xx)
            for t in (sx, sy, ph):
                t[inval] = 100.
            a, b, p = native.fgrid(sy), native.fgrid(sx), native.fgrid(ph)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            native.reconstruct(a.clone().t().contiguous().t(), b, 1e-30, 2. / dim, p.clone().t().contiguous().t(), 3)
            e0.record()
            _, sweeps = native.reconstruct(a, b, 1e-12, 2. / dim, p, 100000, return_sweeps=True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            ent = {"op": "reconstruct %dx%d" % (dim, dim), "ms": round(ms, 3), "sweeps": sweeps,
                   "us_per_sweep": round(1e3 * ms / sweeps, 2)}
        except Exception as e:
            ent = {"op": "reconstruct %dx%d" % (dim, dim), "error": repr(e)[:200]}
        out["ops"].append(ent)
        print(json.dumps(ent), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
