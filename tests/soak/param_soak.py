"""dev: random PRESCRIPTIONS (not just random rays) through the bit-exact and tolerance-parity mirror entries, GPU
vs the oracle.  The committed cases fix r0 = 220, z0 = 8400 ...; this draws r0, z0, psi, conic constants, cone angles,
W-S designs, substrate thicknesses and reports every mismatch.  usage: param_soak.py [iterations] [seed]"""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import api, cases
from util import ALLCOLS, bits_equal, call_gpu, to_dev, to_host
port = api.load_port()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NINE = ALLCOLS[1:]
n = 3000
bad = 0


def annulus(r0, z0, psi, L, off=0.):
    rin, rout = cases.primrad(z0, r0, z0, psi), cases.primrad(z0 + L, r0, z0, psi)
    c = cases.annulus_rays(rng, n, rin, rout, z0 + 2 * L)
    if off:
        c["l"][:] = math.sin(off); c["n"][:] = -math.sqrt(1 - math.sin(off) ** 2)
    return c


def check(name, fn, cols, args, bitwise=True, tol=1e-10):
    global bad
    want = {k: v.copy() for k, v in cols.items()}
    cases.call(port, fn, want, args)
    t = to_dev(cols)
    call_gpu(fn, t, args)
    got = to_host(t)
    if bitwise:
        nb = {k: int((~bits_equal(got[k], want[k])).sum()) for k in ALLCOLS}
        if any(nb.values()):
            bad += 1
            print("MISMATCH", name, [("%.17g" % a) if np.isscalar(a) else "arr" for a in args], {k: v for k, v in nb.items() if v})
    else:
        moved = np.isfinite(want["x"]) & np.isfinite(got["x"])
        d = np.sqrt(sum((got[k][moved] - want[k][moved]) ** 2 for k in "xyz"))
        rms = float(np.sqrt((d ** 2).mean())) if d.size else 0.
        same_nan = np.array_equal(np.isnan(want["x"]), np.isnan(got["x"]))
        if rms > tol or not same_nan:
            bad += 1
            print("TOL", name, [("%.17g" % a) if np.isscalar(a) else "arr" for a in args], "rms %.2e share>1e-10 %.4f max %.2e nan-same %s" % (
                rms, float((d > 1e-10).mean()), float(d.max()), same_nan))


for it in range(iters):
    r0 = float(rng.uniform(80., 1500.)); z0 = float(rng.uniform(3000., 20000.)); psi = float(rng.choice([1., 1., 2., .7, 1.3]))
    L = float(rng.uniform(50., 300.)); off = float(rng.choice([0., 1e-4, 1e-3]))
    c = annulus(r0, z0, psi, L, off)
    check("wolterprimary", "woltsurf.wolterprimary", c, (r0, z0, psi))
    check("wolterprimaryopd", "woltsurf.wolterprimaryopd", annulus(r0, z0, psi, L, off), (r0, z0, psi, float(rng.uniform(1., 2.))))
    c2 = annulus(r0, z0, psi, L, off)
    a = [c2[k] for k in NINE]
    port.woltsurf.wolterprimary(*a, r0, z0, psi); port.transformationsf.reflect(*a[3:])
    check("woltersecondary", "woltsurf.woltersecondary", c2, (r0, z0, psi))
    # SPO cone
    F = float(rng.uniform(5000., 20000.)); R0s = float(rng.uniform(200., 1200.)); tg = float(rng.choice([.25, .75]) * math.atan((R0s + .3) / F))
    cc = cases.annulus_rays(rng, n, R0s - .4, R0s + 1.2, float(rng.uniform(-30., 200.)))
    cc["l"] = rng.normal(0, 2e-4, n); cc["m"] = rng.normal(0, 2e-4, n); cc["n"] = -np.sqrt(1 - cc["l"] ** 2 - cc["m"] ** 2)
    check("spocone", "woltsurf.spocone", cc, (R0s, tg))
    # conics
    for fn_, ar in (("surfacesf.conic", (float(rng.choice([-1, 1]) * rng.uniform(200, 5000)), float(rng.choice([-1., 0., -2.5, .5, rng.uniform(-3, 1)])))),):
        cq = cases._randrays(rng, n, nsign=-1.); cq["l"] *= .1; cq["m"] *= .1; cq["n"] = -np.sqrt(1 - cq["l"] ** 2 - cq["m"] ** 2); cq["z"] = rng.uniform(50, 150, n)
        check("conic", fn_, cq, ar)
        check("conicopd", "surfacesf.conicopd", {k: v.copy() for k, v in cq.items()}, ar + (1.33,))
    for fn_, tag in (("tracesphere", "sphere"), ("tracecyl", "cyl")):
        cs = cases._randrays(rng, n); cs["x"] = rng.uniform(-40, 40, n); cs["y"] = rng.uniform(-40, 40, n); cs["z"] = rng.uniform(150, 250, n)
        cs["l"] *= .2; cs["m"] *= .2; cs["n"] = -np.sqrt(1 - cs["l"] ** 2 - cs["m"] ** 2)
        check(tag, "surfacesf." + fn_, cs, (float(rng.uniform(60., 400.)),))
    ct = cases._randrays(rng, n); ct["x"] = rng.uniform(-10, 10, n); ct["y"] = rng.uniform(-10, 10, n); ct["z"] = rng.uniform(5, 15, n)
    ct["l"] *= .1; ct["m"] *= .1; ct["n"] = -np.sqrt(1 - ct["l"] ** 2 - ct["m"] ** 2)
    check("torus", "surfacesf.torus", ct, (float(rng.uniform(100., 400.)), float(rng.uniform(800., 3000.))))
    # cylindrical conic, conic + even polynomial (OPD too), 2-D Legendre phase surface, paraxial lenses, OPD variants
    cy = cases._randrays(rng, n); cy["x"] = rng.uniform(-20, 20, n); cy["y"] = rng.uniform(5, 10, n)
    cy["l"] = rng.uniform(-.05, .05, n); cy["n"] = rng.uniform(-.05, .05, n); cy["m"] = -np.sqrt(1 - cy["l"] ** 2 - cy["n"] ** 2)
    check("cylconic", "surfacesf.cylconic", cy, (float(rng.choice([-1, 1]) / rng.uniform(300., 1500.)), float(rng.choice([-1., .5, 0., rng.uniform(-2, 1)]))))
    npc = int(rng.integers(1, 17))
    pp = rng.normal(0, 1., npc) * 10. ** (-6. - 4. * np.arange(npc))
    cp = cases._randrays(rng, n); cp["x"] = rng.uniform(-30, 30, n); cp["y"] = rng.uniform(-30, 30, n); cp["z"] = rng.uniform(20, 40, n)
    cp["l"] *= .1; cp["m"] *= .1; cp["n"] = -np.sqrt(1 - cp["l"] ** 2 - cp["m"] ** 2)
    Rp, Kp = float(rng.choice([-1, 1]) * rng.uniform(400., 2000.)), float(rng.uniform(-1.5, .5))
    check("conicplus", "surfacesf.conicplus", cp, (Rp, Kp, pp))
    check("conicplusopd", "surfacesf.conicplusopd", {k: v.copy() for k, v in cp.items()}, (Rp, Kp, pp, 1.5))
    nc = int(rng.integers(1, 40)); om = int(rng.integers(0, 13))
    cl = cases._randrays(rng, n); cl["x"] = rng.uniform(-12, 12, n); cl["y"] = rng.uniform(-8, 8, n)
    cl["l"] *= .05; cl["m"] *= .05; cl["n"] = np.sqrt(1 - cl["l"] ** 2 - cl["m"] ** 2) * rng.choice([-1., 1.], n)
    check("legsurf", "surfacesf.legsurf", cl, (10., 10., float(rng.choice([1e-3, 1., 2.4e-6])), rng.normal(0, .1, nc),
                                                   rng.integers(0, om + 1, nc).astype(np.int32), rng.integers(0, om + 1, nc).astype(np.int32)))
    check("paraxial", "surfacesf.paraxial", cases._randrays(rng, n), (float(rng.choice([-1, 1]) * rng.uniform(20., 2000.)),))
    check("paraxialy", "surfacesf.paraxialy", cases._randrays(rng, n), (float(rng.choice([-1, 1]) * rng.uniform(20., 2000.)),))
    check("flatopd", "surfacesf.flatopd", cases._randrays(rng, n), (float(rng.uniform(1., 2.)),))
    for fn_ in ("tracesphereopd", "tracecylopd"):
        cs = cases._randrays(rng, n); cs["x"] = rng.uniform(-40, 40, n); cs["y"] = rng.uniform(-40, 40, n); cs["z"] = rng.uniform(150, 250, n)
        cs["l"] *= .2; cs["m"] *= .2; cs["n"] = -np.sqrt(1 - cs["l"] ** 2 - cs["m"] ** 2)
        check(fn_, "surfacesf." + fn_, cs, (float(rng.uniform(60., 400.)), float(rng.uniform(1., 2.))))
    # W-S designs: alpha, z0, psi (+ thickness): RMS position error
    alpha = .25 * math.atan(r0 / z0)
    cw = annulus(r0, z0, 1., L, 1e-4)
    check("wsprimary", "woltsurf.wsprimary", cw, (alpha, z0, psi), bitwise=False)
    cw2 = annulus(r0, z0, 1., L)
    a = [cw2[k] for k in NINE]
    port.woltsurf.wsprimary(*a, alpha, z0, psi); port.transformationsf.reflect(*a[3:])
    check("wssecondary", "woltsurf.wssecondary", cw2, (alpha, z0, psi), bitwise=False)
    th = float(rng.uniform(.1, 1.))
    cb = annulus(r0, z0, 1., L, 5e-5)
    h = np.hypot(cb["x"], cb["y"]); cb["x"] += th * cb["x"] / h; cb["y"] += th * cb["y"] / h
    check("wsprimaryback", "woltsurf.wsprimaryback", cb, (alpha, z0, psi, th), bitwise=False)
    cb2 = {k: v.copy() for k, v in cb.items()}
    a = [cb2[k] for k in NINE]
    port.woltsurf.wsprimaryback(*a, alpha, z0, psi, th); port.transformationsf.reflect(*a[3:])
    check("wssecondaryback", "woltsurf.wssecondaryback", cb2, (alpha, z0, psi, th), bitwise=False)
    check("woltersine", "woltsurf.woltersine", annulus(r0, z0, 1., L, 1e-4), (r0, z0, float(rng.uniform(1e-5, 1e-3)), float(rng.uniform(.01, .2))), bitwise=False)
print("param soak: %d iterations, %d mismatches" % (iters, bad))
