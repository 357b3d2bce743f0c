"""TEST INFRASTRUCTURE ONLY -- deterministic input cases for every hot-path function.

Each case is ``(name, fn_path, cols, args)``:
  * ``fn_path``  "module.function" in the f2py-style namespaces of oracle/api.py,
  * ``cols``     dict of float64 input columns (the call mutates copies of them),
  * ``args``     tuple of trailing scalar / array arguments.
The same cases feed oracle/make_golden.py (reference -> tests/golden/*.npz), the
oracle-vs-reference test and the GPU parity tests, so all three see identical inputs.

Geometry follows SURVEY.md section 8d (config 2: r0=220, z0=8400, psi=1, L=100).
No reference code is used here; scalars come from our own restatement.
"""
import math

import numpy as np

R0, Z0, PSI, LMIR = 220., 8400., 1., 100.

# call signature: which columns each function takes, in order (f2py signatures)
SIG = {
    "surfacesf.flat": "x y z l m n ux uy uz",
    "surfacesf.flatopd": "x y z l m n ux uy uz opd",
    "surfacesf.conic": "x y z l m n ux uy uz",
    "surfacesf.conicopd": "opd x y z l m n ux uy uz",
    "woltsurf.wolterprimary": "x y z l m n ux uy uz",
    "woltsurf.wolterprimaryopd": "opd x y z l m n ux uy uz",
    "woltsurf.woltersecondary": "x y z l m n ux uy uz",
    "zernsurf.tracezern": "x y z l m n ux uy uz",
    "zernsurf.tracezernopd": "opd x y z l m n ux uy uz",
    "transformationsf.transform": "x y z l m n ux uy uz",
    "transformationsf.itransform": "x y z l m n ux uy uz",
    "transformationsf.reflect": "l m n ux uy uz",
    "transformationsf.radgrat": "x y l m n",
    "transformationsf.radgratw": "x y l m n",
    "woltsurf.spocone": "x y z l m n ux uy uz",
    "transformationsf.grat": "x y l m n",
    "surfacesf.tracesphere": "x y z l m n ux uy uz",
    "surfacesf.tracesphereopd": "opd x y z l m n ux uy uz",
    "surfacesf.tracecyl": "x y z l m n ux uy uz",
    "surfacesf.tracecylopd": "opd x y z l m n ux uy uz",
    "surfacesf.cylconic": "x y z l m n ux uy uz",
    "surfacesf.paraxial": "x y z l m n ux uy uz",
    "surfacesf.paraxialy": "x y z l m n ux uy uz",
    "transformationsf.refract": "l m n ux uy uz",
    "woltsurf.woltersine": "x y z l m n ux uy uz",
    "woltsurf.wolterprimll": "x y z l m n ux uy uz",
    "woltsurf.woltersecll": "x y z l m n ux uy uz",
    "woltsurf.ellipsoidwoltll": "x y z l m n ux uy uz",
    "woltsurf.wsprimary": "x y z l m n ux uy uz",
    "woltsurf.wssecondary": "x y z l m n ux uy uz",
    "woltsurf.wsprimaryback": "x y z l m n ux uy uz",
    "woltsurf.wssecondaryback": "x y z l m n ux uy uz",
    "surfacesf.torus": "x y z l m n ux uy uz",
    "surfacesf.conicplus": "x y z l m n ux uy uz",
    "surfacesf.conicplusopd": "opd x y z l m n ux uy uz",
    "surfacesf.legsurf": "x y z l m n ux uy uz",
    "zernsurf.zernphase": "opd x y z l m n ux uy uz",
    "zernsurf.tracezernrot": "x y z l m n ux uy uz",
}
ALLCOLS = "opd x y z l m n ux uy uz".split()


def wolter_params(r0, z0, psi=1.):
    """woltsurf.f95:18-23 with libm scalars (math.*, i.e. glibc -- NOT numpy's SIMD trig)."""
    alpha = .25 * math.atan(r0 / z0)
    thetah = 2 * (1 + 2 * psi) / (1 + psi) * alpha
    thetap = 2 * psi / (1 + psi) * alpha
    p = z0 * math.tan(4 * alpha) * math.tan(thetap)
    d = z0 * math.tan(4 * alpha) * math.tan(4 * alpha - thetah)
    e = math.cos(4 * alpha) * (1 + math.tan(4 * alpha) * math.tan(thetah))
    return p, d, e


def primrad(z, r0=R0, z0=Z0, psi=PSI):
    p, d, e = wolter_params(r0, z0, psi)
    return math.sqrt(p ** 2 + 2 * p * z + (4 * e ** 2 * p * d) / (e ** 2 - 1))


def secrad(z, r0=R0, z0=Z0, psi=PSI):
    """conicsolve.py:29-37"""
    p, d, e = wolter_params(r0, z0, psi)
    return math.sqrt(e ** 2 * (d + z) ** 2 - z ** 2)


def noll_orders(nterms):
    """Standard (Noll) ordering consistent with specialFunctions.f95:139-140
    ("Z1 = 1, Z2 = 2 r cos t; sin if odd index, cos if even").  Returns (rorder, aorder) with
    aorder<0 -> sin, >0 -> cos, exactly the convention zernset consumes (:216-228)."""
    ro, ao = [], []
    n = 0
    j = 1
    while len(ro) < nterms:
        ms = [mm for mm in range(n + 1) if (n - mm) % 2 == 0]
        for mm in ms:
            if mm == 0:
                ro.append(n); ao.append(0); j += 1
            else:
                # two terms; even j gets cos, odd j gets sin
                for _ in range(2):
                    ro.append(n); ao.append(mm if j % 2 == 0 else -mm); j += 1
            if len(ro) >= nterms:
                break
        n += 1
    return np.array(ro[:nterms], dtype=np.int32), np.array(ao[:nterms], dtype=np.int32)


def _unit(rng, num, nsign=None, spread=0.3):
    l = rng.uniform(-spread, spread, num)
    m = rng.uniform(-spread, spread, num)
    n = np.sqrt(1. - l * l - m * m)
    if nsign is None:
        n *= rng.choice([-1., 1.], num)
    else:
        n *= nsign
    return l, m, n


def _randrays(rng, num, nsign=None):
    l, m, n = _unit(rng, num, nsign)
    ux, uy, uz = _unit(rng, num)
    return dict(opd=rng.uniform(0, 10, num), x=rng.uniform(-300, 300, num), y=rng.uniform(-300, 300, num),
                z=rng.uniform(-100, 100, num), l=l, m=m, n=n, ux=ux, uy=uy, uz=uz)


def annulus_rays(rng, num, rin, rout, z, zhat=-1.):
    """sources.annulus semantics (sources.py:91-127) with an explicit Generator."""
    rho = np.sqrt(rin ** 2 + rng.random(num) * (rout ** 2 - rin ** 2))
    theta = rng.random(num) * 2 * np.pi
    zero = np.zeros(num)
    return dict(opd=zero.copy(), x=rho * np.cos(theta), y=rho * np.sin(theta), z=np.full(num, float(z)),
                l=zero.copy(), m=zero.copy(), n=np.full(num, float(zhat)), ux=zero.copy(), uy=zero.copy(),
                uz=zero.copy())


def primary_input(rng, num, offaxis=0.):
    """Rays just above the primary (common-focus frame, travelling -z), optional off-axis tilt."""
    rin, rout = primrad(Z0), primrad(Z0 + LMIR)
    c = annulus_rays(rng, num, rin, rout, Z0 + 2 * LMIR)
    if offaxis:
        c["l"][:] = math.sin(offaxis)
        c["n"][:] = -math.sqrt(1. - math.sin(offaxis) ** 2)
    return c


def cases(num=2048, seed=1234):
    """Return a list of (name, fn_path, cols, args).  Secondary-mirror inputs are produced by
    tracing + reflecting off the primary with the C restatement (bit-identical to the reference)."""
    rng = np.random.default_rng(seed)
    out = []

    # ---- flat / flatopd: both signs of n, plus n==0 (Inf/NaN), z==0, huge z ---------------
    c = _randrays(rng, num)
    c["n"][0] = 0.
    c["z"][1] = 0.
    c["z"][2] = 1.2e6
    c["n"][3] = -0.0
    out.append(("flat_rand", "surfacesf.flat", c, ()))
    c = _randrays(rng, num)
    out.append(("flatopd_rand", "surfacesf.flatopd", c, (1.5,)))
    # far plane: delta ~ 1.2e3 so the fp32 rounding of delta is visible (quirk 1)
    c = _randrays(rng, num, nsign=-1.)
    c["z"] = rng.uniform(8000, 12000, num)
    out.append(("flat_far", "surfacesf.flat", c, ()))

    # ---- reflect -------------------------------------------------------------------------
    out.append(("reflect_rand", "transformationsf.reflect", _randrays(rng, num), ()))

    # ---- transform / itransform ----------------------------------------------------------
    tsets = {"translate": (0., 0., -8600., 0., 0., 0.),
             "rot_pi": (0., 0., 0., math.pi, 0., 0.),
             "general": (1.5, -2.25, 100.125, 0.3, -1.1, 2.7),
             "small": (1e-3, 2e-3, -3e-3, 1e-4, -2e-4, 3e-4)}
    for k, t in tsets.items():
        out.append(("transform_" + k, "transformationsf.transform", _randrays(rng, num), t))
        out.append(("itransform_" + k, "transformationsf.itransform", _randrays(rng, num), t))

    # ---- conic / conicopd ----------------------------------------------------------------
    for k, (R, K) in {"parab": (1000., -1.), "sphere": (-750., 0.), "hyper": (1200., -2.5),
                      "oblate": (900., 0.5)}.items():
        c = _randrays(rng, num, nsign=-1.)
        c["l"] *= 0.1
        c["m"] *= 0.1
        c["n"] = -np.sqrt(1. - c["l"] ** 2 - c["m"] ** 2)
        c["z"] = rng.uniform(50, 150, num)
        # edge rays: exactly axial (parabola special branch), on the vertex (s == 0), far miss
        c["l"][0] = c["m"][0] = 0.; c["n"][0] = -1.
        c["l"][1] = c["m"][1] = 0.; c["n"][1] = 1.
        c["x"][2] = c["y"][2] = c["z"][2] = 0.
        c["x"][3] = 5e4; c["l"][3] = 0.9; c["m"][3] = 0.; c["n"][3] = -math.sqrt(1 - 0.81)
        c["x"][4] = c["y"][4] = 0.; c["z"][4] = 0.; c["l"][4] = c["m"][4] = 0.; c["n"][4] = -1.
        out.append(("conic_" + k, "surfacesf.conic", c, (R, K)))
        c2 = {kk: v.copy() for kk, v in c.items()}
        out.append(("conicopd_" + k, "surfacesf.conicopd", c2, (R, K, 1.33)))

    # ---- Wolter-I primary: on-axis and 5 arcmin off-axis, plain and OPD --------------------
    out.append(("wolterprimary_onaxis", "woltsurf.wolterprimary", primary_input(rng, num), (R0, Z0, PSI)))
    out.append(("wolterprimary_offaxis", "woltsurf.wolterprimary",
                primary_input(rng, num, offaxis=5. / 60 * math.pi / 180), (R0, Z0, PSI)))
    out.append(("wolterprimary_psi2", "woltsurf.wolterprimary", primary_input(rng, num), (R0, Z0, 2.)))
    out.append(("wolterprimaryopd_offaxis", "woltsurf.wolterprimaryopd",
                primary_input(rng, num, offaxis=3. / 60 * math.pi / 180), (R0, Z0, PSI, 1.)))

    # ---- Wolter-I secondary: inputs = primary-traced + reflected rays (3-8 Newton its) -----
    from . import api
    port = api.load_port()
    for k, (off, psi) in {"onaxis": (0., PSI), "offaxis": (5. / 60 * math.pi / 180, PSI),
                          "psi2": (2. / 60 * math.pi / 180, 2.)}.items():
        c = primary_input(rng, num, offaxis=off)
        a = [c[kk] for kk in "x y z l m n ux uy uz".split()]
        port.woltsurf.wolterprimary(*a, R0, Z0, psi)
        port.transformationsf.reflect(*a[3:])
        out.append(("woltersecondary_" + k, "woltsurf.woltersecondary", c, (R0, Z0, psi)))

    # ---- Zernike sag surface: 36 terms, rad 50, tilted beam; rho == 0 ray -> NaN -----------
    ro, ao = noll_orders(36)
    coeff = rng.normal(0., 1e-3, 36)
    coeff[0] = 0.
    nz = 512 if num > 512 else num
    rho = np.sqrt(rng.random(nz)) * 50.
    th = rng.random(nz) * 2 * np.pi
    zero = np.zeros(nz)
    c = dict(opd=zero.copy(), x=rho * np.cos(th), y=rho * np.sin(th), z=np.full(nz, 10.),
             l=np.full(nz, 0.01), m=zero.copy(), n=np.full(nz, -math.sqrt(1 - 1e-4)),
             ux=zero.copy(), uy=zero.copy(), uz=zero.copy())
    c["x"][0] = c["y"][0] = 0.          # rho == 0: reference divides by rho -> NaN (zernsurf.f95:63-66)
    c["x"][1] = 1e-9; c["y"][1] = 0.    # tiny rho
    out.append(("tracezern_36", "zernsurf.tracezern", c, (coeff, ro, ao, 50.)))
    c2 = {kk: v.copy() for kk, v in c.items()}
    out.append(("tracezernopd_36", "zernsurf.tracezernopd", c2, (coeff, ro, ao, 50., 1.)))
    ro10, ao10 = noll_orders(10)
    c3 = {kk: v.copy() for kk, v in c.items()}
    out.append(("tracezern_10", "zernsurf.tracezern", c3, (coeff[:10] * 5, ro10, ao10, 50.)))

    # ---- radial grating: hub at origin, rays near y ~ hubdist; evanescent tail -> NaN -------
    hub = 11832.
    for k, (order, wave) in {"o1": (1., 2.4), "om3": (-3., 0.7), "evan": (1., 4.15)}.items():
        c = _randrays(rng, num)
        c["x"] = rng.uniform(-50, 50, num)
        c["y"] = hub + rng.uniform(-50, 50, num)
        g = 1.5 * math.pi / 180
        c["l"] = rng.normal(0, 1e-3, num)
        c["m"] = np.full(num, -math.cos(g)) + rng.uniform(-5e-5, 5e-5, num)
        c["n"] = np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2) * rng.choice([-1., 1.], num)
        out.append(("radgrat_" + k, "transformationsf.radgrat", c, (wave, 160. / hub, order)))
    c = {kk: v.copy() for kk, v in c.items()}
    c["y"] *= rng.choice([-1., 1.], num)       # radgratW takes the sign of n from sign(y)
    wv = rng.uniform(1., 5., num)
    out.append(("radgratw_o1", "transformationsf.radgratw", c, (wv, 160. / hub, 1.)))

    # ================= widening batch 1 (appended: earlier cases keep their random stream) ========
    # ---- SPO cones (Arcus): rays above the intersection plane, travelling -z; some miss ------
    for k, (R0s, F) in {"prim": (600., 12000.), "sec": (737.3, 12000.)}.items():
        tg = (.25 if k == "prim" else .75) * math.atan((R0s + .605 / 2) / F)
        c = annulus_rays(rng, num, R0s - .4, R0s + 1.2, 150. if k == "prim" else -20.)
        c["l"] = rng.normal(0, 2e-4, num); c["m"] = rng.normal(0, 2e-4, num)
        c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
        c["ux"] = rng.normal(size=num)
        c["l"][0] = 0.; c["m"][0] = 1.; c["n"][0] = 0.; c["x"][0] = 5e4      # misses the cone: det < 0
        out.append(("spocone_" + k, "woltsurf.spocone", c, (R0s, tg)))

    # ---- linear grating: scalar-like (broadcast) and per-ray order/wave; evanescent -> zeros --
    c = _randrays(rng, num)
    c["l"] = rng.uniform(-.01, .01, num); c["m"] = rng.uniform(-.9999, -.9990, num)
    c["n"] = np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2) * rng.choice([-1., 1.], num)
    out.append(("grat_scalar", "transformationsf.grat", c, (160., np.full(num, 1.), np.full(num, 2.4))))
    c = {kk: v.copy() for kk, v in c.items()}
    out.append(("grat_arrays", "transformationsf.grat", c,
                (200., rng.choice([-2., -1., 0., 1., 3.], num), rng.uniform(.5, 6., num))))

    # ---- sphere / cylinder (centre at the origin); far rays miss -> zeroed ---------------------
    for fn, tag in (("tracesphere", "sphere"), ("tracecyl", "cyl")):
        c = _randrays(rng, num)
        c["x"] = rng.uniform(-40, 40, num); c["y"] = rng.uniform(-40, 40, num); c["z"] = rng.uniform(150, 250, num)
        c["l"] *= .2; c["m"] *= .2
        c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
        c["x"][:3] = 500.                                # miss
        out.append((tag, "surfacesf." + fn, c, (100.,)))
        c2 = {kk: v.copy() for kk, v in c.items()}
        out.append((tag + "opd", "surfacesf." + fn + "opd", c2, (100., 1.5)))

    # ---- cylindrical conic y = c x^2 / (1 + sqrt(1-(1+k) c^2 x^2)), Newton -----------------------
    c = _randrays(rng, num)
    c["x"] = rng.uniform(-20, 20, num); c["y"] = rng.uniform(5, 10, num)
    c["l"] = rng.uniform(-.05, .05, num); c["n"] = rng.uniform(-.05, .05, num)
    c["m"] = -np.sqrt(1 - c["l"] ** 2 - c["n"] ** 2)
    out.append(("cylconic", "surfacesf.cylconic", c, (1. / 500., -1.)))
    c2 = {kk: v.copy() for kk, v in c.items()}
    out.append(("cylconic_k05", "surfacesf.cylconic", c2, (-1. / 800., .5)))

    # ---- paraxial lenses ----------------------------------------------------------------------
    out.append(("paraxial", "surfacesf.paraxial", _randrays(rng, num), (250.,)))
    out.append(("paraxialy", "surfacesf.paraxialy", _randrays(rng, num), (-75.5,)))

    # ---- refraction: normals on both sides, one exactly-normal ray (dot == 1 -> untouched) -----
    c = _randrays(rng, num)
    c["ux"], c["uy"], c["uz"] = _unit(rng, num, spread=.2)
    c["l"][0], c["m"][0], c["n"][0] = 0., 0., 1.
    c["ux"][0], c["uy"][0], c["uz"][0] = 0., 0., 1.
    out.append(("refract_in", "transformationsf.refract", c, (1., 1.5)))
    c2 = {kk: v.copy() for kk, v in c.items()}
    out.append(("refract_out", "transformationsf.refract", c2, (1.5, 1.)))   # incl. total internal reflection -> NaN

    # ================= widening batch 2 (appended) ================================================
    # ---- sinusoidally perturbed primary ------------------------------------------------------------
    out.append(("woltersine", "woltsurf.woltersine", primary_input(rng, num, offaxis=1e-4), (R0, Z0, 1e-4, 1. / 20.)))

    # ---- Legendre-Legendre deformed primary / secondary / ellipsoid primary (sub-aperture) ------------
    def sub_input(psi, off=0.):
        rin, rout = primrad(Z0, R0, Z0, psi), primrad(Z0 + LMIR, R0, Z0, psi)
        rho = np.sqrt(rin ** 2 + rng.random(num) * (rout ** 2 - rin ** 2))
        th = rng.random(num) * .5 - .25
        zero = np.zeros(num)
        return dict(opd=zero.copy(), x=rho * np.cos(th), y=rho * np.sin(th), z=np.full(num, Z0 + 2 * LMIR),
                    l=np.full(num, math.sin(off)), m=zero.copy(), n=np.full(num, -math.sqrt(1 - math.sin(off) ** 2)),
                    ux=zero.copy(), uy=zero.copy(), uz=zero.copy())
    llc = np.array([2e-4, -1e-4, 5e-5, 3e-5, -2e-5, 1e-5])
    lax = np.array([0, 1, 2, 0, 3, 5], dtype=np.int32)
    laz = np.array([0, 0, 1, 2, 2, 4], dtype=np.int32)
    c = sub_input(PSI, off=2e-4)
    out.append(("wolterprimll", "woltsurf.wolterprimll", c, (R0, Z0, Z0 + LMIR, Z0, .6, llc, lax, laz)))
    c2 = {kk: v.copy() for kk, v in c.items()}
    a = [c2[kk] for kk in "x y z l m n ux uy uz".split()]
    port.woltsurf.wolterprimary(*a, R0, Z0, PSI)
    port.transformationsf.reflect(*a[3:])
    out.append(("woltersecll", "woltsurf.woltersecll", c2, (R0, Z0, PSI, Z0 - 10., Z0 - 10. - LMIR, .6, llc, lax, laz)))
    S_ = 1e5
    out.append(("ellipsoidwoltll", "woltsurf.ellipsoidwoltll", sub_input(PSI),
                (R0, Z0, PSI, S_, Z0 + LMIR, Z0, .6, llc, lax, laz)))

    # ---- Wolter-Schwarzschild primary / secondary (Chase parameters from alpha = atan(r0/z0)/4) ---------
    alpha = .25 * math.atan(R0 / Z0)
    c = primary_input(rng, num, offaxis=1e-4)
    c["x"][:4] *= .9                      # inside the inner radius: the beta <= betas branch
    c["x"][4] = c["y"][4] = 0.            # on the axis -> NaN -> position restored
    out.append(("wsprimary", "woltsurf.wsprimary", c, (alpha, Z0, PSI)))
    c2 = {kk: v.copy() for kk, v in primary_input(rng, num).items()}
    a = [c2[kk] for kk in "x y z l m n ux uy uz".split()]
    port.woltsurf.wsprimary(*a, alpha, Z0, PSI)
    port.transformationsf.reflect(*a[3:])
    out.append(("wssecondary", "woltsurf.wssecondary", c2, (alpha, Z0, PSI)))

    # ================= widening batch 3 (appended) ================================================
    # ---- W-S back surfaces (substrate thickness 0.4 mm) -------------------------------------------
    c = primary_input(rng, num, offaxis=5e-5)
    c["x"] += .4 * c["x"] / np.hypot(c["x"], c["y"]); c["y"] += .4 * c["y"] / np.hypot(c["x"], c["y"])
    out.append(("wsprimaryback", "woltsurf.wsprimaryback", c, (alpha, Z0, PSI, .4)))
    c2 = {kk: v.copy() for kk, v in c.items()}
    a = [c2[kk] for kk in "x y z l m n ux uy uz".split()]
    port.woltsurf.wsprimaryback(*a, alpha, Z0, PSI, .4)
    port.transformationsf.reflect(*a[3:])
    out.append(("wssecondaryback", "woltsurf.wssecondaryback", c2, (alpha, Z0, PSI, .4)))

    # ---- torus (tangent-plane frame), conicplus (conic + even polynomial), Newton -------------------
    c = _randrays(rng, num)
    c["x"] = rng.uniform(-10, 10, num); c["y"] = rng.uniform(-10, 10, num); c["z"] = rng.uniform(5, 15, num)
    c["l"] *= .1; c["m"] *= .1
    c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    out.append(("torus", "surfacesf.torus", c, (200., 1500.)))
    pp = np.array([1e-6, -2e-10, 3e-14])
    c = _randrays(rng, num)
    c["x"] = rng.uniform(-30, 30, num); c["y"] = rng.uniform(-30, 30, num); c["z"] = rng.uniform(20, 40, num)
    c["l"] *= .1; c["m"] *= .1
    c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    out.append(("conicplus", "surfacesf.conicplus", c, (800., -.7, pp)))
    c2 = {kk: v.copy() for kk, v in c.items()}
    out.append(("conicplusopd", "surfacesf.conicplusopd", c2, (-600., .3, pp[:2], 1.5)))

    # ---- 2-D Legendre phase surface --------------------------------------------------------------
    c = _randrays(rng, num)
    c["x"] = rng.uniform(-12, 12, num); c["y"] = rng.uniform(-8, 8, num)       # some |x/xwidth| > 1
    c["l"] *= .05; c["m"] *= .05
    c["n"] = np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2) * rng.choice([-1., 1.], num)
    out.append(("legsurf", "surfacesf.legsurf", c, (10., 10., 1e-3, np.array([.5, -.2, .1, .05, .02]),
                                                     np.array([1, 2, 0, 3, 5], dtype=np.int32),
                                                     np.array([0, 1, 2, 3, 4], dtype=np.int32))))

    # ---- Zernike phase surface and the two-set rotated Zernike surface --------------------------------
    ro, ao = noll_orders(21)
    cf = rng.normal(0., 1e-3, 21)
    nz = 512 if num > 512 else num
    rho = np.sqrt(rng.random(nz)) * 50.
    th = rng.random(nz) * 2 * np.pi
    zero = np.zeros(nz)
    c = dict(opd=zero.copy(), x=rho * np.cos(th), y=rho * np.sin(th), z=zero.copy(),
             l=rng.normal(0, 1e-3, nz), m=rng.normal(0, 1e-3, nz), n=None, ux=zero.copy(), uy=zero.copy(),
             uz=np.ones(nz))
    c["n"] = np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2) * rng.choice([-1., 1.], nz)
    out.append(("zernphase", "zernsurf.zernphase", c, (cf, ro, ao, 50., 5e-4)))
    c = dict(opd=zero.copy(), x=rho * np.cos(th), y=rho * np.sin(th), z=np.full(nz, 10.),
             l=np.full(nz, 0.01), m=zero.copy(), n=np.full(nz, -math.sqrt(1 - 1e-4)),
             ux=zero.copy(), uy=zero.copy(), uz=zero.copy())
    ro2, ao2 = noll_orders(10)
    out.append(("tracezernrot", "zernsurf.tracezernrot", c, (cf, ro, ao, rng.normal(0., 2e-3, 10), ro2, ao2, 50., .37)))
    return out


def chain_config2(be, cols):
    """SURVEY 8d config 2 on a backend with f2py-style namespaces; `cols` is a dict of all ten
    columns at the annulus (z=0, n=-1).  transformations.transform negates its arguments
    (transformations.py:29), so the native call receives (+0,+0,+(z0+2L)) for a frame move of
    -(z0+2L)."""
    a = [cols[k] for k in "x y z l m n ux uy uz".split()]
    be.transformationsf.transform(*a, 0., 0., Z0 + 2 * LMIR, 0., 0., 0.)
    be.woltsurf.wolterprimary(*a, R0, Z0, PSI)
    be.transformationsf.reflect(*a[3:])
    be.woltsurf.woltersecondary(*a, R0, Z0, PSI)
    be.transformationsf.reflect(*a[3:])
    be.surfacesf.flat(*a)


def call(be, fn_path, cols, args):
    """Run one case in place on backend `be`."""
    mod, fn = fn_path.split(".")
    f = getattr(getattr(be, mod), fn)
    arrs = [cols[k] for k in SIG[fn_path].split()]
    f(*arrs, *args)


def rms_centroid(x, y):
    """analyses.py:16-30 (unweighted)."""
    cx, cy = np.average(x), np.average(y)
    return float(np.sqrt(np.average((x - cx) ** 2 + (y - cy) ** 2)))


# ---- wavefront sensing (reconstruct.f95) ----------------------------------------------------
def wavefront_cases():
    """(tag, xdim, ydim, binsize, x, y, l, m): rays filling an elliptical pupil inside an
    xdim x ydim lenslet grid (so the corners are invalid lenslets), slopes of a smooth wavefront
    plus noise.  Even, odd and mixed grid sizes take both bin formulas of southwellbin."""
    out = []
    for tag, xdim, ydim, bs, n, seed in (("g20x20", 20, 20, 1.0, 6000, 1), ("g21x17", 21, 17, 0.5, 5000, 2),
                                         ("g16x33", 16, 33, 2.0, 8000, 3), ("g9x9_sparse", 9, 9, 1.0, 60, 4)):
        rng = np.random.default_rng(1000 + seed)
        ax, ay = (xdim / 2 - 2) * bs, (ydim / 2 - 2) * bs
        x, y = rng.uniform(-ax, ax, n), rng.uniform(-ay, ay, n)
        keep = (x / ax) ** 2 + (y / ay) ** 2 < 1
        x, y = np.ascontiguousarray(x[keep]), np.ascontiguousarray(y[keep])
        l = 1e-4 * np.sin(x / (3 * bs)) + 2e-5 * y / ay + 1e-6 * rng.normal(size=len(x))
        m = 2e-4 * np.cos(y / (4 * bs)) - 3e-5 * x / ax + 1e-6 * rng.normal(size=len(x))
        out.append((tag, xdim, ydim, bs, x, y, np.ascontiguousarray(l), np.ascontiguousarray(m)))
    return out


def wavefront_run(be, case, criteria=1e-12, maxiter=10000):
    """southwellbin -> reconstruct as analyses.wavefront does (analyses.py:305-334: slopes swapped,
    reconstruct(y, x, 1e-12, dx, phase)).  Returns dict of the grids after each stage."""
    tag, xdim, ydim, bs, x, y, l, m = case
    xang, yang, phase = be.reconstruct.southwellbin(x, y, l, m, bs, xdim, ydim)
    res = {"bin_xang": xang.copy(order="F"), "bin_yang": yang.copy(order="F"), "bin_phase": phase.copy(order="F")}
    a, b, ph = yang.copy(order="F"), xang.copy(order="F"), phase.copy(order="F")
    res["phasec"] = be.reconstruct.reconstruct(a, b, criteria, bs, ph, maxiter)
    res["rec_xang"], res["rec_yang"], res["rec_phase"] = a, b, ph
    return res
