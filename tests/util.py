"""Shared helpers for the parity tests."""
import numpy as np

ALLCOLS = "opd x y z l m n ux uy uz".split()

# functions whose HIP kernels are bit-identical to the reference (add/mul/div/sqrt only, no FMA)
BITWISE = {"surfacesf.flat", "surfacesf.flatopd", "surfacesf.conic", "surfacesf.conicopd",
           "woltsurf.wolterprimary", "woltsurf.wolterprimaryopd", "woltsurf.woltersecondary",
           "transformationsf.transform", "transformationsf.itransform", "transformationsf.reflect",
           "woltsurf.spocone", "transformationsf.grat", "surfacesf.tracesphere", "surfacesf.tracesphereopd",
           "surfacesf.tracecyl", "surfacesf.tracecylopd", "surfacesf.cylconic", "surfacesf.paraxial",
           "surfacesf.paraxialy", "surfacesf.torus", "surfacesf.conicplus", "surfacesf.conicplusopd",
           "surfacesf.legsurf"}
# tolerance-parity functions: |gpu - ref| <= atol + rtol*|ref| + cond/|ref| per column.
# radgrat: l, m agree to an ulp; n = sn*sqrt(1-l^2-m^2) is ill-conditioned at grazing incidence
# (|n| -> 0): a 1-ulp (1.1e-16) difference in m moves n by 1.1e-16/|n|, hence the cond term on n.
TOL = {"transformationsf.radgrat": (1e-14, 1e-13, {"n": 3e-16}),
       "transformationsf.radgratw": (1e-14, 1e-13, {"n": 3e-16}),
       "zernsurf.tracezern": (2e-11, 1e-12, {}), "zernsurf.tracezernopd": (2e-11, 1e-12, {}),
       # refract: acos/asin/sin/cos from ocml instead of glibc
       "transformationsf.refract": (2e-15, 1e-14, {}),
       # batch 2: per-ray transcendentals from ocml instead of glibc; Newton to 1e-10 / 1e-7 / 1e-8
       "woltsurf.woltersine": (1e-10, 1e-13, {}), "woltsurf.wolterprimll": (1e-10, 1e-13, {}),
       "woltsurf.woltersecll": (1e-10, 1e-13, {}), "woltsurf.ellipsoidwoltll": (1e-10, 1e-13, {}),
       # W-S: the reference stops at |step| <= 1.e-8 mm and its iteration converges only linearly, so
       # two evaluations that differ in the last bits may stop one step apart: agreement is to the
       # algorithm's own convergence tolerance (positions), normals follow to ~1e-9
       "woltsurf.wsprimary": (2e-8, 1e-13, {}), "woltsurf.wssecondary": (2e-8, 1e-13, {}),
       "woltsurf.wsprimaryback": (2e-8, 1e-13, {}), "woltsurf.wssecondaryback": (2e-8, 1e-13, {}),
       "zernsurf.zernphase": (1e-13, 1e-12, {}), "zernsurf.tracezernrot": (2e-11, 1e-12, {})}


def bits_equal(a, b):
    """Bitwise equality of float64 arrays -- sign of zero included; NaN == NaN with any payload
    (x86 and gfx950 produce different default-NaN signs: -nan(ind) vs +qNaN)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64))
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def assert_cols(fn, got, want, cols, context=""):
    for k in cols:
        g, w = got[k], want[k]
        if fn in BITWISE:
            ok = bits_equal(g, w)
            assert ok.all(), "%s %s col %s: %d of %d rays differ (max |d| = %g)" % (
                context, fn, k, int((~ok).sum()), len(w), float(np.nanmax(np.abs(g - w))))
        else:
            atol, rtol, cond = TOL[fn]
            nan_same = np.isnan(g) == np.isnan(w)
            assert nan_same.all(), "%s %s col %s: NaN pattern differs on %d rays" % (
                context, fn, k, int((~nan_same).sum()))
            f = ~np.isnan(w)
            err = np.abs(g[f] - w[f])
            lim = atol + rtol * np.abs(w[f]) + cond.get(k, 0.) / np.maximum(np.abs(w[f]), 1e-300)
            assert (err <= lim).all(), "%s %s col %s: max err %g (limit %g)" % (
                context, fn, k, float(err.max()), float(lim[np.argmax(err - lim)]))


def to_dev(cols, device="cuda"):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in cols.items()}


def to_host(tcols):
    return {k: v.detach().cpu().numpy() for k, v in tcols.items()}


def call_gpu(fn_path, tcols, args, mask=None):
    """Run one oracle/cases.py case through pyxfocus_amd.native on device tensors."""
    from oracle import cases
    from pyxfocus_amd import native
    mod, fn = fn_path.split(".")
    f = getattr(getattr(native, mod), fn)
    arrs = [tcols[k] for k in cases.SIG[fn_path].split()]
    import torch
    a2 = []
    for a in args:
        if isinstance(a, np.ndarray) and a.dtype == np.int32:
            pass
        if fn_path in ("transformationsf.radgratw", "transformationsf.grat") and isinstance(a, np.ndarray) \
                and a.ndim == 1 and a.dtype == np.float64 and len(a) == len(arrs[0]):
            a = torch.from_numpy(a).to(arrs[0].device)
        a2.append(a)
    f(*arrs, *a2, mask=mask)


def zernike_order_check(port, nmax, masked=False, n=3001):
    """tracezern (stand-alone and as a fused program), tracezernOPD and zernphase with ALL terms up to
    radial order nmax (one kernel instantiation per nmax) against the oracle, 1e-12 relative.
    masked=True: through the MASKED instantiations with ~40 % of the rays selected; the oracle runs
    on the gathered rays (the reference's ind= idiom) and unselected rays must be untouched."""
    import torch
    import pyxfocus_amd as pxf
    from oracle import cases
    from pyxfocus_amd import native
    rng = np.random.default_rng(40 + nmax)
    nterms = (nmax + 1) * (nmax + 2) // 2
    ro, ao = cases.noll_orders(nterms)
    assert max(ro) == nmax
    ro, ao = np.asarray(ro, np.int32), np.asarray(ao, np.int32)
    coeff = rng.normal(0, 1e-3, nterms) / (1 + ro)
    coeff[0] = 0.
    c = {k: np.zeros(n) for k in ALLCOLS}
    rho, th = 1. + 44. * np.sqrt(rng.random(n)), rng.random(n) * 2 * np.pi
    c["x"], c["y"], c["z"] = rho * np.cos(th), rho * np.sin(th), np.full(n, 10.)
    c["l"], c["m"] = np.full(n, .01), rng.normal(0, 1e-3, n)
    c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    c["ux"], c["opd"] = rng.random(n), rng.random(n)          # recognisable content in columns that get overwritten
    nine = ALLCOLS[1:]
    msk = (rng.random(n) < .4) if masked else np.ones(n, bool)
    kw = {"mask": torch.from_numpy(msk).cuda()} if masked else {}

    def oracle(fn, src, cols, *args):
        want = {k: v.copy() for k, v in src.items()}
        sub = {k: np.ascontiguousarray(v[msk]) for k, v in src.items()}
        fn(*[sub[k] for k in cols], *args)
        for k in ALLCOLS:
            want[k][msk] = sub[k]
        return want

    def check(got, want, src, what):
        for k in ALLCOLS:
            g, w = got[k], want[k]
            assert np.array_equal(np.isnan(g), np.isnan(w)), (what, nmax, k)
            f = ~np.isnan(w)
            assert np.abs(g[f] - w[f]).max() <= 1e-12 * (1 + np.abs(w[f]).max()), (what, nmax, k, np.abs(g[f] - w[f]).max())
            assert bits_equal(g[~msk], src[k][~msk]).all(), (what, nmax, k, "unselected rays modified")

    want = oracle(port.zernsurf.tracezern, c, nine, coeff, ro, ao, 50.)
    t = to_dev(c)
    native.zernsurf.tracezern(*[t[k] for k in nine], coeff, ro, ao, 50., **kw)
    check(to_host(t), want, c, "tracezern")
    t = to_dev(c)
    with pxf.fused() as rec:
        native.zernsurf.tracezern(*[t[k] for k in nine], coeff, ro, ao, 50., **kw)
    if not masked:
        assert rec.launches == 1
    check(to_host(t), want, c, "tracezern fused")
    want = oracle(port.zernsurf.tracezernopd, c, ALLCOLS, coeff, ro, ao, 50., 1.4)
    t = to_dev(c)
    native.zernsurf.tracezernopd(*[t[k] for k in ALLCOLS], coeff, ro, ao, 50., 1.4, **kw)
    check(to_host(t), want, c, "tracezernopd")
    c2 = {k: v.copy() for k, v in c.items()}                    # zernphase: rays in the z = 0 plane
    c2["z"][:] = 0.
    want = oracle(port.zernsurf.zernphase, c2, ALLCOLS, coeff, ro, ao, 50., 1e-3)
    t = to_dev(c2)
    native.zernsurf.zernphase(*[t[k] for k in ALLCOLS], coeff, ro, ao, 50., 1e-3, **kw)
    check(to_host(t), want, c2, "zernphase")
