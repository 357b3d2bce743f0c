#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- regenerate tests/golden/*.npz from the REAL reference.

Runs only in the build container (needs /root/reference and oracle/_ref, i.e. the reference's
own Fortran compiled in place by oracle/Makefile).  What it records is DATA ONLY: seeded inputs
and the outputs the reference produced for them.  No reference source text is stored.

  native_cases.npz   every case of oracle/cases.py through the flang-compiled Fortran entry
                     points (surfacesf / woltsurf / zernsurf / transformationsf).
  chain_config2.npz  SURVEY 8d config-2 two-bounce chain, N=4096, + focal rmsCentroid.
  config1.npz        BASELINE config 1: np.random.seed(0); sources.pointsource(0.01, 1e5)
                     (the reference's own sources.py) -> transform -> flat -> reflect;
                     per-column SHA-256 of the outputs + the first 64 rays.
  wrappers.npz       the reference's own transformations.py imported over a `PyXFocus`
                     package shim whose `transformationsf` is the compiled Fortran: coords
                     bookkeeping, applyT/applyTPos, `ind` handling, vignette, radgrat dispatch.
  sources.npz        the reference's own sources.py under fixed np.random seeds.

  analyses.npz       the reference's own analyses.py and surfaces.py (focus*) and
                     transformations.steerX/Y run on focal bundles: centroid, rms*, rho, rhocdf, hpd
                     (weighted and not), hpdY, analytic*Plane, indAngle/grazeAngle, measureOPD,
                     rmsPoint, focusI/X/Y, steerX/Y.  analyses.py imports the un-vendored
                     `utilities` package and astropy at module level (analyses.py:6-13) for
                     functions OUTSIDE this set; those imports are satisfied by EMPTY module
                     objects whose attributes raise when touched, so nothing but the reference's
                     own NumPy code can have produced a recorded number.
  alignment.npz      examples/single_pass_alignment.py (examples/axro/singlePassAlignment.py:20-190) through the
                     reference's own modules: Zernike wavefront (OPD), misaligned primary / pair, applyT, centroids.
  ws_heights.npz     examples/ws_axial_heights.py (examples/axro/axialHeights.py:50-133) through the reference's own
                     modules: W-S radius probes, off-axis W-S pair, focusI, hpd, rmsCentroid.
  retrace.npz        examples/retrace_error.py (examples/axro/retraceError.py:20-65) through the reference's own modules:
                     pointTo, double pass of a misaligned / Legendre-deformed primary off a retro-reflector.
  config5.npz        a small Arcus-style multi-shell trace (sector.py:540-757 call shape) through
                     the reference's own surfaces.py / transformations.py / analyses.py wrappers
                     over the compiled Fortran: per-shell SPO pair, grating stage, focusY.

Usage: make_golden.py [--only legacy|analyses|config5[,...]]   (legacy = the seven files above config5/analyses)
"""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import api, cases  # noqa: E402

REF = os.environ.get("PXF_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
NGOLD = 256


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def shim_reference_python(ref):
    """Make `import PyXFocus.transformationsf / transformMod` resolve: the former to the compiled
    Fortran (f2py-style adapters of oracle/api.py), the latter to the reference's own file."""
    pkg = types.ModuleType("PyXFocus")
    pkg.__path__ = []
    sys.modules["PyXFocus"] = pkg
    tf = types.ModuleType("PyXFocus.transformationsf")
    for k, v in vars(ref.transformationsf).items():
        setattr(tf, k, v)
    sys.modules["PyXFocus.transformationsf"] = tf
    pkg.transformationsf = tf
    pkg.transformMod = _load_by_path("PyXFocus.transformMod", os.path.join(REF, "transformMod.py"))
    tran = _load_by_path("PyXFocus.transformations", os.path.join(REF, "transformations.py"))
    src = _load_by_path("PyXFocus.sources", os.path.join(REF, "sources.py"))
    return tran, src


class _Absent(types.ModuleType):
    """Stand-in for a module the reference imports but this image lacks (`utilities.*`, astropy).
    It has NO behaviour: any attribute access raises, so a recorded value cannot depend on it."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise RuntimeError("reference code touched %s.%s, which is absent from this image" % (self.__name__, name))


def _untouchable(modname, name):
    def f(*a, **k):
        raise RuntimeError("reference code called %s.%s, which is absent from this image" % (modname, name))
    return f


def shim_reference_analyses(ref):
    """Import the reference's own analyses.py and surfaces.py by path.  Needs shim_reference_python
    first.  PyXFocus.{surfacesf,woltsurf,zernsurf,reconstruct} -> the compiled Fortran adapters;
    PyXFocus.conicsolve -> the reference's file; un-vendored third-party modules -> _Absent."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    for name in ("utilities", "utilities.imaging", "utilities.imaging.man", "utilities.imaging.zernikemod",
                 "astropy", "astropy.io", "astropy.io.fits", "PyXFocus.specialfunctions"):
        if name not in sys.modules:
            m = _Absent(name)
            m.__path__ = []
            sys.modules[name] = m
    for name, names in (("utilities.imaging.fitting", ("circle", "circleMerit")),
                        ("utilities.imaging.analysis", ("ptov", "rms"))):
        m = types.ModuleType(name)                 # `from X import a, b` needs the names to exist
        for n in names:
            setattr(m, n, _untouchable(name, n))
        sys.modules[name] = m
    sys.modules["utilities"].imaging = sys.modules["utilities.imaging"]
    for sub in ("man", "zernikemod", "fitting", "analysis"):
        object.__setattr__(sys.modules["utilities.imaging"], sub, sys.modules["utilities.imaging." + sub])
    object.__setattr__(sys.modules["astropy"], "io", sys.modules["astropy.io"])
    object.__setattr__(sys.modules["astropy.io"], "fits", sys.modules["astropy.io.fits"])
    pkg = sys.modules["PyXFocus"]
    for sub in ("surfacesf", "woltsurf", "zernsurf", "reconstruct"):
        m = types.ModuleType("PyXFocus." + sub)
        for k, v in vars(getattr(ref, sub)).items():
            setattr(m, k, v)
        sys.modules["PyXFocus." + sub] = m
        setattr(pkg, sub, m)
    pkg.specialfunctions = sys.modules["PyXFocus.specialfunctions"]
    pkg.transformations = sys.modules["PyXFocus.transformations"]
    pkg.conicsolve = _load_by_path("PyXFocus.conicsolve", os.path.join(REF, "conicsolve.py"))
    anal = _load_by_path("PyXFocus.analyses", os.path.join(REF, "analyses.py"))
    pkg.analyses = anal
    surf = _load_by_path("PyXFocus.surfaces", os.path.join(REF, "surfaces.py"))
    pkg.surfaces = surf
    return anal, surf


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def focal_bundles(ref):
    """Two focal-plane bundles for the analyses goldens: (a) the config-2 chain of chain_config2.npz
    stopped 3 mm short of focus (a finite, ring-shaped spot), (b) a random converging bundle with
    wide direction spread and non-trivial normals.  Dict of ten columns each."""
    rng = np.random.default_rng(2)
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    a = cases.annulus_rays(rng, 4096, rin, rout, 0.)
    cases.chain_config2(ref, a)
    nine = [a[k] for k in "x y z l m n ux uy uz".split()]
    ref.transformationsf.transform(*nine, 0., 0., -3., 0., 0., 0.)
    ref.surfacesf.flat(*nine)
    rng = np.random.default_rng(9)
    b = cases._randrays(rng, 3001, nsign=-1.)
    b["x"], b["y"] = rng.normal(1.5, .8, 3001), rng.normal(-.7, 2.1, 3001)
    b["z"] = rng.normal(0., .01, 3001)
    b["opd"] = rng.uniform(0., 1., 3001)
    return {"ring": a, "blob": b}


def gold_analyses(ref, tran, anal, surf):
    """The reference's own analyses.py (:16-187, :231-244), surfaces.focus* (:502-519) and
    transformations.steerX/Y (:79-89) on the bundles of focal_bundles()."""
    blob = {}
    for tag, c in focal_bundles(ref).items():
        rays = [c[k] for k in cases.ALLCOLS]
        n = len(rays[1])
        rng = np.random.default_rng(31)
        wts = {"none": None, "smooth": 1. + .5 * np.sin(np.arange(n) * .37), "heavy": rng.pareto(1.5, n) + .01}
        for k, v in zip(cases.ALLCOLS, rays):
            blob["%s/in/%s" % (tag, k)] = v.copy()
        for wtag, w in wts.items():
            pre = "%s/%s/" % (tag, wtag)
            if w is not None:
                blob[pre + "w"] = w
            blob[pre + "centroid"] = np.array(anal.centroid(rays, weights=w))
            blob[pre + "rmsCentroid"] = np.array(anal.rmsCentroid(rays, weights=w))
            blob[pre + "rmsX"] = np.array(anal.rmsX(rays, weights=w))
            blob[pre + "rmsY"] = np.array(anal.rmsY(rays, weights=w))
            blob[pre + "rmsPoint"] = np.array(anal.rmsPoint(rays, [.1, -.2, .3], weights=w))
            blob[pre + "rho_cent"] = anal.rho(rays, weights=w, cent=True)
            blob[pre + "hpd"] = np.array(anal.hpd(rays, weights=w))
            blob[pre + "hpdY"] = np.array(anal.hpdY(rays, weights=w))
            r, cdf = anal.rhocdf(rays, weights=w)
            blob[pre + "rhocdf_r"], blob[pre + "rhocdf_cdf"] = r, cdf
            blob[pre + "analyticImagePlane"] = np.array(anal.analyticImagePlane(rays, weights=w))
            blob[pre + "analyticXPlane"] = np.array(anal.analyticXPlane(rays, weights=w))
            blob[pre + "analyticYPlane"] = np.array(anal.analyticYPlane(rays, weights=w))
            for fname in ("focusI", "focusX", "focusY"):
                rr = tran.copy_rays(rays)
                coords = tran.newCoords()
                dz = getattr(surf, fname)(rr, weights=w, coords=coords)
                blob[pre + fname + "/dz"] = np.array(dz)
                blob[pre + fname + "/coords"] = np.array(coords)
                for k, v in zip(cases.ALLCOLS, rr):
                    if k in "xy":                   # flat leaves z = 0, u = (0,0,1), direction unchanged
                        blob[pre + fname + "/out/" + k] = v
                assert not rr[3].any() and (rr[9] == 1).all() and all(np.array_equal(rr[j], rays[j]) for j in (4, 5, 6))
        blob[tag + "/rho_origin"] = anal.rho(rays)
        blob[tag + "/measureOPD"] = anal.measureOPD(rays, [.1, -.2, .3])
        blob[tag + "/measureOPD_ray"] = anal.measureOPD(rays, [r[7] for r in rays])
        blob[tag + "/rmsPoint_ray"] = np.array(anal.rmsPoint(rays, [r[7] for r in rays]))
        blob[tag + "/indAngle"] = anal.indAngle(rays)
        blob[tag + "/grazeAngle"] = anal.grazeAngle(rays)
        sel = np.random.default_rng(5).random(n) < .3
        blob[tag + "/sel"] = sel
        blob[tag + "/indAngle_sel"] = anal.indAngle(rays, ind=sel)
        blob[tag + "/grazeAngle_sel"] = anal.grazeAngle(rays, ind=sel)
        blob[tag + "/indAngle_normal"] = anal.indAngle(rays, normal=[0., .6, .8])
        blob[tag + "/indAngle_normal_sel"] = anal.indAngle(rays, ind=sel, normal=[0., .6, .8])
        if tag == "blob":         # interpolateVec (analyses.py:189-230; scipy griddata): first 400 rays, 9 x 7 grid
            sub = [r[:400].copy() for r in rays]
            for meth in ("linear", "nearest", "cubic"):
                res, dx, dy = anal.interpolateVec(sub, 4, 9, 7, method=meth)
                blob["%s/interp/%s" % (tag, meth)] = res
                blob["%s/interp/dxdy" % tag] = np.array([dx, dy])
            blob["%s/interp/polar" % tag] = anal.interpolateVec(sub, 5, 8, 8, polar=True)[0]
            blob["%s/interp/vec" % tag] = anal.interpolateVec(sub, 0, 6, 6, xr=[0., 3.], yr=[-3., 1.], interpVec=sub[1] * sub[2])[0]
        for fname in ("steerX", "steerY"):
            rr = tran.copy_rays(rays)
            coords = tran.newCoords()
            getattr(tran, fname)(rr, coords=coords)
            blob["%s/%s/coords" % (tag, fname)] = np.array(coords)
            for k, v in zip(cases.ALLCOLS, rr):
                blob["%s/%s/out/%s" % (tag, fname, k)] = v
    np.savez_compressed(os.path.join(OUT, "analyses.npz"), **blob)
    print("analyses:", len(blob), "arrays")


CONFIG5_GOLD = {"N": 16, "shell_stride": 8, "modules_per_row": 2, "seed": 0, "order": -3, "wave": 2.4}


def gold_config5(tran, src, anal, surf):
    """examples/arcus_sector.py (the call shapes of the reference's examples/arcus/sector.py) bound
    to the reference's OWN sources / surfaces / transformations / analyses modules over the
    compiled Fortran, NumPy columns.  Records the focal bundle, weights, focusY shift, coords."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import arcus_sector as A
    a = A.Api(src, surf, tran, anal, np)
    g = CONFIG5_GOLD
    np.random.seed(g["seed"])
    out = A.traceArcus(a, g["N"], order=g["order"], wave=g["wave"], shell_stride=g["shell_stride"],
                       modules_per_row=g["modules_per_row"])
    blob = {"par/" + k: np.array(v) for k, v in g.items()}
    for k, v in zip(cases.ALLCOLS, out["rays"]):
        blob["out/" + k] = v
    blob["weights"] = out["weights"]
    blob["dz"] = np.array(out["dz"])
    blob["grating_rows"] = np.array(out["grating_rows"])
    blob["hubdist"], blob["angle"] = np.array(out["hubdist"]), np.array(out["angle"])
    blob["coords"] = np.array(out["coords"])
    r, w = out["rays"], out["weights"]
    blob["centroid_w"] = np.array(anal.centroid(r, weights=w))
    blob["hpd_w"] = np.array(anal.hpd(r, weights=w))
    blob["rmsY_w"] = np.array(anal.rmsY(r, weights=w))
    np.savez_compressed(os.path.join(OUT, "config5.npz"), **blob)
    print("config5: %d rays, %d grating rows, focusY dz = %.9f, line centroid (%.6f, %.6f)" % (
        len(r[1]), out["grating_rows"], out["dz"], *blob["centroid_w"]))


ALIGN_GOLD = {"seed": 0, "nterms": 21, "rad": 125., "num": 3000, "masknum": 15,
              "primalign": [0, 0, 0, 1e-5, -2e-5, 0.], "detalign": [.1, 0, .5, 0, 1e-4, 0.], "secalign": [0, 0, .02, 0, 1e-5, 0.]}


def gold_alignment(tran, src, anal, surf):
    """examples/single_pass_alignment.py (the reference's examples/axro/singlePassAlignment.py:20-190) on the
    reference's OWN modules: Hartmann-mask wavefront off a Zernike surface (OPD variants), through a misaligned
    primary / mirror pair with coords bookkeeping and applyT(inverse), slit centroids by boolean weights."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import single_pass_alignment as S
    a = S.Api(src, surf, tran, anal, np, sys.modules["PyXFocus.conicsolve"])
    g = ALIGN_GOLD
    np.random.seed(g["seed"])
    ro, ao = cases.noll_orders(g["nterms"])
    coeff = np.random.normal(0, 1e-4, g["nterms"])
    rays, mask = S.createWavefront(a, g["rad"], g["num"], coeff, ro, ao, masknum=g["masknum"])
    blob = {"par/" + k: np.array(v) for k, v in g.items()}
    blob["coeff"], blob["rorder"], blob["aorder"] = coeff, ro, ao
    for k, v in zip(cases.ALLCOLS, rays):
        blob["wavefront/" + k] = v.copy()
    blob["mask"] = mask
    cen, r = S.traceThroughPrimary(a, [x.copy() for x in rays], mask, g["masknum"], primalign=np.array(g["primalign"]),
                                   detalign=np.array(g["detalign"]))
    blob["primary/cen"] = cen
    for k, v in zip(cases.ALLCOLS, r):
        blob["primary/" + k] = v
    cen, r = S.traceThroughPair(a, [x.copy() for x in rays], mask, g["masknum"], primalign=np.array(g["primalign"]),
                                secalign=np.array(g["secalign"]))
    blob["pair/cen"] = cen
    for k, v in zip(cases.ALLCOLS, r):
        blob["pair/" + k] = v
    np.savez_compressed(os.path.join(OUT, "alignment.npz"), **blob)
    print("alignment: %d rays, first slit centroid primary (%.6f, %.6f), pair (%.6f, %.6f)" % (
        len(rays[1]), blob["primary/cen"][0, 0], blob["primary/cen"][1, 0], cen[0, 0], cen[1, 0]))


def gold_ws_heights(tran, src, anal, surf):
    """examples/ws_axial_heights.py (the reference's examples/axro/axialHeights.py:50-133) on the reference's
    own modules: one-ray radius probes, a W-S pair with an off-axis kick, focusI, hpd, rmsCentroid."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import ws_axial_heights as W
    a = W.Api(src, surf, tran, anal, np, sys.modules["PyXFocus.conicsolve"])
    blob = {"prim_z": np.array([1e4 + 50., 1e4 + 150., 1e4 + 250.]), "sec_z": np.array([1e4 - 50., 1e4 - 150., 9750.])}
    blob["primrad"] = np.array([W.wsPrimrad(a, z, 220., 1e4, psi) for z in blob["prim_z"] for psi in (1., .6)])
    blob["secrad"] = np.array([W.wsSecrad(a, z, 220., 1e4, psi) for z in blob["sec_z"] for psi in (1., .6)])
    for tag, kw in (("onaxis", dict(offaxis=0.)), ("offaxis", dict(offaxis=1e-4)), ("zeta", dict(offaxis=5e-5, psi=.7))):
        np.random.seed(3)
        smin, smax, f, hpd, rms, rays = W.traceZeta(a, 1e4 + 50., num=4000, **kw)
        blob[tag + "/scalars"] = np.array([smin, smax, f, hpd, rms])
        for k, v in zip(cases.ALLCOLS, rays):
            blob["%s/%s" % (tag, k)] = v
    np.random.seed(4)
    blob["merit"] = np.array(W.onaxisMerit(a, 1e4 + 50., float(blob["onaxis/scalars"][1]), num=1500), dtype=float)
    np.savez_compressed(os.path.join(OUT, "ws_heights.npz"), **blob)
    print("ws_heights: primrad %s, traceZeta(onaxis) %s" % (blob["primrad"][:2], blob["onaxis/scalars"]))


RETRACE_CASES = (("perfect", {}), ("tilted", dict(primalign=np.array([0., 0., 0., 2e-5, -1e-5, 0.]))),
                 ("shifted", dict(primalign=np.array([.02, -.01, .5, 0., 0., 0.]))),
                 ("figure", dict(primCoeffs=(np.array([2e-4, -1e-4, 5e-5, 3e-5]), np.array([1, 2, 0, 3]), np.array([0, 0, 2, 1])))),
                 ("figure+tilt", dict(primCoeffs=(np.array([2e-4, -1e-4, 5e-5, 3e-5]), np.array([1, 2, 0, 3]), np.array([0, 0, 2, 1])),
                                      primalign=np.array([0., 0., 0., -1e-5, 2e-5, 1e-4]))))


def gold_retrace(tran, src, anal, surf):
    """examples/retrace_error.py (the reference's examples/axro/retraceError.py:20-65) on the reference's own
    modules: pointTo(reverse=1.), a double pass of a (misaligned, Legendre-deformed) Wolter primary off a flat
    retro-reflector, the dropped vignette, the primary's focus."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import retrace_error as R
    a = R.Api(src, surf, tran, anal, np, sys.modules["PyXFocus.conicsolve"])
    blob = {}
    for tag, kw in RETRACE_CASES:
        np.random.seed(6)
        rays, theta = R.tracePrimary(a, num=3000, **kw)
        for k, v in zip(cases.ALLCOLS, rays):
            blob["%s/%s" % (tag, k)] = np.asarray(v)
        blob[tag + "/theta"] = theta
        blob[tag + "/scalars"] = np.array(list(anal.centroid(rays)) + [anal.rmsCentroid(rays)])
    np.savez_compressed(os.path.join(OUT, "retrace.npz"), **blob)
    print("retrace: centroid+rms perfect %s, tilted %s" % (blob["perfect/scalars"], blob["tilted/scalars"]))


def write_provenance():
    """Where the fixtures' numbers were made: the reference's host scalars (sin/cos of frame angles, Van Speybroeck
    constants) and the W-S libm calls come from THIS machine's C library -- glibc selects sin / cos / sincos / tan per
    CPU through ifunc, and the bit-identical parity claims are tied to that selection."""
    import json
    import platform
    cpu = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    flags = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("flags"):
                fl = ln.split(":", 1)[1].split()
                flags = " ".join(f for f in ("fma", "avx2", "avx512f") if f in fl)
                break
    except OSError:
        pass
    json.dump({"libc": " ".join(platform.libc_ver()), "machine": platform.machine(), "cpu": cpu, "cpu_isa": flags,
               "numpy": np.__version__, "python": platform.python_version(),
               "compiler": "AMD flang (ROCm 7.2.0) -O2 for oracle/_ref; gcc -O2 -ffp-contract=off for oracle/pxf_oracle.c",
               "note": "fixtures are data produced by the reference's own Fortran/Python on this machine; the GPU suite "
                       "checks the box's libm against the restated routines first "
                       "(tests/test_gpu_parity.py::test_box_libm_is_the_one_the_kernels_restate)"},
              open(os.path.join(OUT, "PROVENANCE.json"), "w"), indent=1)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma list: legacy (the seven round-1 files), analyses, config5, alignment, ws_heights, retrace, provenance")
    only = [t for t in ap.parse_args().only.split(",") if t]
    os.makedirs(OUT, exist_ok=True)
    write_provenance()
    if only == ["provenance"]:
        return
    ref = api.load_ref()
    tran, src = shim_reference_python(ref)
    if not only or set(only) & {"analyses", "config5", "alignment", "ws_heights", "retrace"}:
        anal, surf = shim_reference_analyses(ref)
        if not only or "retrace" in only:
            gold_retrace(tran, src, anal, surf)
        if not only or "alignment" in only:
            gold_alignment(tran, src, anal, surf)
        if not only or "ws_heights" in only:
            gold_ws_heights(tran, src, anal, surf)
        if not only or "analyses" in only:
            gold_analyses(ref, tran, anal, surf)
        if not only or "config5" in only:
            gold_config5(tran, src, anal, surf)
    if only and "legacy" not in only:
        return

    # ---- native cases ------------------------------------------------------------------
    blob = {}
    names = []
    for name, fn, cols, args in cases.cases(num=NGOLD, seed=1234):
        names.append(name + "|" + fn)
        for k, v in cols.items():
            blob["%s/in/%s" % (name, k)] = v.copy()
        for i, a in enumerate(args):
            blob["%s/arg/%d" % (name, i)] = np.asarray(a)
        cases.call(ref, fn, cols, args)
        for k in cases.SIG[fn].split():
            blob["%s/out/%s" % (name, k)] = cols[k]
    blob["__names__"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "native_cases.npz"), **blob)
    print("native_cases:", len(names), "cases")

    # ---- config-2 chain ----------------------------------------------------------------
    rng = np.random.default_rng(2)
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    c = cases.annulus_rays(rng, 4096, rin, rout, 0.)
    cin = {k: v.copy() for k, v in c.items()}
    cases.chain_config2(ref, c)
    blob = {"in/" + k: v for k, v in cin.items()}
    blob.update({"out/" + k: v for k, v in c.items()})
    blob["rms_centroid"] = np.array(cases.rms_centroid(c["x"], c["y"]))
    np.savez_compressed(os.path.join(OUT, "chain_config2.npz"), **blob)
    print("chain_config2: rmsCentroid = %.6e mm" % blob["rms_centroid"])

    # ---- config 1 (plumbing) -----------------------------------------------------------
    np.random.seed(0)
    rays = src.pointsource(0.01, 100000)
    ins = [r.copy() for r in rays]
    tran.transform(rays, 0, 0, -100, 0, 0, 0)
    a = rays[1:]
    ref.surfacesf.flat(*a)                    # surfaces.flat(rays) -> surfacesf.flat (surfaces.py:27-28)
    tran.reflect(rays)
    blob = {"seed": np.array(0), "ang": np.array(0.01), "num": np.array(100000)}
    for k, vin, vout in zip(cases.ALLCOLS, ins, rays):
        blob["sha_in/" + k] = np.array(sha(vin))
        blob["sha_out/" + k] = np.array(sha(vout))
        blob["head_in/" + k] = vin[:64].copy()
        blob["head_out/" + k] = vout[:64].copy()
    np.savez_compressed(os.path.join(OUT, "config1.npz"), **blob)
    print("config1: sha(x_out) =", blob["sha_out/x"])

    # ---- reference Python wrappers over the compiled Fortran ------------------------------
    rng = np.random.default_rng(77)
    blob = {}

    def put(prefix, rays):
        for k, v in zip(cases.ALLCOLS, rays):
            blob["%s/%s" % (prefix, k)] = np.array(v, copy=True)

    def fresh(n=300):
        c = cases._randrays(rng, n)
        return [c[k] for k in cases.ALLCOLS]

    # coords bookkeeping through a sequence of transforms + applyT both ways
    rays = fresh()
    put("coords/in", rays)
    orig = tran.copy_rays(rays)
    coords = tran.newCoords()
    seq = [(1., -2., 3., 0.1, 0.2, -0.3), (0., 0., 50., 0., 0., 1.2), (-4., 0.5, 0., -0.7, 0., 0.)]
    for t in seq:
        tran.transform(rays, *t, coords=coords)
    tran.itransform(rays, 0.3, 0.2, 0.1, 0.01, -0.02, 0.03, coords=coords)
    blob["coords/seq"] = np.array(seq)
    blob["coords/iseq"] = np.array([0.3, 0.2, 0.1, 0.01, -0.02, 0.03])
    put("coords/out", rays)
    blob["coords/matrices"] = np.array(coords)
    put("coords/applyT_fwd", tran.applyT(orig, coords))
    put("coords/applyT_inv", tran.applyT(rays, coords, inverse=True))
    blob["coords/applyTPos_fwd"] = np.array(tran.applyTPos(orig[1], orig[2], orig[3], coords))
    blob["coords/applyTPos_inv"] = np.array(tran.applyTPos(rays[1], rays[2], rays[3], coords, inverse=True))
    blob["coords/rotationM"] = tran.rotationM(0.3, -0.4, 0.5)
    blob["coords/rotationM_inv"] = tran.rotationM(0.3, -0.4, 0.5, inverse=True)
    blob["coords/translationM"] = tran.translationM(1., 2., 3.)

    # `ind` handling: bool mask, index array, np.where tuple
    for tag, mk in {"mask": lambda n: rng.random(n) < 0.4,
                    "index": lambda n: rng.permutation(n)[: n // 3],
                    "where": lambda n: np.where(rng.random(n) < 0.5)}.items():
        rays = fresh()
        ind = mk(len(rays[0]))
        put("ind_%s/in" % tag, rays)
        blob["ind_%s/ind" % tag] = np.asarray(ind[0] if isinstance(ind, tuple) else ind)
        tran.transform(rays, 1., 2., 3., 0.1, 0.2, 0.3, ind=ind)
        tran.reflect(rays, ind=ind)
        tran.itransform(rays, -1., 0., 2., 0., 0.5, 0., ind=ind)
        put("ind_%s/out" % tag, rays)

    # vignette: default (|d|^2 > .1), mask, index
    rays = fresh()
    kill = rng.random(len(rays[0])) < 0.3
    for k in (4, 5, 6):
        rays[k][kill] = 0.
    rays[6][5] = np.nan
    put("vignette/in", rays)
    put("vignette/default", tran.vignette(rays))
    msk = rng.random(len(rays[0])) < 0.5
    blob["vignette/mask"] = msk
    put("vignette/bymask", tran.vignette(rays, ind=msk))
    idx = rng.permutation(len(rays[0]))[:57]
    blob["vignette/index"] = idx
    put("vignette/byindex", tran.vignette(rays, ind=idx))

    # radgrat dispatch: scalar wave -> radgrat, ndarray -> radgratw; with and without ind
    hub = 11832.
    for tag in ("scalar", "array", "scalar_ind", "array_ind"):
        c = cases._randrays(rng, 300)
        c["x"] = rng.uniform(-50, 50, 300)
        c["y"] = hub + rng.uniform(-50, 50, 300)
        c["l"] = rng.normal(0, 1e-3, 300)
        c["m"] = np.full(300, -0.99965)
        c["n"] = np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2) * rng.choice([-1., 1.], 300)
        rays = [c[k] for k in cases.ALLCOLS]
        put("radgrat_%s/in" % tag, rays)
        wave = rng.uniform(1., 3., 300) if tag.startswith("array") else 2.4
        ind = (rng.random(300) < 0.5) if tag.endswith("ind") else None
        blob["radgrat_%s/wave" % tag] = np.asarray(wave)
        if ind is not None:
            blob["radgrat_%s/ind" % tag] = ind
        tran.radgrat(rays, 160. / hub, 1., wave, ind=ind)
        put("radgrat_%s/out" % tag, rays)

    # pointTo
    rays = fresh()
    put("pointto/in", rays)
    tran.pointTo(rays, 1., -2., 5000., reverse=1.)
    put("pointto/out", rays)
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **blob)
    print("wrappers:", len(blob), "arrays")

    # ---- sources -----------------------------------------------------------------------
    blob = {}
    for tag, seed, fn, args in [("pointsource", 11, src.pointsource, (0.01, 64)),
                                ("circularbeam", 12, src.circularbeam, (50., 64)),
                                ("annulus", 13, src.annulus, (220., 220.65, 64)),
                                ("annulus_zhat", 14, src.annulus, (100., 150., 64, 1.)),
                                ("subannulus", 15, src.subannulus, (220., 221., 0.5, 64)),
                                ("subannulus_zhat", 16, src.subannulus, (220., 221., 0.5, 64, -1.)),
                                ("xslit", 17, src.xslit, (-5., 5., 64)),
                                ("rectArray", 18, src.rectArray, (10., 20., 8)),
                                ("rectbeam", 19, src.rectbeam, (3., 4., 64)),
                                ("convergingbeam", 20, src.convergingbeam, (8400., 220., 221., -0.2, 0.3, 64, 1.5)),
                                ("convergingbeam2", 21, src.convergingbeam2, (8400., 200., 220., -10., 10., 64, 1.5)),
                                ("gaussianBeam", 22, src.gaussianBeam, (0.01, 64)),
                                ("fanBeam", 23, src.fanBeam, (0.01, 0.02, 8)),
                                ("circFan", 24, src.circFan, (0.02, 4, 6))]:
        np.random.seed(seed)
        rays = fn(*args)
        blob[tag + "/seed"] = np.array(seed)
        blob[tag + "/args"] = np.array(args, dtype=float)
        for k, v in zip(cases.ALLCOLS, rays):
            blob["%s/%s" % (tag, k)] = np.asarray(v, dtype=float)
    np.savez_compressed(os.path.join(OUT, "sources.npz"), **blob)
    print("sources:", len(blob), "arrays")

    # ---- wavefront sensing: southwellbin + reconstruct through the compiled reconstruct.f95 ----
    blob = {}
    for case in cases.wavefront_cases():
        tag, xdim, ydim, bs, x, y, l, m = case
        blob[tag + "/dims"] = np.array([xdim, ydim])
        blob[tag + "/binsize"] = np.array(bs)
        for k, v in zip("xylm", (x, y, l, m)):
            blob["%s/in_%s" % (tag, k)] = v
        for k, v in cases.wavefront_run(ref, case).items():
            blob["%s/%s" % (tag, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "wavefront.npz"), **blob)
    print("wavefront:", len(blob), "arrays")

    # ---- host design helpers: the reference's own conicsolve.py (imported by path; the pyplot star
    # import it performs needs a non-interactive backend), plus transformations.skew ---------------
    os.environ.setdefault("MPLBACKEND", "Agg")
    con = _load_by_path("refconicsolve", os.path.join(REF, "conicsolve.py"))
    blob = {}
    z = np.linspace(8400., 8500., 7)
    for tag, (r0, z0, psi) in (("a", (220., 8400., 1.)), ("b", (150., 10000., 2.5)), ("c", (610., 8400., .5))):
        blob[tag + "/par"] = np.array([r0, z0, psi])
        blob[tag + "/primrad"] = con.primrad(z, r0, z0, psi)
        blob[tag + "/secrad"] = con.secrad(z - 200., r0, z0, psi)
        blob[tag + "/woltparam"] = np.array(con.woltparam(r0, z0))
        blob[tag + "/primfocus"] = np.array(con.primfocus(r0, z0))
        blob[tag + "/primsag"] = np.array(con.primsag(z0 + 100., r0, z0))
        blob[tag + "/secsag"] = np.array(con.secsag(z0 - 100., z0 - 5., r0, z0, psi))
        blob[tag + "/rGoal_to_rMax"] = np.array(con.rGoal_to_rMax(r0 + 1., z0, z0 + 100.))
        blob[tag + "/wsRMS"] = np.array(con.wsRMS(psi, 1e-3, 6.5e-3, 100., z0))
        blob[tag + "/wsFoc"] = np.array(con.wsFoc(np.array([0., 1., 5.]), psi, 100., z0, 6.5e-3))
        S = 50000.
        blob[tag + "/ellipsoidFunction"] = np.array(con.ellipsoidFunction(S, psi, r0, z0))
        blob[tag + "/ellipsoidRad"] = con.ellipsoidRad(S, psi, r0, z0, z)
        blob[tag + "/ehSecRad"] = con.ehSecRad(S, psi, r0, z0, z - 200.)
        blob[tag + "/ellipsoidSag"] = np.array(con.ellipsoidSag(S, psi, r0, z0, z0 + 100., z0))
        P, a, b, e, f = con.ellipsoidFunction(S, psi, r0, z0)
        blob[tag + "/solveS"] = np.array(con.solveS(P, a, b, e, f, r0 + .3, .2, z0 + 50., 1e-3, -2e-3,
                                                    -np.sqrt(1 - 5e-6)))
    blob["skew"] = tran.skew(np.array([1., -2., 3.5]))
    # the reference's own southwell.py over the compiled reconstruct.f95 (its `import reconstruct`
    # resolved to the f2py-style adapter of oracle/api.py); data of its example(), 60 x 60
    rec = types.ModuleType("reconstruct")
    rec.reconstruct = ref.reconstruct.reconstruct
    sys.modules["reconstruct"] = rec
    sw = _load_by_path("refsouthwell", os.path.join(REF, "southwell.py"))
    xg, yg = np.meshgrid(np.linspace(-1, 1, 60), np.linspace(-1, 1, 60))
    img = np.polynomial.legendre.legval2d(xg, yg, [[0, 1, 0], [0, .5, 0], [1, 0, 0]])
    gx, gy = np.gradient(img)
    rad = np.sqrt(xg ** 2 + yg ** 2)
    gx[rad > 1] = np.nan
    gy[rad > 1] = np.nan
    blob["southwell/gx"], blob["southwell/gy"] = gx.copy(), gy.copy()
    blob["southwell/out"] = sw.southwell(gx, gy, 1e-12, 1.)
    grt = _load_by_path("refgrating", os.path.join(REF, "grating.py"))
    inc, wv, dd = np.array([1.5, 1.7, 2.0]) * np.pi / 180, np.array([2.4e-6, 3.6e-6, 4.2e-6]), 1. / 6033.
    blob["grating/args"] = np.array([inc, wv])
    for nm in ("blazeYaw", "blazeAngle", "blazeAngle2"):
        blob["grating/" + nm] = getattr(grt, nm)(inc, wv, 3, dd)
    blob["grating/litBetaAlpha"] = np.array(grt.litBetaAlpha(inc, wv, 3, dd))
    blob["grating/eta"] = grt.eta(inc, 2 * inc)
    blob["grating/yaw"] = grt.yaw(inc, 2 * inc)
    np.savez_compressed(os.path.join(OUT, "design.npz"), **blob)
    print("design:", len(blob), "arrays")


if __name__ == "__main__":
    main()
