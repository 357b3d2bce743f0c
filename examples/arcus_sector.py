"""Arcus-style SPO petal + fanned radial gratings: the call shapes of the reference's
examples/arcus/sector.py (`defineSPOaperture` :392-538, `traceSPO` :540-634, `gratArray` :636-757,
the focal-plane tail of `traceArcus` :253-389) against the drop-in API.

The reference script itself cannot run anywhere but its author's machine (it reads reflectivity,
grating-efficiency and CCD tables from /home/rallured/Dropbox at import, :27-65).  What is kept here
is the sequence of PyXFocus calls and of element-wise operations on ray columns; what is replaced:

* the tabulated SPO reflectivity `sporef(graze, energy)` (:62-65) by a smooth analytic model
  `reflectivity(graze)`; grating efficiency and CCD QE (scalar factors, :268-274) are dropped;
* module/shell layout can be thinned (`rows`, `modules_per_row`, `shell_stride`) so that small
  bundles still exercise every code path; layout numbers are the reference's (:402-417);
* `gratArray`'s `while` loop (:685-702) gets an iteration cap (the reference has none).

Every function takes an `api` namespace (``sources, surf, tran, anal, xp``): `Api.amd()` binds
pyxfocus_amd + torch (device tensors), `Api(...)` can bind anything with the same call signatures --
tests/ binds the reference's own Python modules over the oracle to produce the goldens this file is
checked against (oracle/make_golden.py, tests/test_gpu_config5.py).
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

# module layout of the petal (sector.py:402-417)
ANG = [[-27.653, -16.592, -5.531, 5.531, 16.592, 27.653], [-23.174, -13.904, -4.635, 4.635, 13.904, 23.174],
       [-19.942, -11.965, -3.988, 3.988, 11.965, 19.942], [-16.065, -5.355, 5.355, 16.065],
       [-14.317, -4.772, 4.772, 14.317], [-12.972, -4.324, 4.324, 12.972], [-11.756, -3.919, 3.919, 11.756],
       [-10.791, -3.597, 3.597, 10.791]]
RIN = [320.443, 382.638, 444.833, 507.027, 569.222, 631.417, 693.612, 755.807]
ROUT = [376.243, 438.438, 500.633, 562.827, 625.022, 681.217, 749.412, 811.607]
SPAN = [50.159, 49.839, 49.614, 89.363, 82.476, 77.572, 86.892, 82.053]
FOCAL = 12e3
PITCH, PORE = .775, .605                 # shell pitch and pore height (sector.py:430, 540)
OUTERRAD = 798.895210199 + 2.            # sector.py:41
YAW = 0.022509613654884453               # sector.py:42
INC = 1.5 * math.pi / 180                # grating incidence angle (sector.py:254)
GAP, GRATLEN = 50., 95.                  # sector.py:253-254


class Api:
    """The five namespaces the trace needs.  xp = the array module of the ray columns."""

    def __init__(self, sources, surf, tran, anal, xp, conic=None):
        self.sources, self.surf, self.tran, self.anal, self.xp, self.conic = sources, surf, tran, anal, xp, conic

    @staticmethod
    def amd():
        import torch
        import pyxfocus_amd.analyses as anal
        import pyxfocus_amd.sources as sources
        import pyxfocus_amd.surfaces as surf
        import pyxfocus_amd.transformations as tran
        from pyxfocus_amd import conicsolve
        return Api(sources, surf, tran, anal, torch, conicsolve)

    def fused(self):
        """`with api.fused():` -- on pyxfocus_amd consecutive calls run as one launch (bit-identical);
        on any other binding (the reference's modules) a no-op."""
        import contextlib
        if self.xp is np:
            return contextlib.nullcontext()
        import pyxfocus_amd
        return pyxfocus_amd.fused()

    def zeros(self, n, like):
        if self.xp is np:
            return np.zeros(n)
        return self.xp.zeros(n, dtype=like.dtype, device=like.device)

    def asarray(self, a, like):
        if self.xp is np:
            return np.asarray(a, dtype=float)
        return self.xp.as_tensor(np.asarray(a, dtype=float), device=like.device)


def reflectivity(api, graze):
    """Stand-in for the tabulated B4C/Ir reflectivity of sector.py:62-65: smooth, in (0,1], falling
    with graze angle (radians)."""
    return api.xp.exp(-(graze / .03) ** 2)


def traceSPO(api, R, L, focVec, N, M, spanv, d=PORE, offX=0., offY=0., ang=0., coords=None):
    """One SPO module: a Python loop over its M shells (sector.py:540-634).  Returns the module's
    bundle (columns 1..6 filled, the rest zero, as the reference collects them :622-626) and the
    per-ray product of the two reflectivities."""
    xp = api.xp
    trays, ref = None, None
    for i in range(M):
        rays = api.sources.subannulus(R[i], R[i] + d, spanv[i], N, zhat=-1.)
        if trays is None:
            trays = [api.zeros(M * N, rays[1]) for _ in range(10)]
            ref = api.zeros(M * N, rays[1])
        api.tran.transform(rays, 0, 0, 0, 0, 0, ang)
        api.surf.spoPrimary(rays, R[i], focVec[i])
        rays = [rays[0], rays[1], rays[2], rays[3], rays[4] + offX, rays[5] + offY,
                -xp.sqrt(rays[6] ** 2 - offX ** 2 - offY ** 2), rays[7], rays[8], rays[9]]
        api.tran.reflect(rays)
        refl = reflectivity(api, api.anal.grazeAngle(rays))
        api.surf.spoSecondary(rays, R[i], focVec[i])
        api.tran.reflect(rays)
        ref[i * N:(i + 1) * N] = refl * reflectivity(api, api.anal.grazeAngle(rays))
        api.tran.transform(rays, 0, 0, -focVec[i], 0, 0, 0, coords=coords)
        for t in range(1, 7):
            trays[t][i * N:(i + 1) * N] = rays[t]
    return trays, ref


def module_geometry(i, shell_stride=1):
    """Shell radii, primary lengths, focal lengths, azimuthal spans and geometric weights per ray
    (before the 1/N) of the modules in row i (sector.py:430-461)."""
    R = np.arange(RIN[i], ROUT[i], PITCH)[::shell_stride]
    tg = .25 * np.arctan((R + PITCH / 2) / FOCAL)
    L = PITCH / np.tan(tg)
    focVec = np.sqrt(FOCAL ** 2 - R ** 2)
    spanv = 2 * np.arcsin(SPAN[i] / 2 / R)
    area = ((R + PORE) ** 2 - R ** 2) * spanv / 2 / 100.
    return R, L, focVec, spanv, area


def defineSPOaperture(api, N, offX=0., offY=0., gap=GAP, rows=range(8), modules_per_row=None, shell_stride=1):
    """All modules of the petal, concatenated, then moved to the plane of the outermost grating
    (sector.py:392-538).  Returns rays, weights, focal length of the last shell, longest primary of
    row 0 (as the reference takes it, :432-433), coords."""
    xp = api.xp
    coords = api.tran.newCoords()
    parts, wparts, lmax = [], [], None
    for i in rows:
        R, L, focVec, spanv, area = module_geometry(i, shell_stride)
        if lmax is None:
            lmax = L.max()
        M = len(R)
        vrad = np.maximum(0., 1 - abs(offX) / (PORE / 2 / L))          # radial / azimuthal vignetting factors
        vaz = np.maximum(0., 1 - abs(offY) / (.83 / 2 / L))
        tw = np.repeat(area / N * vrad * vaz, N)
        angs = ANG[i] if modules_per_row is None else ANG[i][:modules_per_row]
        for a in angs:
            trays, tref = traceSPO(api, R, L, focVec, N, M, spanv, offX=offX, offY=offY, ang=a * np.pi / 180)
            parts.append(trays)
            wparts.append(api.asarray(tw, trays[1]) * tref)
    cat = np.concatenate if xp is np else xp.cat
    rays = [cat([p[t] for p in parts]) for t in range(10)]
    weights = cat(wparts)
    api.tran.transform(rays, 0, 0, focVec[-1] - (L.max() + gap + GRATLEN), 0, 0, 0, coords=coords)
    api.surf.flat(rays)
    return rays, weights, focVec[-1], lmax, coords


def petal_layout(rows=range(8), modules_per_row=None, shell_stride=1):
    """Every shell of every module of the petal, in the order defineSPOaperture traces them: a list of
    (R, primary length L, focal length, azimuthal span, module angle [rad], geometric area)."""
    out = []
    for i in rows:
        R, L, focVec, spanv, area = module_geometry(i, shell_stride)
        angs = ANG[i] if modules_per_row is None else ANG[i][:modules_per_row]
        for a in angs:
            out.extend((float(R[k]), float(L[k]), float(focVec[k]), float(spanv[k]), a * np.pi / 180, float(area[k]))
                       for k in range(len(R)))
    return out


def defineSPOaperture_segmented(N, gap=GAP, rows=range(8), modules_per_row=None, shell_stride=1, source="philox",
                                seed=0, fast=False, rays=None):
    """defineSPOaperture (offX = offY = 0) the MI355X way: every shell of the petal is one SEGMENT of N rays
    of ONE bundle; the shells' sources are one launch (`sources.segments`, device Philox), their SPO pair
    -- frame rotation, spoPrimary, reflect, graze angle, spoSecondary, reflect, graze angle, move to the
    focus -- is ONE launch (`compile_segments`: per-shell constants staged in LDS, the two graze-angle
    columns written by the same kernel), and the move to the grating plane + flat is one more.  The
    reference (sector.py:556) loops over the 2732 shells in Python.  N must be even and >= 172.

    source="numpy": the shells' rays are drawn on the host from the np.random stream in the reference's
    order (parity runs); fast=True: the opt-in tolerance-mode kernel.  Returns what defineSPOaperture
    returns, plus the compiled program (re-runnable on a refilled bundle)."""
    import torch
    import pyxfocus_amd as pxf
    api = Api.amd()
    shells = petal_layout(rows, modules_per_row, shell_stride)
    ns = len(shells)
    n = ns * N
    dev = torch.device("cuda", torch.cuda.current_device())
    if rays is None:
        buf = torch.empty((10, n), dtype=torch.float64, device=dev)
        rays = [buf[i] for i in range(10)]
    if source == "philox":
        api.sources.segments("subannulus", [(sh[0], sh[0] + PORE, sh[3], -1.) for sh in shells], N, out=rays, seed=seed)
    else:
        host = np.zeros((10, n))
        for k, sh in enumerate(shells):            # sources.subannulus(R, R+d, span, N, zhat=-1) per shell
            rho = np.sqrt(sh[0] ** 2 + np.random.rand(N) * ((sh[0] + PORE) ** 2 - sh[0] ** 2))
            theta = np.random.rand(N) * sh[3] - sh[3] / 2.
            host[1, k * N:(k + 1) * N], host[2, k * N:(k + 1) * N] = rho * np.cos(theta), rho * np.sin(theta)
            host[6, k * N:(k + 1) * N] = -1.
        for i in range(10):
            rays[i].copy_(torch.from_numpy(host[i]))
    g1 = torch.empty(n, dtype=torch.float64, device=dev)
    g2 = torch.empty(n, dtype=torch.float64, device=dev)

    def shell(sl, R, foc, ang, o1, o2):
        api.tran.transform(sl, 0, 0, 0, 0, 0, ang)
        api.surf.spoPrimary(sl, R, foc)
        api.tran.reflect(sl)
        api.anal.grazeAngle(sl, out=o1)
        api.surf.spoSecondary(sl, R, foc)
        api.tran.reflect(sl)
        api.anal.grazeAngle(sl, out=o2)
        api.tran.transform(sl, 0, 0, -foc, 0, 0, 0)

    prog = pxf.compile_segments(rays, N, shell, [(sh[0], sh[2], sh[4], g1[k * N:(k + 1) * N], g2[k * N:(k + 1) * N])
                                                 for k, sh in enumerate(shells)], fast=fast)
    prog.run()
    for j in (0, 7, 8, 9):                               # traceSPO collects columns 1..6 only (:622-626)
        rays[j].zero_()
    weights = api.asarray(np.repeat([sh[5] / N for sh in shells], N), rays[1]) * reflectivity(api, g1) * reflectivity(api, g2)
    coords = api.tran.newCoords()
    lmax = module_geometry(list(rows)[0], shell_stride)[1].max()               # L.max() of the first row (:432-433)
    R, L, focVec, spanv, area = module_geometry(list(rows)[-1], shell_stride)
    with pxf.fused(fast=fast):
        api.tran.transform(rays, 0, 0, focVec[-1] - (L.max() + gap + GRATLEN), 0, 0, 0, coords=coords)
        api.surf.flat(rays)
    return rays, weights, focVec[-1], lmax, coords, prog


def gratArray(api, rays, outerrad, hubdist, angle, inc, l=GRATLEN, weights=None, order=0, blazeYaw=0., wave=1.,
              offX=0., coords=None, maxiter=10000, focus_kw=None, fan=None):
    """Rays leaving the petal -> fanned grating array rotated about the hub -> back to the frame of
    the outermost grating -> best line focus (sector.py:636-757).  Returns (focusY shift, number of
    grating rows visited).  fan: a callable with the signature of pyxfocus_amd.grating.fanArray that
    replaces the host `while` loop over grating rows (bit-identical, two launches instead of ~226 rounds of a
    dozen passes each); None = the reference's loop."""
    xp, tran, surf = api.xp, api.tran, api.surf
    pi = math.pi
    x, y = rays[1:3]
    with api.fused():
        tran.transform(rays, outerrad, 0, 0, 0, 0, 0, coords=coords)
        tran.transform(rays, 0, 0, 0, 0, 0, -pi / 2, coords=coords)
        tran.transform(rays, 0, 0, 0, -pi / 2 - angle + inc, 0, 0, coords=coords)
        tran.transform(rays, 0, 0, 0, 0, 0, blazeYaw, coords=coords)
        tran.transform(rays, 0, hubdist, 0, 0, 0, 0, coords=coords)
    ang = l * math.sin(inc - offX) / hubdist * .95
    if fan is not None:
        i = fan(rays, hubdist, l, ang, 160. / hubdist, order, wave, thr=.001, coords=coords, maxiter=maxiter)
        return _gratArray_tail(api, rays, outerrad, hubdist, angle, inc, ang, i, blazeYaw, weights, coords, focus_kw)
    indg = xp.abs(xp.arcsin(rays[6])) > .001
    surf.flat(rays, ind=indg)
    rho = -xp.sqrt(x ** 2 + y ** 2) * xp.sign(y)
    ind = xp.logical_and(rho > hubdist, rho < l + hubdist)
    ind2 = ind
    nrays = len(rays[1])
    i = 0
    prev = ind
    while int(xp.sum(prev)) < nrays:
        i = i + 1
        if i > maxiter:
            raise RuntimeError("gratArray: %d rays never reached a grating" % (nrays - int(xp.sum(prev))))
        if int(xp.sum(ind2)) > 0:
            tran.reflect(rays, ind=ind2)
            tran.radgrat(rays, 160. / hubdist, order, wave, ind=ind2)
        tran.transform(rays, 0, 0, 0, ang, 0, 0, coords=coords)
        indg = xp.abs(xp.arcsin(rays[6])) > .001
        indg = xp.logical_and(xp.logical_not(prev), indg)
        surf.flat(rays, ind=indg)
        rho = -xp.sqrt(x ** 2 + y ** 2) * xp.sign(y)
        ind = xp.logical_and(rho > hubdist, rho < l + hubdist)
        ind2 = xp.logical_and(xp.logical_not(prev), ind)
        prev = xp.logical_or(prev, ind)
    tran.reflect(rays, ind=ind2)
    tran.radgrat(rays, 160. / hubdist, order, wave, ind=ind2)
    return _gratArray_tail(api, rays, outerrad, hubdist, angle, inc, ang, i, blazeYaw, weights, coords, focus_kw)


def _gratArray_tail(api, rays, outerrad, hubdist, angle, inc, ang, i, blazeYaw, weights, coords, focus_kw):
    """Undo the frames of the grating array (sector.py:718-723), flat, focusY."""
    tran, surf, pi = api.tran, api.surf, math.pi
    with api.fused():
        tran.transform(rays, 0, 0, 0, -ang * i, 0, 0, coords=coords)
        tran.transform(rays, 0, -hubdist, 0, 0, 0, 0, coords=coords)
        tran.transform(rays, 0, 0, 0, 0, 0, -blazeYaw, coords=coords)
        tran.transform(rays, 0, 0, 0, pi / 2 + angle - inc, 0, 0, coords=coords)
        tran.transform(rays, 0, 0, 0, 0, 0, pi / 2, coords=coords)
        tran.transform(rays, -outerrad, 0, 0, 0, 0, 0, coords=coords)
        surf.flat(rays)
    return surf.focusY(rays, weights=weights, **(focus_kw or {})), i


def traceArcus(api, N, order=-3, wave=2.4, blazeYaw=YAW, inc=INC, l=GRATLEN, gap=GAP, offX=0., offY=0.,
               rows=range(8), modules_per_row=None, shell_stride=1, focus_kw=None, fan=None):
    """The petal end to end (sector.py:253-389 with bestFocus=None, then the focal-plane tail):
    returns a dict with the focal bundle, the weights, the focusY shift, the number of grating rows
    and the weighted centroid of the diffracted line."""
    rays, weights, minfoc, lmax, coords = defineSPOaperture(api, N, offX=offX, offY=offY, gap=gap, rows=rows,
                                                            modules_per_row=modules_per_row,
                                                            shell_stride=shell_stride)
    # sector.py:301 calls applyTPos(0,0,0,coords,inverse=True) with SCALARS, which NumPy >= 1.24 rejects
    # inside the reference's own applyTPos (ragged list); the same product on the host:
    foc = np.dot(coords[3], [0., 0., 0., 1.])[:3]
    hubdist = math.sqrt(OUTERRAD ** 2 + float(foc[2]) ** 2)
    angle = math.atan(OUTERRAD / (minfoc - (lmax + gap + GRATLEN)))
    dz, nrows = gratArray(api, rays, OUTERRAD, hubdist, angle, inc, l=l, weights=weights, order=order,
                          blazeYaw=blazeYaw, wave=wave, offX=offX, coords=coords, focus_kw=focus_kw, fan=fan)
    weights = weights * (1 - abs(offX) / inc)
    return {"rays": rays, "weights": weights, "dz": dz, "grating_rows": nrows, "hubdist": hubdist,
            "angle": angle, "coords": coords}


if __name__ == "__main__":
    import torch
    api = Api.amd()
    np.random.seed(0)
    out = traceArcus(api, int(sys.argv[1]) if len(sys.argv) > 1 else 200, shell_stride=8, modules_per_row=2)
    torch.cuda.synchronize()
    r, w = out["rays"], out["weights"]
    keep = torch.isfinite(r[2]) & torch.isfinite(r[1])
    r, w = api.tran.vignette(r, keep), w[keep]
    print("rays %d  grating rows %d  focusY dz %.6f mm  line centroid (%.4f, %.4f) mm  hpd %.4f mm" % (
        len(r[1]), out["grating_rows"], out["dz"], *api.anal.centroid(r, weights=w), api.anal.hpd(r, weights=w)))
