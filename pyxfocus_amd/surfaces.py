"""Drop-in for the hot-path part of PyXFocus ``surfaces.py`` on ROCm tensors.

Same names, argument order and defaults as the reference (surfaces.py:14-47, 104-113, 219-243,
502-519).  Every function mutates the ray list in place and returns None (``focus*`` return the
total focus shift, like the reference).
"""
import numpy as np

from . import analyses
from . import _fused
from . import conicsolve as con
from . import transformations as tran
from .native import surfacesf as surf
from .native import woltsurf as wolt
from .native import zernsurf as zern
from .transformations import ind_to_mask

__all__ = ["flat", "zernsurf", "conic", "wolterprimary", "wolterprimarynode", "woltersecondary", "focus",
           "focusX", "focusY", "focusI", "zmodes", "sphere", "tanSphere", "cyl", "cylconic", "paraxial",
           "paraxialY", "spoCone", "spoPrimary", "spoSecondary", "woltersine", "primaryLL", "secondaryLL",
           "wsPrimary", "wsSecondary", "ellipsoidPrimary", "ellipsoidSecondary", "ellipsoidPrimaryLL",
           "ellipsoidSecondaryLL", "zernphase", "zernsurfrot", "conicplus", "torus", "legSurf", "wsPrimaryB",
           "wsSecondaryB", "oapCollimate", "zernconic"]


def flat(rays, ind=None, nr=None):
    """Trace rays to the XY plane (surfaces.py:14-29).  As in the reference, `ind` takes
    precedence and the ind path does not update the OPD."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if ind is not None:
        surf.flat(x, y, z, l, m, n, ux, uy, uz, mask=ind_to_mask(ind, x.shape[0], x.device))
    elif nr is not None:
        surf.flatopd(x, y, z, l, m, n, ux, uy, uz, opd, nr)
    else:
        surf.flat(x, y, z, l, m, n, ux, uy, uz)
    return


def zmodes(N):
    """Default (rorder, aorder) for N standard Zernike terms.

    The reference takes these from `utilities.imaging.zernikemod.zmodes`, a package that is not
    vendored with PyXFocus and pinned nowhere (surfaces.py:10,36-37), so this default is PARITY
    UNPINNED.  It is the Noll ordering that matches the comment in specialFunctions.f95:139-140
    (Z1 = 1, Z2 = 2 r cos t; within a pair the even index is the cosine term), with the sign
    convention zernset consumes: aorder > 0 -> cos, < 0 -> sin, 0 -> radial."""
    ro, ao = [], []
    n, j = 0, 1
    while len(ro) < N:
        for mm in range(n % 2, n + 1, 2):
            if mm == 0:
                ro.append(n); ao.append(0); j += 1
            else:
                for _ in range(2):
                    ro.append(n); ao.append(mm if j % 2 == 0 else -mm); j += 1
        n += 1
    return np.array(ro[:N], dtype=np.int32), np.array(ao[:N], dtype=np.int32)


def zernsurf(rays, coeff, rad, rorder=None, aorder=None, nr=None):
    """Zernike sag surface about z=0; coordinates are the usual arctan2(y,x) (surfaces.py:31-47)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if rorder is None:
        rorder, aorder = zmodes(np.size(coeff))
    if nr is None:
        zern.tracezern(x, y, z, l, m, n, ux, uy, uz, coeff, np.array(rorder), np.array(aorder), rad)
    else:
        zern.tracezernopd(opd, x, y, z, l, m, n, ux, uy, uz, coeff, np.array(rorder), np.array(aorder), rad, nr)
    return


def zernconic(rays, R, K, coeff, rad, rorder=None, aorder=None, nr=None):
    """EXTENSION -- not in PyXFocus: a conic (radius of curvature R, conic constant K, vertex at the
    origin, as `conic`) perturbed by a Zernike sag (as `zernsurf`): BASELINE config 4's
    "Zernike-perturbed conic".  Newton trace, tolerance 1e-10 mm, normal as in `zernsurf`."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if rorder is None:
        rorder, aorder = zmodes(np.size(coeff))
    zern.tracezernconic(opd if nr is not None else None, x, y, z, l, m, n, ux, uy, uz, R, K, coeff,
                        np.array(rorder), np.array(aorder), rad, nr=0. if nr is None else nr)
    return


def zernphase(rays, coeff, rad, wave, rorder=None, aorder=None):
    """Standard Zernike phase surface as in ZEMAX; wavelength, radius, coeff in mm (surfaces.py:49-60)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if rorder is None:
        rorder, aorder = zmodes(np.size(coeff))
    zern.zernphase(opd, x, y, z, l, m, n, ux, uy, uz, coeff, np.array(rorder), np.array(aorder), rad, wave)
    return


def zernsurfrot(rays, coeff1, coeff2, rad, rot, rorder1=None, aorder1=None, rorder2=None, aorder2=None):
    """Zernike surface with two Zernike sets, the second rotated by `rot` (surfaces.py:62-79)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if rorder1 is None:
        rorder1, aorder1 = zmodes(np.size(coeff1))
    if rorder2 is None:
        rorder2, aorder2 = zmodes(np.size(coeff2))
    zern.tracezernrot(x, y, z, l, m, n, ux, uy, uz, coeff1, np.array(rorder1), np.array(aorder1), coeff2,
                      np.array(rorder2), np.array(aorder2), rad, rot)
    return


def conicplus(rays, R, K, p, nr=None):
    """Conic with radius of curvature R and conic constant K plus even polynomial terms p
    (surfaces.py:115-124)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if nr is not None:
        surf.conicplusopd(opd, x, y, z, l, m, n, ux, uy, uz, R, K, p, nr)
    else:
        surf.conicplus(x, y, z, l, m, n, ux, uy, uz, R, K, p)
    return


def torus(rays, rin, rout):
    """Toroidal surface: outer radius in the xy plane, inner radius orthogonal (surfaces.py:158-164)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    surf.torus(x, y, z, l, m, n, ux, uy, uz, rin, rout)
    return


def legSurf(rays, xwidth, ywidth, order, coeff, xo, yo):
    """Diffract rays from a phase surface defined by 2-D Legendre coefficients; rays must already be
    in the x,y plane (surfaces.py:206-217)."""
    x, y, z, l, m, n, ux, uy, uz = rays[1:]
    surf.legsurf(x, y, z, l, m, n, ux, uy, uz, xwidth, ywidth, order, np.asarray(coeff).flatten(),
                 np.asarray(xo).flatten(), np.asarray(yo).flatten())
    return


def oapCollimate(rays, efl, oapangle, nr=None):
    """Trace rays to a collimating off-axis paraboloid; origin at the nominal OAP intersection, -z
    toward the OAP focus, collimated beam leaves along -y.  Returns a NEW ray list in the original
    frame (surfaces.py:126-156)."""
    fp = efl * (1 + np.cos(oapangle)) / 2
    coords = tran.newCoords()
    tran.transform(rays, 0, 0, -efl, 0, 0, 0, coords=coords)
    tran.transform(rays, 0, 0, 0, np.pi - oapangle, 0, 0, coords=coords)
    tran.transform(rays, 0, 0, -fp, 0, 0, 0, coords=coords)
    conic(rays, fp * 2, -1, nr=nr)
    tran.reflect(rays)
    return tran.applyT(rays, coords, inverse=True)


def conic(rays, R, K, nr=None):
    """Conic surface with radius of curvature R and conic constant K (surfaces.py:104-113)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if nr is not None:
        surf.conicopd(opd, x, y, z, l, m, n, ux, uy, uz, R, K, nr)
    else:
        surf.conic(x, y, z, l, m, n, ux, uy, uz, R, K)
    return


def sphere(rays, rad, nr=None):
    """Spherical surface centred on the origin (surfaces.py:81-89)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if nr is not None:
        surf.tracesphereopd(opd, x, y, z, l, m, n, ux, uy, uz, rad, nr)
    else:
        surf.tracesphere(x, y, z, l, m, n, ux, uy, uz, rad)
    return


def tanSphere(rays, rad, nr=None):
    """Spherical surface tangent to the XY plane; positive radius curves toward +Z (surfaces.py:91-102)."""
    tran.transform(rays, 0, 0, rad, 0, 0, 0)
    sphere(rays, rad, nr=nr)
    tran.transform(rays, 0, 0, -rad, 0, 0, 0)
    return


def cyl(rays, rad, nr=None):
    """Cylinder about the y axis through the origin (surfaces.py:166-178)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if nr is not None:
        surf.tracecylopd(opd, x, y, z, l, m, n, ux, uy, uz, rad, nr)
    else:
        surf.tracecyl(x, y, z, l, m, n, ux, uy, uz, rad)
    return


def cylconic(rays, rad, k):
    """Cylindrical conic (surfaces.py:180-184)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    surf.cylconic(x, y, z, l, m, n, ux, uy, uz, rad, k)


def paraxial(rays, F):
    """Ideal paraxial lens in the xy plane, axis along z (surfaces.py:186-194)."""
    x, y, z, l, m, n, ux, uy, uz = rays[1:]
    surf.paraxial(x, y, z, l, m, n, ux, uy, uz, F)
    return


def paraxialY(rays, F):
    """Ideal paraxial cylindrical lens acting on y only (surfaces.py:196-204)."""
    x, y, z, l, m, n, ux, uy, uz = rays[1:]
    surf.paraxialy(x, y, z, l, m, n, ux, uy, uz, F)
    return


def spoCone(rays, R0, tg, ind=None):
    """SPO cone with intersection radius R0 and slope angle tg; the XY plane is the SPO
    intersection plane (surfaces.py:403-419)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    wolt.spocone(x, y, z, l, m, n, ux, uy, uz, R0, tg, mask=ind_to_mask(ind, x.shape[0], x.device))
    return


def spoPrimary(rays, R0, F, d=.605, ind=None):
    """SPO primary with intersection radius R0 and focal length F (surfaces.py:421-430)."""
    tg = .25 * np.arctan((R0 + d / 2) / F)
    spoCone(rays, R0, tg, ind=ind)
    return


def spoSecondary(rays, R0, F, d=.605, ind=None):
    """SPO secondary with intersection radius R0 and focal length F (surfaces.py:432-441)."""
    tg = .75 * np.arctan((R0 + d / 2) / F)
    spoCone(rays, R0, tg, ind=ind)
    return


def wolterprimary(rays, r0, z0, psi=1., nr=None):
    """Wolter-I primary surface - no vignetting (surfaces.py:219-227)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    if nr is None:
        wolt.wolterprimary(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi)
    else:
        wolt.wolterprimaryopd(opd, x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, nr)
    return


def wolterprimarynode(rays, r0, z0, psi=1.):
    """Place Wolter node at current origin, focus at (-r0,0,-z0) (surfaces.py:229-236)."""
    tran.transform(rays, -r0, 0, -z0, 0, 0, 0)
    wolterprimary(rays, r0, z0, psi)
    tran.itransform(rays, -r0, 0, -z0, 0, 0, 0)
    return


def woltersecondary(rays, r0, z0, psi=1.):
    """Wolter-I secondary surface - no vignetting (surfaces.py:238-243)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    wolt.woltersecondary(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi)
    return


def woltersine(rays, r0, z0, amp, freq):
    """Wolter-I primary with a sinusoidal axial perturbation (surfaces.py:265-270)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    wolt.woltersine(x, y, z, l, m, n, ux, uy, uz, r0, z0, amp, freq)
    return


def secondaryLL(rays, r0, z0, psi, zmax, zmin, dphi, coeff, axial, az):
    """Legendre-Legendre deformed secondary, placed at focus (surfaces.py:292-299)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    wolt.woltersecll(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, zmax, zmin, dphi, coeff, axial, az)
    return


def primaryLL(rays, r0, z0, zmax, zmin, dphi, coeff, axial, az):
    """Legendre-Legendre deformed primary, placed at focus (surfaces.py:301-307)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    wolt.wolterprimll(x, y, z, l, m, n, ux, uy, uz, r0, z0, zmax, zmin, dphi, coeff, axial, az)
    return


def _ws(fn, rays, r0, z0, psi, check):
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    a, p, d, e = con.woltparam(r0, z0)
    if check is True:
        _fused.flush()           # (inside a fused() block: the positions must be current before and after)
        x0, y0, zz0 = x.clone(), y.clone(), z.clone()
    fn(x, y, z, l, m, n, ux, uy, uz, a, z0, psi)
    if check is True:
        _fused.flush()
        return (x0 == x) & (y0 == y) & (zz0 == z)
    return


def wsPrimaryB(rays, r0, z0, psi, thick, check=False):
    """W-S primary, back surface of a substrate of thickness `thick` (surfaces.py:349-365)."""
    return _ws(lambda *a: wolt.wsprimaryback(*a, thick), rays, r0, z0, psi, check)


def wsSecondaryB(rays, r0, z0, psi, thick, check=False):
    """W-S secondary, back surface (surfaces.py:385-401)."""
    return _ws(lambda *a: wolt.wssecondaryback(*a, thick), rays, r0, z0, psi, check)


def wsPrimary(rays, r0, z0, psi, check=False):
    """Wolter-Schwarzschild primary; Chase parameters from an equivalent W-I (alpha, z0).  With
    check=True returns the mask of rays that failed to converge (did not move) (surfaces.py:331-347)."""
    return _ws(wolt.wsprimary, rays, r0, z0, psi, check)


def wsSecondary(rays, r0, z0, psi, check=False):
    """Wolter-Schwarzschild secondary (surfaces.py:367-383)."""
    return _ws(wolt.wssecondary, rays, r0, z0, psi, check)


def ellipsoidPrimary(rays, R0, F, S, psi):
    """Primary of an ellipsoid-hyperboloid telescope; call at focus, like Wolter-I (surfaces.py:443-460)."""
    P, a, b, e, f = con.ellipsoidFunction(S, psi, R0, F)
    R = b ** 2 / a
    tran.transform(rays, 0, 0, F + f - P - a, 0, 0, 0)
    conic(rays, R, -e ** 2)
    tran.itransform(rays, 0, 0, F + f - P - a, 0, 0, 0)
    return


def ellipsoidSecondary(rays, R0, F, S, psi):
    """Secondary of an ellipsoid-hyperboloid telescope: a Wolter secondary with the effective psi
    of the ellipsoid geometry (surfaces.py:462-475)."""
    P, a, b, e, f = con.ellipsoidFunction(S, psi, R0, F)
    psi_eff = np.arctan(R0 / P) / (np.arctan(R0 / F) - np.arctan(R0 / P))
    woltersecondary(rays, R0, F, psi=psi_eff)
    return


def ellipsoidPrimaryLL(rays, R0, F, S, psi, zmax, zmin, dphi, coeff, axial, az):
    """Ellipsoid primary with Legendre-Legendre distortions (surfaces.py:477-487)."""
    opd, x, y, z, l, m, n, ux, uy, uz = rays
    wolt.ellipsoidwoltll(x, y, z, l, m, n, ux, uy, uz, R0, F, psi, S, zmax, zmin, dphi, coeff, axial, az)
    return


def ellipsoidSecondaryLL(rays, R0, F, S, psi, zmax, zmin, dphi, coeff, axial, az):
    """Hyperboloid secondary of the ellipsoid telescope with L-L distortions (surfaces.py:489-500)."""
    P, a, b, e, f = con.ellipsoidFunction(S, psi, R0, F)
    psi_eff = np.arctan(R0 / P) / (np.arctan(R0 / F) - np.arctan(R0 / P))
    secondaryLL(rays, R0, F, psi_eff, zmax, zmin, dphi, coeff, axial, az)
    return


def focus(rays, fn, weights=None, nr=None, coords=None, group=None):
    """Two passes of: analytic best-focus shift -> move frame -> flat (surfaces.py:502-510).

    ``group=`` (a torch.distributed process group, or True for the default one) is an addition for
    ray-sharded runs: `rays` is then this rank's shard and the moments behind `fn` are all-reduced,
    so every rank moves its frame by the SAME dz (the reference has one bundle, hence one shift)."""
    kw = {} if group is None else {"group": group}
    from_sums = getattr(fn, "_from_sums", None)
    if from_sums is None or len(rays) != 10:      # somebody else's fn: the reference's five calls as they stand
        dz1 = fn(rays, weights=weights, **kw)
        tran.transform(rays, 0, 0, dz1, 0, 0, 0, coords=coords)
        flat(rays, nr=nr)
        dz2 = fn(rays, weights=weights, **kw)
        tran.transform(rays, 0, 0, dz2, 0, 0, 0, coords=coords)
        flat(rays, nr=nr)
        return dz1 + dz2
    # The same five calls as THREE passes over the bundle (448 -> 248 B/ray): the moments behind fn; then frame change
    # + flat + the moments of the result in one launch; then frame change + flat in one launch.  Same bits throughout.
    dz1 = from_sums(analyses._moments(rays, weights, group=group))
    sums = analyses._focus_step(rays, dz1, weights, nr, True, group=group)
    if coords is not None:
        tran.update_coords(coords, 0, 0, dz1, 0, 0, 0)
    dz2 = from_sums(sums)
    analyses._focus_step(rays, dz2, None, nr, False)
    if coords is not None:
        tran.update_coords(coords, 0, 0, dz2, 0, 0, 0)
    return dz1 + dz2


def focusY(rays, weights=None, nr=None, coords=None, group=None):
    return focus(rays, analyses.analyticYPlane, weights=weights, nr=nr, coords=coords, group=group)


def focusX(rays, weights=None, nr=None, coords=None, group=None):
    return focus(rays, analyses.analyticXPlane, weights=weights, nr=nr, coords=coords, group=group)


def focusI(rays, weights=None, nr=None, coords=None, group=None):
    return focus(rays, analyses.analyticImagePlane, weights=weights, nr=nr, coords=coords, group=group)
