"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's Python wrapper layer over an
oracle backend (oracle/api.py: `load_port()` = our C restatement, `load_ref()` = the reference's
own Fortran compiled in place).

``bind(be)`` returns ``(sources, surf, tran, anal)`` namespaces with the reference's call
signatures on lists of ten NumPy float64 columns, so a caller written against the PyXFocus API
(examples/arcus_sector.py, the chains of bench.py) runs on the CPU checker unchanged and its
result can be compared with the same caller on pyxfocus_amd.  Each function cites the reference
lines it restates.  Pinned by tests/test_oracle.py against tests/golden/{analyses,config5,
wrappers}.npz, which were produced by the reference's OWN surfaces.py / transformations.py /
analyses.py (oracle/make_golden.py).  Nothing in pyxfocus_amd/ imports this.
"""
import math
import types

import numpy as np


def _rot(angle, axis):
    """transformMod.rotation_matrix(angle, direction) for a unit axis (transformMod.py:295-334)."""
    s, c = math.sin(angle), math.cos(angle)
    d = np.asarray(axis, dtype=float)
    R = np.diag([c, c, c])
    R += np.outer(d, d) * (1. - c)
    d = d * s
    R += np.array([[0., -d[2], d[1]], [d[2], 0., -d[0]], [-d[1], d[0], 0.]])
    M = np.identity(4)
    M[:3, :3] = R
    return M


def bind(be):
    nine = lambda rays: rays[1:]

    # ---- transformations.py -------------------------------------------------------------------
    tran = types.SimpleNamespace()

    def rotationM(rx, ry, rz, inverse=False):                      # transformations.py:234-248
        if inverse is True:
            rx, ry, rz = -rx, -ry, -rz
        r1, r2, r3 = _rot(-rx, [1, 0, 0]), _rot(-ry, [0, 1, 0]), _rot(-rz, [0, 0, 1])
        if inverse is True:
            return np.dot(r1, np.dot(r2, r3))
        return np.dot(r3, np.dot(r2, r1))

    def translationM(tx, ty, tz):                                  # transformations.py:250-255
        M = np.identity(4)
        M[:3, 3] = [-tx, -ty, -tz]
        return M

    def newCoords():                                               # transformations.py:228-232
        return [np.identity(4)] * 4

    def _masked(fn, cols, ind, *args):
        """The reference's `ind` idiom (transformations.py:20-27, surfaces.py:17-24): gather the
        selected rays, run the native routine on the copies, scatter back."""
        t = [np.ascontiguousarray(c[ind]) for c in cols]
        fn(*t, *args)
        for c, v in zip(cols, t):
            c[ind] = v

    def transform(rays, dx, dy, dz, rx, ry, rz, ind=None, coords=None):    # transformations.py:11-44
        args = [-float(v) for v in (dx, dy, dz, rx, ry, rz)]
        if ind is not None:
            _masked(be.transformationsf.transform, nine(rays), ind, *args)
        else:
            be.transformationsf.transform(*nine(rays), *args)
        if coords is not None:
            rotm, tranm = rotationM(rx, ry, rz), translationM(dx, dy, dz)
            rotmi, tranmi = rotationM(rx, ry, rz, inverse=True), translationM(-dx, -dy, -dz)
            coords[0] = np.dot(rotm, coords[0])
            coords[1] = np.dot(np.dot(rotm, tranm), coords[1])
            coords[2] = np.dot(coords[2], rotmi)
            coords[3] = np.dot(coords[3], np.dot(tranmi, rotmi))

    def itransform(rays, dx, dy, dz, rx, ry, rz, coords=None, ind=None):   # transformations.py:47-77
        args = [-float(v) for v in (dx, dy, dz, rx, ry, rz)]
        if ind is not None:
            _masked(be.transformationsf.itransform, nine(rays), ind, *args)
        else:
            be.transformationsf.itransform(*nine(rays), *args)
        if coords is not None:
            rotm, tranm = rotationM(rx, ry, rz, inverse=True), translationM(-dx, -dy, -dz)
            rotmi, tranmi = rotationM(rx, ry, rz), translationM(dx, dy, dz)
            coords[0] = np.dot(rotm, coords[0])
            coords[1] = np.dot(np.dot(tranm, rotm), coords[1])
            coords[2] = np.dot(coords[2], rotmi)
            coords[3] = np.dot(coords[3], np.dot(rotmi, tranmi))

    def steerY(rays, coords=None):                                 # transformations.py:79-83
        while np.abs(np.mean(rays[5])) > 1e-6:
            transform(rays, 0, 0, 0, -np.mean(rays[5]), 0, 0, coords=coords)

    def steerX(rays, coords=None):                                 # transformations.py:85-89
        while np.abs(np.mean(rays[4])) > 1e-6:
            transform(rays, 0, 0, 0, 0, -np.mean(rays[4]), 0, coords=coords)

    def reflect(rays, ind=None):                                   # transformations.py:102-112
        if ind is not None:
            _masked(be.transformationsf.reflect, rays[4:], ind)
        else:
            be.transformationsf.reflect(*rays[4:])

    def radgrat(rays, dpermm, order, wave, ind=None):              # transformations.py:124-172
        x, y, z, l, m, n = rays[1:7]
        fn = be.transformationsf.radgratw if type(wave) == np.ndarray else be.transformationsf.radgrat
        if ind is not None:
            tw = wave if np.size(wave) == 1 else wave[ind]
            t = [np.ascontiguousarray(c[ind]) for c in (x, y, l, m, n)]
            fn(*t, tw, dpermm, order)
            for c, v in zip((x, y, l, m, n), t):
                c[ind] = v
        else:
            fn(x, y, l, m, n, wave, dpermm, order)

    def vignette(rays, ind=None):                                  # transformations.py:214-225
        if ind is None:
            mag = rays[4] ** 2 + rays[5] ** 2 + rays[6] ** 2
            ind = np.where(mag > .1)
        return [np.ascontiguousarray(rays[i][ind]) for i in range(10)]

    def applyTPos(x, y, z, coords, inverse=False):                 # transformations.py:282-289
        i = 2 if inverse is True else 0
        pos = [np.ravel(x).astype(float), np.ravel(y).astype(float), np.ravel(z).astype(float), np.ones(np.size(x))]
        return np.dot(coords[i + 1], pos)[:3]

    def copy_rays(rays):
        return [np.copy(r) for r in rays]

    def applyT(rays, coords, inverse=False):                       # transformations.py:257-280
        i = 2 if inverse is True else 0
        on = np.ones(np.shape(rays)[1])
        pos = np.dot(coords[i + 1], [rays[1], rays[2], rays[3], on])[:3]
        wave = np.dot(coords[i], [rays[4], rays[5], rays[6], on])[:3]
        norm = np.dot(coords[i], [rays[7], rays[8], rays[9], on])[:3]
        return [rays[0], pos[0], pos[1], pos[2], wave[0], wave[1], wave[2], norm[0], norm[1], norm[2]]

    for f in (rotationM, translationM, newCoords, transform, itransform, steerX, steerY, reflect, radgrat, vignette,
              applyTPos, copy_rays, applyT):
        setattr(tran, f.__name__, f)

    # ---- analyses.py --------------------------------------------------------------------------
    anal = types.SimpleNamespace()

    def centroid(rays, weights=None):                              # analyses.py:16-22
        return np.average(rays[1], weights=weights), np.average(rays[2], weights=weights)

    def rmsCentroid(rays, weights=None):                           # analyses.py:24-30
        cx, cy = centroid(rays, weights=weights)
        return np.sqrt(np.average((rays[1] - cx) ** 2 + (rays[2] - cy) ** 2, weights=weights))

    def rmsX(rays, weights=None):                                  # analyses.py:46-51
        cx = np.average(rays[1], weights=weights)
        return np.sqrt(np.average((rays[1] - cx) ** 2, weights=weights))

    def rmsY(rays, weights=None):                                  # analyses.py:53-58
        cy = np.average(rays[2], weights=weights)
        return np.sqrt(np.average((rays[2] - cy) ** 2, weights=weights))

    def rho(rays, weights=None, cent=False):                       # analyses.py:60-71
        cx, cy = centroid(rays, weights=weights) if cent is True else (0, 0)
        return np.sqrt((rays[1] - cx) ** 2 + (rays[2] - cy) ** 2)

    def rhocdf(rays, weights=None, cent=True):                     # analyses.py:73-86
        r = rho(rays, weights=weights, cent=cent)
        if weights is None:
            weights = np.repeat(1, len(r))
        ind = np.argsort(r)
        cdf = np.cumsum(weights[ind])
        return r[ind], cdf / cdf.max()

    def hpd(rays, weights=None, cent=True):                        # analyses.py:88-97
        if weights is not None:
            r, cdf = rhocdf(rays, weights=weights, cent=True)
            return r[np.argmin(np.abs(cdf - .75))] - r[np.argmin(np.abs(cdf - .25))]
        return np.median(rho(rays, weights=weights, cent=True)) * 2.

    def hpdY(rays, weights=None):                                  # analyses.py:99-116
        y = rays[2]
        r = np.abs(y - np.average(y, weights=weights))
        if weights is None:
            return np.median(r) * 2.
        ind = np.argsort(r)
        cdf = np.cumsum(weights[ind])
        cdf = cdf / cdf.max()
        r = r[ind]
        return r[np.argmin(np.abs(cdf - .75))] - r[np.argmin(np.abs(cdf - .25))]

    def _ab(p, c, n, weights):
        """b = <p c/n> - <p><c/n>, a = <(c/n)^2> - <c/n>^2 with the reference's operation order
        (`x*l/n` is (x*l)/n, analyses.py:123-130)."""
        b = np.average(p * c / n, weights=weights) - np.average(p, weights=weights) * np.average(c / n, weights=weights)
        a = np.average((c / n) ** 2, weights=weights) - np.average(c / n, weights=weights) ** 2
        return b, a

    def analyticImagePlane(rays, weights=None):                    # analyses.py:118-133
        x, y, z, l, m, n = rays[1:7]
        bx, ax = _ab(x, l, n, weights)
        by, ay = _ab(y, m, n, weights)
        return -(bx + by) / (ax + ay)

    def analyticYPlane(rays, weights=None):                        # analyses.py:135-144
        by, ay = _ab(rays[2], rays[5], rays[6], weights)
        return -by / ay

    def analyticXPlane(rays, weights=None):                        # analyses.py:146-155
        bx, ax = _ab(rays[1], rays[4], rays[6], weights)
        return -bx / ax

    def indAngle(rays, ind=None, normal=None):                     # analyses.py:164-182
        sel = (lambda c: c) if ind is None else (lambda c: c[ind])
        l, m, n = sel(rays[4]), sel(rays[5]), sel(rays[6])
        if normal is None:
            return np.arccos(l * sel(rays[7]) + m * sel(rays[8]) + n * sel(rays[9]))
        return np.arccos(np.dot(np.array([normal]), np.vstack((l, m, n))))[0]

    def grazeAngle(rays, ind=None):                                # analyses.py:184-187
        return np.pi / 2 - indAngle(rays, ind=ind)

    def measureOPD(rays, point):                                   # analyses.py:231-244
        p = point[1:4] if len(point) == 10 else point[:3]
        return np.sqrt((rays[1] - p[0]) ** 2 + (rays[2] - p[1]) ** 2 + (rays[3] - p[2]) ** 2)

    def rmsPoint(rays, point, weights=None):                       # analyses.py:32-44
        p = point[1:4] if np.size(point) == 10 else point[:3]       # sums the squares directly (no sqrt round trip)
        rho2 = (rays[1] - p[0]) ** 2 + (rays[2] - p[1]) ** 2 + (rays[3] - p[2]) ** 2
        return np.sqrt(np.average(rho2, weights=weights))

    for f in (centroid, rmsCentroid, rmsX, rmsY, rho, rhocdf, hpd, hpdY, analyticImagePlane, analyticYPlane,
              analyticXPlane, indAngle, grazeAngle, measureOPD, rmsPoint):
        setattr(anal, f.__name__, f)

    # ---- surfaces.py --------------------------------------------------------------------------
    surf = types.SimpleNamespace()

    def flat(rays, ind=None, nr=None):                             # surfaces.py:14-29
        opd, x, y, z, l, m, n, ux, uy, uz = rays
        if ind is not None:
            _masked(be.surfacesf.flat, nine(rays), ind)
        elif nr is not None:
            be.surfacesf.flatopd(x, y, z, l, m, n, ux, uy, uz, opd, nr)
        else:
            be.surfacesf.flat(x, y, z, l, m, n, ux, uy, uz)

    def conic(rays, R, K, nr=None):                                # surfaces.py:104-113
        if nr is not None:
            be.surfacesf.conicopd(*rays, R, K, nr)
        else:
            be.surfacesf.conic(*nine(rays), R, K)

    def wolterprimary(rays, r0, z0, psi=1., nr=None):              # surfaces.py:219-227
        if nr is None:
            be.woltsurf.wolterprimary(*nine(rays), r0, z0, psi)
        else:
            be.woltsurf.wolterprimaryopd(*rays, r0, z0, psi, nr)

    def woltersecondary(rays, r0, z0, psi=1.):                     # surfaces.py:238-243
        be.woltsurf.woltersecondary(*nine(rays), r0, z0, psi)

    def spoCone(rays, R0, tg, ind=None):                           # surfaces.py:403-419
        if ind is not None:
            _masked(be.woltsurf.spocone, nine(rays), ind, R0, tg)
        else:
            be.woltsurf.spocone(*nine(rays), R0, tg)

    def spoPrimary(rays, R0, F, d=.605, ind=None):                 # surfaces.py:421-430
        spoCone(rays, R0, .25 * np.arctan((R0 + d / 2) / F), ind=ind)

    def spoSecondary(rays, R0, F, d=.605, ind=None):               # surfaces.py:432-441
        spoCone(rays, R0, .75 * np.arctan((R0 + d / 2) / F), ind=ind)

    def zernsurf(rays, coeff, rad, rorder=None, aorder=None, nr=None):     # surfaces.py:31-47 (explicit orders)
        ro, ao = np.asarray(rorder, dtype=np.int32), np.asarray(aorder, dtype=np.int32)
        if nr is not None:
            be.zernsurf.tracezernopd(*rays, np.asarray(coeff, dtype=float), ro, ao, rad, nr)
        else:
            be.zernsurf.tracezern(*nine(rays), np.asarray(coeff, dtype=float), ro, ao, rad)

    def focus(rays, fn, weights=None, nr=None, coords=None):       # surfaces.py:502-510
        dz1 = fn(rays, weights=weights)
        transform(rays, 0, 0, dz1, 0, 0, 0, coords=coords)
        flat(rays, nr=nr)
        dz2 = fn(rays, weights=weights)
        transform(rays, 0, 0, dz2, 0, 0, 0, coords=coords)
        flat(rays, nr=nr)
        return dz1 + dz2

    def focusY(rays, weights=None, nr=None, coords=None):
        return focus(rays, analyticYPlane, weights=weights, nr=nr, coords=coords)

    def focusX(rays, weights=None, nr=None, coords=None):
        return focus(rays, analyticXPlane, weights=weights, nr=nr, coords=coords)

    def focusI(rays, weights=None, nr=None, coords=None):
        return focus(rays, analyticImagePlane, weights=weights, nr=nr, coords=coords)

    for f in (flat, conic, wolterprimary, woltersecondary, spoCone, spoPrimary, spoSecondary, zernsurf, focus, focusY,
              focusX, focusI):
        setattr(surf, f.__name__, f)

    # ---- sources.py (global np.random stream, rho first then theta) ------------------------------
    sources = types.SimpleNamespace()

    def _bundle(x, y, l, m, n):
        num = len(x)
        z = lambda: np.repeat(0., num)
        return [z(), x, y, z(), l, m, n, z(), z(), z()]

    def annulus(rin, rout, num, zhat=-1.):                         # sources.py:91-127
        rho_ = np.sqrt(rin ** 2 + np.random.rand(num) * (rout ** 2 - rin ** 2))
        theta = np.random.rand(num) * 2 * np.pi
        return _bundle(rho_ * np.cos(theta), rho_ * np.sin(theta), np.repeat(0., num), np.repeat(0., num),
                       np.repeat(zhat, num))

    def subannulus(rin, rout, dphi, num, zhat=1.):                 # sources.py:130-170
        rho_ = np.sqrt(rin ** 2 + np.random.rand(num) * (rout ** 2 - rin ** 2))
        theta = np.random.rand(num) * dphi - dphi / 2.
        return _bundle(rho_ * np.cos(theta), rho_ * np.sin(theta), np.repeat(0., num), np.repeat(0., num),
                       np.repeat(zhat, num))

    def circularbeam(rad, num):                                    # sources.py:56-88
        rho_ = np.sqrt(np.random.rand(num)) * rad
        theta = np.random.rand(num) * 2 * np.pi
        return _bundle(rho_ * np.cos(theta), rho_ * np.sin(theta), np.repeat(0., num), np.repeat(0., num),
                       np.repeat(1., num))

    def pointsource(ang, num):                                     # sources.py:20-53
        rho_ = np.sqrt(np.random.rand(num)) * np.sin(ang)
        theta = np.random.rand(num) * 2 * np.pi
        l, m = rho_ * np.cos(theta), rho_ * np.sin(theta)
        return _bundle(np.repeat(0., num), np.repeat(0., num), l, m, np.sqrt(1. - l ** 2 - m ** 2))

    for f in (annulus, subannulus, circularbeam, pointsource):
        setattr(sources, f.__name__, f)
    return sources, surf, tran, anal


def conic_namespace():
    """conicsolve.primrad / secrad / woltparam / primfocus (conicsolve.py:7-15, 29-37, 51-64), NumPy as there."""
    ns = types.SimpleNamespace()

    def _pde(r0, z0, psi):
        alpha = .25 * np.arctan(r0 / z0)
        thetah = 2 * (1 + 2 * psi) / (1 + psi) * alpha
        thetap = 2 * psi / (1 + psi) * alpha
        p = z0 * np.tan(4 * alpha) * np.tan(thetap)
        d = z0 * np.tan(4 * alpha) * np.tan(4 * alpha - thetah)
        e = np.cos(4 * alpha) * (1 + np.tan(4 * alpha) * np.tan(thetah))
        return p, d, e

    def primrad(z, r0, z0, psi=1.):
        p, d, e = _pde(r0, z0, psi)
        return np.sqrt(p ** 2 + 2 * p * z + (4 * e ** 2 * p * d) / (e ** 2 - 1))

    def secrad(z, r0, z0, psi=1.):
        p, d, e = _pde(r0, z0, psi)
        return np.sqrt(e ** 2 * (d + z) ** 2 - z ** 2)

    def woltparam(r0, z0):
        alpha = .25 * np.arctan(r0 / z0)
        thetah, thetap = 3 * alpha, alpha
        p = z0 * np.tan(4 * alpha) * np.tan(thetap)
        d = z0 * np.tan(4 * alpha) * np.tan(4 * alpha - thetah)
        e = np.cos(4 * alpha) * (1 + np.tan(4 * alpha) * np.tan(thetah))
        return alpha, p, d, e

    def primfocus(r0, z0):
        alpha, p, d, e = woltparam(r0, z0)
        return z0 + 2 * e ** 2 * d / (e ** 2 - 1)

    ns.primrad, ns.secrad, ns.woltparam, ns.primfocus = primrad, secrad, woltparam, primfocus
    return ns
