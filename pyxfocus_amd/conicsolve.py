"""Closed-form Wolter-I geometry helpers (the four functions of PyXFocus ``conicsolve.py`` the
hot path needs: conicsolve.py:7-15, 29-37, 51-64).  Host scalar math; NumPy arrays accepted for z."""
import numpy as np


def _vsp(r0, z0, psi):
    alpha = .25 * np.arctan(r0 / z0)
    thetah = 2 * (1 + 2 * psi) / (1 + psi) * alpha
    thetap = 2 * psi / (1 + psi) * alpha
    p = z0 * np.tan(4 * alpha) * np.tan(thetap)
    d = z0 * np.tan(4 * alpha) * np.tan(4 * alpha - thetah)
    e = np.cos(4 * alpha) * (1 + np.tan(4 * alpha) * np.tan(thetah))
    return alpha, p, d, e


def primrad(z, r0, z0, psi=1.):
    """Radius of the Wolter-I primary at axial position z."""
    alpha, p, d, e = _vsp(r0, z0, psi)
    return np.sqrt(p ** 2 + 2 * p * z + (4 * e ** 2 * p * d) / (e ** 2 - 1))


def secrad(z, r0, z0, psi=1.):
    """Radius of the Wolter-I secondary at axial position z."""
    alpha, p, d, e = _vsp(r0, z0, psi)
    return np.sqrt(e ** 2 * (d + z) ** 2 - z ** 2)


def woltparam(r0, z0):
    """(alpha, p, d, e) for psi = 1."""
    alpha = .25 * np.arctan(r0 / z0)
    thetah = 3 * alpha
    thetap = alpha
    p = z0 * np.tan(4 * alpha) * np.tan(thetap)
    d = z0 * np.tan(4 * alpha) * np.tan(4 * alpha - thetah)
    e = np.cos(4 * alpha) * (1 + np.tan(4 * alpha) * np.tan(thetah))
    return (alpha, p, d, e)


def primfocus(r0, z0):
    """Distance to the primary focus."""
    alpha, p, d, e = woltparam(r0, z0)
    return z0 + 2 * e ** 2 * d / (e ** 2 - 1)


def ellipsoidFunction(S, psi, R, F):
    """Ellipsoid-hyperboloid telescope: primary focal length P and ellipsoid a, b, e, f
    (conicsolve.py:263-281)."""
    P = R / np.sin((psi * np.arcsin(R / F) - np.arcsin(R / S)) / (1 + psi))
    f = (S + P) / 2.
    a = 1.
    b = -(R ** 2 + (f - P) ** 2 + f ** 2)
    c = f ** 2 * (f - P) ** 2
    a = np.sqrt((-b + np.sqrt(b ** 2 - 4 * a * c)) / (2 * a))
    b = np.sqrt(a ** 2 - f ** 2)
    e = f / a
    return P, a, b, e, f


# ---- remaining design helpers (host scalar/array math, conicsolve.py:17-27,39-49,83-98,250-324).
# The reference's wsPrimFunction*/wsSecFunction*/mathraytrace stop in pdb.set_trace() as shipped and
# have no callable behaviour; the W-S surface functions themselves run on the device
# (surfaces.wsPrimary / wsSecondary).
def _sag(z, r, half):
    fit = np.polyfit(z, r, 2)
    return np.abs(fit[0] * half ** 2)


def primsag(z1, r0, z0):
    """Sag of a primary of length z1-z0 starting at the node (conicsolve.py:17-26)."""
    z = np.linspace(z0, z1, 100)
    return _sag(z, primrad(z, r0, z0), (z1 - z0) / 2.)


def secsag(z1, z0, r0, F, psi=1.):
    """Sag of a secondary between z0 and z1 for the prescription (r0, F) (conicsolve.py:39-48)."""
    z = np.linspace(z0, z1, 100)
    return _sag(z, secrad(z, r0, F, psi=psi), (z1 - z0) / 2.)


def rGoal_to_rMax(rgoal, z0, zmax):
    """Node radius r0 whose primary reaches radius rgoal at zmax: search of 10000 candidates in
    [rgoal-2, rgoal] (conicsolve.py:83-86)."""
    rguess = np.linspace(rgoal - 2., rgoal, 10000)
    rnext = primrad(zmax, rguess, z0)
    return rguess[np.argmin(np.abs(rgoal - rnext))]


def primaryintercept(rmax, rmin, z0, zmin, zmax):
    """Nested set of primary prescriptions intercepting the beam from rmax down to rmin
    (conicsolve.py:89-96).  The reference only prints the radii; they are also returned here."""
    out = []
    rnew = rGoal_to_rMax(rmax, z0, zmax)
    print(rnew)
    out.append(rnew)
    while rnew > rmin:
        rmax = primrad(zmin, rnew, z0)
        rnew = rGoal_to_rMax(rmax, z0, zmax)
        print(rnew)
        out.append(rnew)
    return out


def wsRMS(psi, theta, alpha, L1, z0):
    """RMS blur at the optimum focal surface, Chase & Van Speybroeck (conicsolve.py:250-254)."""
    return .135 * (psi + 1) * (np.tan(theta) ** 2 / np.tan(alpha)) * L1 / z0


def wsFoc(r, psi, L1, z0, alpha):
    """Optimum focal-surface height at radius r, Chase & Van Speybroeck (conicsolve.py:257-261)."""
    return .0625 * (psi + 1) * (r ** 2 * L1 / z0 ** 2) / np.tan(alpha) ** 2


def ellipsoidRad(S, psi, R, F, z):
    """Radius of an ellipsoid primary at height z above the two-mirror focus (conicsolve.py:283-290)."""
    P, a, b, e, f = ellipsoidFunction(S, psi, R, F)
    zfoc = f - P + F
    return np.sqrt(1 - (z - zfoc) ** 2 / a ** 2) * b


def ehSecRad(S, psi, R, F, z):
    """Radius of the ellipsoid-hyperboloid secondary at height z (conicsolve.py:292-300)."""
    P, a, b, e, f = ellipsoidFunction(S, psi, R, F)
    psi_eff = np.arctan(R / P) / (np.arctan(R / F) - np.arctan(R / P))
    return secrad(z, R, F, psi=psi_eff)


def ellipsoidSag(S, psi, R0, F, z1, z0):
    """Sag of an ellipsoid primary between z0 and z1 (conicsolve.py:302-309)."""
    z = np.linspace(z0, z1, 100)
    return _sag(z, ellipsoidRad(S, psi, R0, F, z), (z1 - z0) / 2.)


def solveS(P, a, b, e, f, x, y, z, l, m, n):
    """Analytic conic intersection for one ray: (b2, c2, s1, s2) (conicsolve.py:311-324)."""
    K = -e ** 2
    R = b ** 2 / a
    denom = l ** 2 + m ** 2 + (K + 1) * n ** 2
    b2 = (l * x + m * y - R * n + (K + 1) * n * z) / denom
    c2 = (x ** 2 + y ** 2 - 2 * R * z + (K + 1) * z ** 2) / denom
    s1 = -b2 + np.sqrt(b2 ** 2 - c2)
    s2 = -b2 - np.sqrt(b2 ** 2 - c2)
    return b2, c2, s1, s2
