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
