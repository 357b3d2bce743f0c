"""Blaze / Littrow design helpers of PyXFocus ``grating.py`` (grating.py:5-38): scalar (or array)
formulas on the host.  (``transformations.radgrat`` / ``grat`` are the per-ray grating kernels.)

Everything derives from one quantity, the order term q = m*wave/d: the Littrow yaw is
asin(q / (2 cos inc)), and the diffracted direction cosines follow from the yawed geometry."""
import numpy as np


def _littrow(inc, wave, m, d):
    """(yaw, beta, alpha): Littrow yaw and the direction cosines along / across the grooves."""
    c = np.cos(inc)
    yaw_ = np.arcsin(m * wave / 2 / d / c)
    return yaw_, c * np.cos(yaw_), c * np.sin(yaw_) - m * wave / d


def blazeYaw(inc, wave, m, d):
    """Yaw that puts wavelength `wave` of order m (groove period d) at blaze for incidence `inc`."""
    return _littrow(inc, wave, m, d)[0]


def litBetaAlpha(inc, wave, m, d):
    """(beta, alpha) direction cosines at the Littrow condition."""
    return _littrow(inc, wave, m, d)[1:]


def blazeAngle(inc, wave, m, d):
    """Blaze angle for wavelength `wave` at order m, incidence `inc`, groove period d."""
    _, beta, alpha = _littrow(inc, wave, m, d)
    return np.arcsin(alpha / np.cos(np.arcsin(beta)))


def blazeAngle2(inc, wave, m, d):
    """The same from the closed form asin(a / sqrt(sin^2(inc) + a^2)), a = m wave / 2d."""
    a = m * wave / 2 / d
    return np.arcsin(a / np.sqrt(1 - np.cos(inc) ** 2 + a ** 2))


def eta(phi0, theta0):
    """Graze angle from the polar pair (phi0, theta0)."""
    return np.arcsin(np.cos(phi0) * np.cos(theta0))


def yaw(phi0, theta0):
    """Yaw from the polar pair (phi0, theta0)."""
    return np.arctan(np.sin(theta0) / np.tan(phi0))
