"""Closed-form blaze / Littrow helpers of PyXFocus ``grating.py`` (grating.py:5-38): scalar design
formulas, host NumPy.  (``transformations.radgrat`` / ``grat`` are the per-ray grating kernels.)"""
import numpy as np
from numpy import cos, sin, sqrt


def blazeYaw(inc, wave, m, d):
    """Yaw for the blaze wavelength at order m, groove period d, incidence angle inc."""
    return np.arcsin(m * wave / 2 / d / cos(inc))


def blazeAngle(inc, wave, m, d):
    """Blaze angle for wavelength wave at order m, incidence inc, groove period d."""
    psi = blazeYaw(inc, wave, m, d)
    beta1 = cos(inc) * cos(psi)
    alpha1 = cos(inc) * sin(psi) - m * wave / d
    return np.arcsin(alpha1 / cos(np.arcsin(beta1)))


def litBetaAlpha(inc, wave, m, d):
    """(beta, alpha) direction cosines at the Littrow condition."""
    psi = blazeYaw(inc, wave, m, d)
    beta1 = cos(inc) * cos(psi)
    alpha1 = cos(inc) * sin(psi) - m * wave / d
    return beta1, alpha1


def blazeAngle2(inc, wave, m, d):
    a = m * wave / 2 / d
    return np.arcsin(a / sqrt(1 - cos(inc) ** 2 + a ** 2))


def eta(phi0, theta0):
    return np.arcsin(np.cos(phi0) * np.cos(theta0))


def yaw(phi0, theta0):
    return np.arctan(np.sin(theta0) / np.tan(phi0))
