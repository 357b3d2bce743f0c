"""Torus surface-function helpers of ``PyXFocus/torus.py`` (host NumPy formulas; the torus TRACE is
``surfaces.torus`` on the device)."""
import numpy as np


def torusF(x, y, z, rin, rout):
    """Torus surface function centred on the torus (torus.py:4-11; including the reference's
    ``y**z`` term, which is what it evaluates)."""
    return (x ** 2 + y ** z + z ** 2 - rin ** 2 - rout ** 2) ** 2 + 4 * rout ** 2 * (z ** 2 - rin ** 2)


def torusF2(x, y, z, rin, rout):
    """Torus surface function with the origin tangent to the outer surface (torus.py:13-18)."""
    s = z + rin + rout
    return (s ** 2 + y ** 2 + x ** 2 + rout ** 2 - rin ** 2) ** 2 - 4 * rout ** 2 * (s ** 2 + y ** 2)


def torusGrad(x, y, z, rin, rout):
    """Derivatives of the tangent-origin torus function (torus.py:20-26)."""
    common = 2 * rin * (rout + z) + 2 * rout * z + x ** 2 + y ** 2 + z ** 2
    dfdz = 4 * (rin + rout + z) * common
    dfdy = 4 * y * common
    dfdx = 4 * x * (-rin ** 2 + (rin + rout + z) ** 2 + rout ** 2 + x ** 2 + y ** 2)
    return dfdx, dfdy, dfdz


def constructQuartic(p, d, rin, rout):
    """Coefficients [a4..a0] of the line-torus intersection quartic for position p, direction d
    (torus.py:28-43)."""
    alpha = np.dot(d, d)
    beta = 2 * np.dot(p, d)
    gamma = np.dot(p, p) - rin ** 2 - rout ** 2
    a4 = alpha ** 2
    a3 = 2 * alpha * beta
    a2 = beta ** 2 + 2 * alpha * gamma + 4 * rout ** 2 * d[2] ** 2
    a1 = 2 * beta * gamma + 8 * rout ** 2 * p[2] * d[2]
    a0 = gamma ** 2 + 4 * rout ** 2 * p[2] ** 2 - 4 * rout ** 2 * rin ** 2
    return np.array([a4, a3, a2, a1, a0])
