"""Singlet / doublet / cylinder lens recipes of PyXFocus ``lenses.py`` on the HIP kernels.

Composites of sphere / cyl / flat / cylconic + refract + frame moves, with the reference's
conventions (lenses.py:7-91, 161-204, 231-246): reference frame +z toward the optical axis, xy
plane tangent to the first surface, positive R = convex, OPD accumulated with the local index.

Not ported: ``lens`` / ``cyllens`` and the catalogue entries built on them (``LJ1653L2``,
``LJ1629L2``, ``AC254_400_A``, ``AC508_200_A``): in the reference they call undefined bare names
(lenses.py:101-157 use ``transform``/``sphere``/``vignette`` without a module, ``cyllens`` has a
NameError on its first line) and discard the result of ``vignette`` -- there is no behaviour to match.
"""
import numpy as np

from . import surfaces as surf
from . import transformations as tran


def singlet(rays, r1, r2, thick, nl, reverse=False):
    """Spherical singlet lens (lenses.py:7-28)."""
    if reverse is True:
        r1, r2 = r2, r1
    tran.transform(rays, 0, 0, r1, 0, 0, 0)
    surf.sphere(rays, r1, nr=1.)
    tran.refract(rays, 1., nl)
    tran.transform(rays, 0, 0, -r1 + thick - r2, 0, 0, 0)
    surf.sphere(rays, r2, nr=nl)
    tran.transform(rays, 0, 0, r2, 0, 0, 0)
    tran.refract(rays, nl, 1.)
    surf.flat(rays, nr=1.)
    return


def singletCyl(rays, r1, r2, thick, nl, reverse=False):
    """Cylindrical singlet lens, cylinder axis along y, radius 0 = flat (lenses.py:30-60)."""
    if reverse is True:
        r1, r2 = r2, r1
    tran.transform(rays, 0, 0, r1, 0, 0, 0)
    if r1 == 0:
        surf.flat(rays, nr=1.)
    else:
        surf.cyl(rays, r1, nr=1.)
    tran.refract(rays, 1., nl)
    tran.transform(rays, 0, 0, -r1 + thick - r2, 0, 0, 0)
    if r2 == 0:
        surf.flat(rays, nr=1.)
    else:
        surf.cyl(rays, r2, nr=nl)
    tran.transform(rays, 0, 0, r2, 0, 0, 0)
    tran.refract(rays, nl, 1.)
    surf.flat(rays, nr=1.)
    return


def doublet(rays, r1, r2, r3, n1, n2, t1, t2, reverse=False):
    """Cemented doublet, similar to Thorlabs achromats (lenses.py:62-91)."""
    if reverse is True:
        r1, r3, r2 = r3, r1, -r2
        n1, n2 = n2, n1
        t1, t2 = t2, t1
    tran.transform(rays, 0, 0, r1, 0, 0, 0)
    surf.sphere(rays, r1, nr=1.)
    tran.refract(rays, 1., n1)
    tran.transform(rays, 0, 0, -r1 + t1 + r2, 0, 0, 0)
    surf.sphere(rays, r2, nr=n1)
    tran.refract(rays, n1, n2)
    tran.transform(rays, 0, 0, -r2 + t2 - r3, 0, 0, 0)
    surf.sphere(rays, r3, nr=n2)
    tran.transform(rays, 0, 0, r3, 0, 0, 0)
    tran.refract(rays, n2, 1.)
    surf.flat(rays, nr=1.)
    return


# ---- catalogue lenses (lenses.py:161-204) --------------------------------------------------------
def AC508_250(rays, reverse=False):
    """Thorlabs AC508-250, highly curved surface first."""
    doublet(rays, 137.1, -111.7, 459.2, 1.5150885, 1.6437928, 7.5, 2., reverse=reverse)
    return


def collimator6(rays, reverse=False):
    """Six inch collimator from Cumberland (R1=1124, R2=9324); reverse focuses a plane wave."""
    singlet(rays, 9324., 1124., 20., 1.5150885, reverse=reverse)
    return


def edmundCollimator(rays, reverse=False):
    """5" collimator from Edmund; reverse focuses a plane wave."""
    if reverse is True:
        singlet(rays, 1131.72, 780.87, 15.42, 1.51501)
        tran.transform(rays, 0, 0, .1, 0, 0, 0)
        singlet(rays, -779.37, 2704.01, 10.92, 1.64363)
    else:
        singlet(rays, 2704.01, -779.37, 10.92, 1.64363)
        tran.transform(rays, 0, 0, .1, 0, 0, 0)
        singlet(rays, 780.87, 1131.72, 15.42, 1.51501)
    return


def LJ1516_L2(rays, reverse=False):
    singletCyl(rays, 0, 516.8, 3.2, 1.5150885, reverse=reverse)
    return


def LJ1144_L2(rays, reverse=False):
    singletCyl(rays, 0, 258.4, 3.4, 1.5150885, reverse=reverse)
    return


def cylNull(rays, reverse=False):
    """Cylindrical null lens (lenses.py:231-246)."""
    if reverse is False:
        tran.transform(rays, 0, 0, 0, np.pi / 2, 0, 0)
        surf.cylconic(rays, .007626, -.575)
        tran.refract(rays, 1., 1.51501)
        tran.transform(rays, 0, 0, 0, -np.pi / 2, 0, 0)
        tran.transform(rays, 0, 0, 50, 0, 0, 0)
        surf.flat(rays)
        tran.refract(rays, 1.51501, 1.)
    else:
        tran.refract(rays, 1., 1.51501)
        tran.transform(rays, 0, 0, 50, 0, 0, 0)
        tran.transform(rays, 0, 0, 0, np.pi / 2, 0, 0)
        surf.cylconic(rays, -.007626, -.575)
        tran.refract(rays, 1.51501, 1.)
        tran.transform(rays, 0, 0, 0, -np.pi / 2, 0, 0)
    return
