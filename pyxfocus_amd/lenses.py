"""Singlet / doublet / cylinder lens recipes of PyXFocus ``lenses.py`` on the HIP kernels.

A lens here is DATA: a list of steps (move the frame along z, intersect a sphere / cylinder /
plane with a given index for the OPD, refract) that one small runner executes -- inside
``pyxfocus_amd.fused()`` a whole lens is then a single launch.  The step lists reproduce the
reference's conventions (lenses.py:7-91, 161-204, 231-246): frame +z toward the optical axis, xy
plane tangent to the first surface, positive R = convex, OPD accumulated with the local index.

Not provided: ``lens`` / ``cyllens`` and the catalogue entries built on them (``LJ1653L2``,
``LJ1629L2``, ``AC254_400_A``, ``AC508_200_A``): in the reference they call undefined bare names
(lenses.py:101-157 use ``transform``/``sphere``/``vignette`` without a module, ``cyllens`` has a
NameError on its first line) and discard the result of ``vignette`` -- there is no behaviour to match.
"""
import math

from . import surfaces as surf
from . import transformations as tran

# step kinds
_Z, _RX, _SPH, _CYL, _FLAT, _CYLCONIC, _REFRACT = range(7)


def _run(rays, steps):
    for kind, a, b in steps:
        if kind == _Z:
            tran.transform(rays, 0, 0, a, 0, 0, 0)
        elif kind == _RX:
            tran.transform(rays, 0, 0, 0, a, 0, 0)
        elif kind == _SPH:
            surf.sphere(rays, a, nr=b)
        elif kind == _CYL:                        # radius 0 means a flat face
            surf.flat(rays, nr=1.) if a == 0 else surf.cyl(rays, a, nr=b)
        elif kind == _FLAT:
            surf.flat(rays, nr=a)
        elif kind == _CYLCONIC:
            surf.cylconic(rays, a, b)
        else:
            tran.refract(rays, a, b)


def _two_surface(face, r1, r2, thick, nl):
    """centre of curvature 1 -> face 1 -> glass -> centre of curvature 2 -> face 2 -> vertex plane -> air"""
    return [(_Z, r1, None), (face, r1, 1.), (_REFRACT, 1., nl), (_Z, -r1 + thick - r2, None), (face, r2, nl),
            (_Z, r2, None), (_REFRACT, nl, 1.), (_FLAT, 1., None)]


def singlet(rays, r1, r2, thick, nl, reverse=False):
    """Spherical singlet lens (lenses.py:7-28)."""
    if reverse is True:
        r1, r2 = r2, r1
    _run(rays, _two_surface(_SPH, r1, r2, thick, nl))


def singletCyl(rays, r1, r2, thick, nl, reverse=False):
    """Cylindrical singlet lens, cylinder axis along y, radius 0 = flat (lenses.py:30-60)."""
    if reverse is True:
        r1, r2 = r2, r1
    _run(rays, _two_surface(_CYL, r1, r2, thick, nl))


def doublet(rays, r1, r2, r3, n1, n2, t1, t2, reverse=False):
    """Cemented doublet, similar to Thorlabs achromats (lenses.py:62-91)."""
    if reverse is True:
        r1, r3, r2 = r3, r1, -r2
        n1, n2 = n2, n1
        t1, t2 = t2, t1
    _run(rays, [(_Z, r1, None), (_SPH, r1, 1.), (_REFRACT, 1., n1), (_Z, -r1 + t1 + r2, None), (_SPH, r2, n1),
                (_REFRACT, n1, n2), (_Z, -r2 + t2 - r3, None), (_SPH, r3, n2), (_Z, r3, None),
                (_REFRACT, n2, 1.), (_FLAT, 1., None)])


# ---- catalogue (lenses.py:161-204): name -> (recipe, parameters) -----------------------------------
_BK7, _SF2 = 1.5150885, 1.6437928
_CATALOGUE = {
    "AC508_250": (doublet, (137.1, -111.7, 459.2, _BK7, _SF2, 7.5, 2.), "Thorlabs AC508-250, highly curved surface first"),
    "collimator6": (singlet, (9324., 1124., 20., _BK7), "six inch collimator from Cumberland; reverse focuses a plane wave"),
    "LJ1516_L2": (singletCyl, (0, 516.8, 3.2, _BK7), "Thorlabs LJ1516L2 cylinder"),
    "LJ1144_L2": (singletCyl, (0, 258.4, 3.4, _BK7), "Thorlabs LJ1144L2 cylinder"),
}


def _catalogue_entry(name):
    recipe, par, doc = _CATALOGUE[name]

    def trace(rays, reverse=False):
        recipe(rays, *par, reverse=reverse)
    trace.__name__ = trace.__qualname__ = name
    trace.__doc__ = doc + " (lenses.py:161-204)."
    return trace


for _name in _CATALOGUE:
    globals()[_name] = _catalogue_entry(_name)
del _name


def edmundCollimator(rays, reverse=False):
    """5" two-element collimator from Edmund, 0.1 mm air gap; reverse focuses a plane wave
    (lenses.py:180-195)."""
    front, back = (2704.01, -779.37, 10.92, 1.64363), (780.87, 1131.72, 15.42, 1.51501)
    if reverse is True:                            # elements swap places and turn around
        front, back = (back[1], back[0]) + back[2:], (front[1], front[0]) + front[2:]
    singlet(rays, *front)
    tran.transform(rays, 0, 0, .1, 0, 0, 0)
    singlet(rays, *back)


def cylNull(rays, reverse=False):
    """Cylindrical null lens: a conic cylinder face and a flat face 50 mm apart in 1.51501 glass
    (lenses.py:231-246)."""
    n, quarter = 1.51501, math.pi / 2
    if reverse is False:
        steps = [(_RX, quarter, None), (_CYLCONIC, .007626, -.575), (_REFRACT, 1., n), (_RX, -quarter, None),
                 (_Z, 50, None), (_FLAT, None, None), (_REFRACT, n, 1.)]
    else:
        steps = [(_REFRACT, 1., n), (_Z, 50, None), (_RX, quarter, None), (_CYLCONIC, -.007626, -.575),
                 (_REFRACT, n, 1.), (_RX, -quarter, None)]
    _run(rays, steps)
