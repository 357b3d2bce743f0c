"""Which rays reach a Wolter-Schwarzschild mirror?  -- the experiment of the reference's
examples/wolterSchwarzschildTest.py on the device layer.

Rays are launched either along the optical axis from infinity ("normal") or perpendicular to it
("transverse"), with an optional extra direction cosine and an offset of the launch plane, at the
primary or the secondary; the trace leaves rays that never converge where they started, so comparing
positions before and after tells which rays were actually traced to the mirror.

The reference script spells this out as four functions (testNormalPrimary, testTransversePrimary,
testNormalSecondary, testTransverseSecondary); here they are four names for one routine.  The other
differences: the imports, and torch instead of NumPy where a ray column is edited in place."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import pyxfocus_amd.sources as sources  # noqa: E402
import pyxfocus_amd.surfaces as surf  # noqa: E402
import pyxfocus_amd.transformations as tran  # noqa: E402

OPD, X, Y, Z, L, M, N = range(7)


def launch(mirror, direction, r0, z0, zeta=1., tilt=0., shift=0., num=10 ** 4, rays=None):
    """mirror: "primary" | "secondary"; direction: "normal" | "transverse".
    normal:     circular beam of radius 2 r0 brought to z = z0 - shift, travelling -z, direction cosine
                l increased by `tilt`;
    transverse: z0 x r0 rectangular beam turned to travel along -x, placed at x = r0 - shift,
                direction cosine n increased by `tilt`.
    Returns (rays, a, b): the traced bundle and the two launch coordinates that span the beam."""
    trace = {"primary": surf.wsPrimary, "secondary": surf.wsSecondary}[mirror]
    if direction == "normal":
        if rays is None:
            rays = sources.circularbeam(r0 * 2, int(num))
        tran.transform(rays, 0, 0, -z0 + shift, 0, 0, 0)
        tilted, dependent, span = L, N, (X, Y)
    else:
        if rays is None:
            rays = sources.rectbeam(z0, r0, int(num))
        tran.transform(rays, 0, 0, 0, 0, np.pi / 2, 0)
        tran.transform(rays, -r0 + shift, 0, -z0, 0, 0, 0)
        tilted, dependent, span = N, L, (Y, Z)
    rays[tilted] = rays[tilted] + tilt
    rays[dependent] = -torch.sqrt(1 - rays[tilted] ** 2)
    a, b = rays[span[0]].clone(), rays[span[1]].clone()
    trace(rays, r0, z0, zeta)
    return rays, a, b


def testNormalPrimary(r0, z0, zeta=1., offx=0., dz=0., **kw):
    return launch("primary", "normal", r0, z0, zeta, offx, dz, **kw)


def testTransversePrimary(r0, z0, zeta=1., offz=0., dx=0., **kw):
    return launch("primary", "transverse", r0, z0, zeta, offz, dx, **kw)


def testNormalSecondary(r0, z0, zeta=1., offx=0., dz=0., **kw):
    return launch("secondary", "normal", r0, z0, zeta, offx, dz, **kw)


def testTransverseSecondary(r0, z0, zeta=1., offz=0., dx=0., **kw):
    return launch("secondary", "transverse", r0, z0, zeta, offz, dx, **kw)


if __name__ == "__main__":
    for off in (0., 1e-3, 5e-3):
        rays, xi, yi = testNormalPrimary(220., 8400., offx=off)
        moved = (rays[X] != xi) | (rays[Y] != yi)
        print("offx = %.0e: %d of %d rays traced to the primary" % (off, int(moved.sum()), len(xi)))
