"""Retro-reflection test of a Wolter primary: `tracePrimary` of the reference's examples/axro/retraceError.py:20-65.

A point source at the primary's focus illuminates a sub-aperture of the primary (`pointTo(..., reverse=1.)`),
the collimated beam is sent back by a flat retro-reflector 400 mm above the mirror, passes the primary a second
time and comes to the primary's focus again; the mirror may carry a six-degree-of-freedom misalignment
(`transform(*primalign)` ... `itransform(*primalign)`) and a Legendre-Legendre figure error (`primaryLL`).  Returns the
focal bundle and the launch azimuth of every ray.

Sequence of PyXFocus calls and element-wise operations as in the reference -- INCLUDING its quirk: the second
pass computes the mask of rays inside the mirror's axial extent and calls `tran.vignette(rays, ind=ind)` WITHOUT
keeping the result, so nothing is vignetted (retraceError.py:55-56); the call is made, and dropped, here too.  The
coefficient file the reference reads from a private Dropbox path is an argument.  `api` namespace as in
examples/arcus_sector.py (the fixture tests/golden/retrace.npz was produced by this file on the reference's own
modules).  Between the points where the script needs host values the calls run as fused programs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from arcus_sector import Api  # noqa: E402,F401

R0, Z0, ZMAX = 220., 8400., 8500.
DPHI = 100. / 220


def _primary(api, rays, primCoeffs):
    if primCoeffs is None:
        api.surf.wolterprimary(rays, R0, Z0)
    else:
        api.surf.primaryLL(rays, R0, Z0, ZMAX, Z0, DPHI, *primCoeffs)


def tracePrimary(api, primCoeffs=None, primalign=np.zeros(6), num=100000):
    """retraceError.py:20-65.  primCoeffs = (coeff, axial orders, azimuthal orders) or None for the perfect
    Wolter-I primary."""
    xp = api.xp
    primfoc = api.conic.primfocus(R0, Z0)
    r1 = api.conic.primrad(ZMAX, R0, Z0)
    rays = api.sources.subannulus(R0, r1, DPHI, int(num), zhat=1.)
    api.tran.pointTo(rays, 0., 0., -primfoc, reverse=1.)
    theta = xp.arctan2(rays[2], rays[1])

    with api.fused():                                  # source -> primary -> retro-reflector -> primary again
        api.tran.transform(rays, *primalign)
        api.tran.transform(rays, 0., 0, -Z0, 0, 0, 0)
        _primary(api, rays, primCoeffs)
        api.tran.transform(rays, 0, 0, Z0, 0, 0, 0)
        api.tran.itransform(rays, *primalign)
        api.tran.reflect(rays)

        api.tran.transform(rays, 0, 0, 400., 0, 0, 0)
        api.surf.flat(rays)
        api.tran.reflect(rays)
        api.tran.transform(rays, 0, 0, -400., 0, 0, 0)

    with api.fused():
        api.tran.transform(rays, *primalign)
        api.tran.transform(rays, 0., 0, -Z0, 0, 0, 0)
        _primary(api, rays, primCoeffs)
    ind = xp.logical_and(rays[3] > Z0, rays[3] < ZMAX)
    api.tran.vignette(rays, ind=ind)                   # (result dropped, as in the reference)
    with api.fused():
        api.tran.transform(rays, 0, 0, Z0, 0, 0, 0)
        api.tran.itransform(rays, *primalign)
        api.tran.reflect(rays)
        api.tran.transform(rays, 0, 0, -primfoc, 0, 0, 0)
        api.surf.flat(rays)
    return rays, theta


if __name__ == "__main__":
    api = Api.amd()
    co = (np.array([2e-4, -1e-4, 5e-5]), np.array([1, 2, 0]), np.array([0, 0, 2]))
    for tag, kw in (("perfect", {}), ("tilted 20 urad", dict(primalign=np.array([0, 0, 0, 2e-5, 0, 0.]))),
                    ("figure error", dict(primCoeffs=co))):
        np.random.seed(1)
        rays, theta = tracePrimary(api, num=10 ** 5, **kw)
        cx, cy = api.anal.centroid(rays)
        print("%-16s centroid (%.5f, %.5f) mm, rms %.3e mm" % (tag, cx, cy, api.anal.rmsCentroid(rays)))
