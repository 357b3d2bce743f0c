"""Wolter-Schwarzschild shell study: `wsPrimrad`, `wsSecrad`, `traceZeta` and `onaxisMerit` of the reference's
examples/axro/axialHeights.py:50-133.

`wsPrimrad / wsSecrad` find the radius of a W-S mirror at an axial height by shooting ONE ray sideways at it
(a one-ray bundle through pointsource / transform / wsPrimary); `traceZeta` traces a sub-aperture through a
W-S pair with an off-axis kick between the mirrors and reports where the rays strike the secondary, the best
focus (`focusI`) and the spot's HPD / RMS; `onaxisMerit` counts the rays that miss the secondary's extent.
Sequence of PyXFocus calls and element-wise operations as in the reference; `api` namespace as in
examples/arcus_sector.py (the fixture tests/golden/ws_heights.npz was produced by this file on the
reference's own modules)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from arcus_sector import Api  # noqa: E402,F401


def wsPrimrad(api, z, r0, z0, psi=1.):
    """axialHeights.py:50-61"""
    ray = api.sources.pointsource(0., 1)
    api.tran.transform(ray, 0, 0, 0, 0, -np.pi / 2, 0)
    api.tran.transform(ray, -r0 - 2., 0, -z, 0, 0, 0)
    api.surf.wsPrimary(ray, r0, z0, psi)
    return float(ray[1][0])


def wsSecrad(api, z, r0, z0, psi=1.):
    """axialHeights.py:63-74"""
    ray = api.sources.pointsource(0., 1)
    api.tran.transform(ray, 0, 0, 0, 0, -np.pi / 2, 0)
    api.tran.transform(ray, -r0 - 2., 0, -z, 0, 0, 0)
    api.surf.wsSecondary(ray, r0, z0, psi)
    return float(ray[1][0])


def traceZeta(api, pmin, R0=220., Z0=1e4, psi=1., offaxis=0., L=200., az=100., num=10 ** 4):
    """axialHeights.py:76-113.  Returns smin, smax, focus shift, hpd, rms and the focal bundle."""
    xp = api.xp
    a0 = wsPrimrad(api, pmin, R0, Z0, psi)
    a1 = wsPrimrad(api, pmin + L, R0, Z0, psi)
    rays = api.sources.subannulus(a0, a1, az / R0, int(num))
    api.tran.transform(rays, 0, 0, -Z0, 0, 0, 0)
    api.surf.wsPrimary(rays, R0, Z0, psi)
    rays[4] = rays[4] + np.sin(offaxis)
    rays[6] = -xp.sqrt(1. - rays[4] ** 2)
    api.tran.reflect(rays)
    api.surf.wsSecondary(rays, R0, Z0, psi)
    api.tran.reflect(rays)
    z = rays[3] if xp is np else rays[3].cpu().numpy()
    smax, smin = float(np.nanmax(z)), float(np.nanmin(z))
    f = api.surf.focusI(rays)
    return smin, smax, f, api.anal.hpd(rays), api.anal.rmsCentroid(rays), rays


def onaxisMerit(api, pmin, smax, R0=220., Z0=1e4, L=200., psi=1., num=10 ** 3):
    """axialHeights.py:115-136.  Returns (rays outside the secondary's extent, mean strike height)."""
    xp = api.xp
    a0 = wsPrimrad(api, pmin, R0, Z0, psi)
    a1 = wsPrimrad(api, pmin + L, R0, Z0, psi)
    rays = api.sources.annulus(a0, a1, int(num))
    api.tran.transform(rays, 0, 0, -Z0, 0, 0, 0)
    api.surf.wsPrimary(rays, R0, Z0, psi)
    api.tran.reflect(rays)
    api.surf.wsSecondary(rays, R0, Z0, psi)
    out = xp.logical_or(rays[3] < smax - L, rays[3] > smax)
    return int(out.sum()), float(rays[3].mean())


if __name__ == "__main__":
    api = Api.amd()
    np.random.seed(0)
    smin, smax, f, hpd, rms, _ = traceZeta(api, 1e4 + 50., offaxis=1e-4, num=10 ** 5)
    print("secondary strike heights %.3f .. %.3f mm, focus shift %.4f mm, hpd %.3e mm, rms %.3e mm" % (smin, smax, f, hpd, rms))
