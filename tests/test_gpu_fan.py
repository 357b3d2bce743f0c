"""grating.fanArray (pxf_fan_search / pxf_fan_finish: every ray walks a fanned grating array in registers) against the
loop it replaces, run through the drop-in API on the same device -- the `while` loop of the reference's
examples/arcus/sector.py:663-704 as grating.fanArray's docstring restates it -- BIT FOR BIT, over random fans (either
sign of the pitch, different thresholds), rays that meet their grating anywhere between the first and the ~60th row,
grazing rounds, and bundles salted with the values the kernels' shortcuts must step around (-0, +0, denormals, huge and
tiny magnitudes in every column).  The Arcus fixture (tests/test_gpu_config5.py) pins one geometry; this walks the
parameter space."""
import math

import numpy as np
import pytest

from util import bits_equal

torch = pytest.importorskip("torch")


def fan_rays(rng, n, H, Lg, ang, inc, kmax, salt=False):
    """n rays that each meet ONE grating of the fan (hub on the x axis, row k = the z=0 plane of the frame after k pitch
    rotations): a point on row k's grating, a direction at graze angle ~inc to it, moved back along the ray, and
    expressed in the frame of row 0."""
    k = rng.integers(0, kmax + 1, n)
    rho = H + Lg * (.02 + .96 * rng.random(n))
    x = 20. * (rng.random(n) - .5)
    y = -np.sqrt(rho ** 2 - x ** 2)
    z = np.zeros(n)
    a = inc * (.7 + .6 * rng.random(n))
    l = 2e-3 * rng.standard_normal(n)
    m = -np.cos(a) * np.sqrt(1 - l ** 2)            # towards the hub, dipping through the plane
    nn = -np.sin(a) * np.sqrt(1 - l ** 2)
    t = 5. + 40. * rng.random(n)                    # back along the ray: above the plane, short of the previous row
    x, y, z = x - t * l, y - t * m, z - t * nn
    # undo k pitch rotations: transform(0,0,0,ang,0,0) rotates vectors about x by -(-ang) ... the kernels' own
    # convention is found by construction below (the caller checks that every ray hits)
    th = k * ang
    c, s = np.cos(th), np.sin(th)
    y, z = c * y - s * z, s * y + c * z
    m, nn = c * m - s * nn, s * m + c * nn
    cols = [np.zeros(n), x, y, z, l, m, nn, rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)]
    if salt:
        tiny = np.nextafter(0., 1.) * 37
        for col, vals in ((1, (-0., 0., tiny, -tiny)), (4, (-0., 0., tiny)), (7, (-0., 0., tiny, 1e300, -1e-300)),
                          (8, (-0., 0., 1e-310, 1e250)), (9, (-0., 0., -1e-320, 1e-200))):
            for v in vals:
                cols[col][rng.integers(0, n, 5)] = v
    return cols


def fan_loop(xp, tran, surf, rays, H, Lg, ang, dpermm, order, wave, thr, maxiter):
    """The host loop grating.fanArray replaces (its docstring; sector.py:663-704), on any PyXFocus-shaped binding."""
    x, y = rays[1], rays[2]

    def inrange():
        rho = -xp.sqrt(x ** 2 + y ** 2) * xp.sign(y)
        return xp.logical_and(rho > H, rho < Lg + H)
    surf.flat(rays, ind=xp.abs(xp.arcsin(rays[6])) > thr)
    ind = inrange()
    ind2, prev, i = ind, ind, 0
    while int(prev.sum()) < len(x):
        i += 1
        assert i <= maxiter, "%d rays never reached a grating" % (len(x) - int(prev.sum()))
        if int(ind2.sum()) > 0:
            tran.reflect(rays, ind=ind2)
            tran.radgrat(rays, dpermm, order, wave, ind=ind2)
        tran.transform(rays, 0, 0, 0, ang, 0, 0)
        surf.flat(rays, ind=xp.logical_and(xp.logical_not(prev), xp.abs(xp.arcsin(rays[6])) > thr))
        ind = inrange()
        ind2 = xp.logical_and(xp.logical_not(prev), ind)
        prev = xp.logical_or(prev, ind)
    tran.reflect(rays, ind=ind2)
    tran.radgrat(rays, dpermm, order, wave, ind=ind2)
    return i


CASES = [  # H, Lg, inc [deg], sign of the pitch, thr, kmax, salted
    (11800., 95., 1.5, +1, .001, 60, False),
    (11800., 95., 1.5, +1, .001, 60, True),
    (2000., 40., 2.5, +1, .02, 25, True),          # a threshold inside the graze-angle range: grazing rounds happen
    (5000., 60., 1.0, -1, .001, 30, True),         # negative pitch (sin < 0), rays built the other way round
]


def test_fan_case_generator_on_the_oracle(port):
    """(CPU) The generator above really produces bundles in which every ray meets a grating -- checked with the same
    loop over the CPU checker's bindings, so that the GPU comparison below cannot pass vacuously."""
    from oracle import wrappers
    _, osurf, otran, _ = wrappers.bind(port)
    for H, Lg, inc, sgn, thr, kmax, salt in CASES:
        ang = sgn * Lg * math.sin(math.radians(inc)) / H * .95
        rays = fan_rays(np.random.default_rng(7), 4000, H, Lg, ang, math.radians(inc), kmax, salt)
        rows = fan_loop(np, otran, osurf, rays, H, Lg, ang, 160. / H, 1, .24, thr, 4 * kmax + 50)
        assert kmax // 2 <= rows <= 2 * kmax + 5, (H, rows)
        assert np.isfinite(rays[4]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("n", [70_001, 1_000_000])
def test_fan_array_equals_the_host_loop_bitwise(case, n):
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import grating
    H, Lg, inc, sgn, thr, kmax, salt = CASES[case]
    ang = sgn * Lg * math.sin(math.radians(inc)) / H * .95
    cols = fan_rays(np.random.default_rng(100 * case + 7), n, H, Lg, ang, math.radians(inc), kmax, salt)
    a = [torch.from_numpy(c.copy()).cuda() for c in cols]
    b = [torch.from_numpy(c.copy()).cuda() for c in cols]
    rows_a = fan_loop(torch, tran, surf, a, H, Lg, ang, 160. / H, 1, .24, thr, 4 * kmax + 50)
    ca, cb = tran.newCoords(), tran.newCoords()
    for _ in range(rows_a):
        tran.update_coords(ca, 0, 0, 0, ang, 0, 0)
    rows_b = grating.fanArray(b, H, Lg, ang, 160. / H, 1, .24, thr=thr, coords=cb, maxiter=4 * kmax + 50)
    assert rows_a == rows_b and rows_a >= kmax // 2
    assert np.array_equal(np.array(ca), np.array(cb))
    for k in range(1, 10):
        assert bits_equal(a[k].cpu().numpy(), b[k].cpu().numpy()).all(), (case, n, k)
