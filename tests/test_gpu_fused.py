"""GPU tests of the fused ray program (pxf_program_run / pyxfocus_amd.fused()): results must be
BIT-IDENTICAL to issuing the same calls one launch at a time, for every fusable operation, with
and without `ind`, including failed (sentinel) rays."""
import numpy as np
import pytest

from oracle import cases
from util import ALLCOLS, bits_equal, call_gpu, to_dev, to_host

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev_rays(c):
    return [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]


def host(rays):
    return [r.cpu().numpy() for r in rays]


def same(a, b):
    for k, u, v in zip(ALLCOLS, a, b):
        assert bits_equal(u.cpu().numpy(), v.cpu().numpy()).all(), "column " + k


def chain(rays, tran, surf, nr=None):
    tran.transform(rays, 0, 0, -(cases.Z0 + 2 * cases.LMIR), 0, 0, 0)
    surf.wolterprimary(rays, cases.R0, cases.Z0, cases.PSI, nr=nr)
    tran.reflect(rays)
    surf.woltersecondary(rays, cases.R0, cases.Z0, cases.PSI)
    tran.reflect(rays)
    surf.flat(rays, nr=nr)


@pytest.mark.parametrize("n", [1, 2, 1001, 500_000, 4_000_001])
def test_fused_chain_equals_eager_and_golden(golden, n):
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(n)
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    c = cases.annulus_rays(rng, n, rin, rout, 0.)
    c["ux"][:] = rng.normal(size=n)          # the dropped transform-normal must not matter
    a, b = dev_rays(c), dev_rays(c)
    chain(a, tran, surf)
    with pxf.fused() as rec:
        chain(b, tran, surf)
    assert rec.launches == 1 and rec.fused_ops == 6
    same(a, b)
    if n == 1001:
        g = golden["chain_config2"]
        r = [torch.from_numpy(g["in/" + k].copy()).cuda() for k in ALLCOLS]
        with pxf.fused():
            chain(r, tran, surf)
        for k, t in zip(ALLCOLS, r):
            assert bits_equal(t.cpu().numpy(), g["out/" + k]).all(), k


def test_fused_opd_and_offaxis():
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(3)
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    c = cases.annulus_rays(rng, 300_001, rin, rout, 0.)
    c["l"][:] = 1e-3
    c["n"][:] = -np.sqrt(1 - 1e-6)
    a, b = dev_rays(c), dev_rays(c)
    chain(a, tran, surf, nr=1.25)
    with pxf.fused():
        chain(b, tran, surf, nr=1.25)
    same(a, b)


def test_fused_every_op_with_and_without_ind():
    """A long mixed program: frame moves (general rotation), conic incl. missed rays, grating with
    scalar and per-ray wavelength, masked and unmasked -- eager vs fused."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(17)
    n = 200_003
    c = cases._randrays(rng, n, nsign=-1.)
    c["l"] *= .1; c["m"] *= .1
    c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    c["z"] = rng.uniform(50, 150, n)
    c["x"][:5] = 5e4                                   # conic misses -> in-band sentinel
    m1 = rng.random(n) < .5
    m2 = np.where(rng.random(n) < .3)[0]
    wave = rng.uniform(1., 3., n)

    def program(r):
        tran.transform(r, 1., -2., 3., .01, -.02, .03)
        surf.conic(r, 1000., -1.)
        tran.reflect(r, ind=m1)
        tran.itransform(r, .5, .5, -.5, 0., .1, 0., ind=m2)
        surf.flat(r, ind=m1)
        tran.radgrat(r, 160. / 11832., 1., 2.4)
        tran.radgrat(r, 160. / 11832., -1., wave, ind=m1)
        surf.conic(r, -2000., .5, nr=1.5)
        tran.transform(r, 0, 0, 10., 0, 0, 0)
        surf.flat(r, nr=1.)
        tran.reflect(r)

    a, b = dev_rays(c), dev_rays(c)
    program(a)
    with pxf.fused() as rec:
        program(b)
    assert rec.launches == 1 and rec.fused_ops == 11
    same(a, b)


def test_fused_flushes_around_unfusable_calls():
    import pyxfocus_amd as pxf
    import pyxfocus_amd.analyses as anal
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(5)
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    c = cases.annulus_rays(rng, 100_000, rin, rout, 0.)
    a, b = dev_rays(c), dev_rays(c)
    chain(a, tran, surf)
    ra = anal.rmsCentroid(a)
    va = tran.vignette(a, ind=np.arange(0, 100_000, 3))
    surf.flat(va)
    with pxf.fused() as rec:
        chain(b, tran, surf)
        rb = anal.rmsCentroid(b)                   # reduction inside the block -> flush first
        vb = tran.vignette(b, ind=np.arange(0, 100_000, 3))
        surf.flat(vb)                              # a different bundle -> its own program
    assert ra == rb
    same(a, b)
    same(va, vb)
    assert rec.launches == 2
    # an exception inside the block discards the pending program instead of running it
    d = dev_rays(c)
    with pytest.raises(RuntimeError):
        with pxf.fused():
            tran.transform(d, 0, 0, 1., 0, 0, 0)
            raise RuntimeError("boom")
    assert bits_equal(d[3].cpu().numpy(), c["z"]).all()


def test_fused_more_ops_than_one_program_holds():
    import pyxfocus_amd as pxf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(8)
    c = cases._randrays(rng, 10_001)
    a, b = dev_rays(c), dev_rays(c)
    for i in range(75):
        tran.transform(a, .1, 0, 0, 0, 0, 1e-3 * i)
    with pxf.fused() as rec:
        for i in range(75):
            tran.transform(b, .1, 0, 0, 0, 0, 1e-3 * i)
    assert rec.launches == 3 and rec.fused_ops == 75
    same(a, b)


def test_fused_batch1_ops_equal_eager():
    """SPO cones, linear grating, sphere, cylinder, paraxial lens, refraction in one program."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(23)
    n = 150_001
    c = cases.annulus_rays(rng, n, 599.6, 601.2, 150.)
    c["l"] = rng.normal(0, 2e-4, n); c["m"] = rng.normal(0, 2e-4, n)
    c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    c["m"][:7] = 1.; c["l"][:7] = 0.; c["n"][:7] = 0.; c["x"][:7] = 5e4          # miss the cone
    m1 = rng.random(n) < .6
    ow = rng.uniform(-3, 3, n)

    def program(r):
        surf.spoPrimary(r, 600., 12000.)
        tran.reflect(r)
        tran.transform(r, 0, 0, -30., 0, 0, 0)
        surf.spoSecondary(r, 600., 12000., ind=m1)
        tran.reflect(r, ind=m1)
        tran.grat(r, 160., 1., 2.4)
        tran.grat(r, 200., ow, 1., ind=m1)
        surf.paraxial(r, 1e4)
        surf.paraxialY(r, -2e4)
        tran.transform(r, 0, 0, -500., 0, 0, 0)
        surf.sphere(r, 900., nr=1.5)
        tran.refract(r, 1., 1.5)
        surf.cyl(r, 2000.)
        surf.flat(r)

    a, b = dev_rays(c), dev_rays(c)
    program(a)
    with pxf.fused() as rec:
        program(b)
    assert rec.launches == 1 and rec.fused_ops == 14
    same(a, b)


def test_segmented_program_equals_per_segment_loop():
    """Multi-shell bundle: one launch with per-segment parameters (LDS-staged) == a loop of per-shell
    fused programs == a loop of plain calls, bit for bit; segmented source == per-shell sources."""
    import math
    import pyxfocus_amd as pxf
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import conicsolve as con
    F, lmir = 12e3, 50.
    shells = [(320. + .775 * i, math.sqrt(F * F - (320. + .775 * i) ** 2), .01 * (i % 7 - 3), .15) for i in range(37)]
    per = 1000                                   # even, several segments per 512-ray workgroup tile boundary
    n = per * len(shells)
    geo = [(r0, z0, a, w, float(con.primrad(z0, r0, z0)), float(con.primrad(z0 + lmir, r0, z0)))
           for (r0, z0, a, w) in shells]

    def shell(r, r0, z0, a):
        tran.transform(r, 0, 0, -(z0 + 2 * lmir), 0, 0, a)
        surf.wolterprimary(r, r0, z0)
        tran.reflect(r)
        surf.woltersecondary(r, r0, z0)
        tran.reflect(r)

    def fresh():
        buf = torch.empty((10, n), dtype=torch.float64, device="cuda")
        return [buf[i] for i in range(10)]

    # per-shell sources vs one segmented source
    a = fresh()
    for k, (r0, z0, ang, w, rin, rout) in enumerate(geo):
        sources.subannulus(rin, rout, w, per, zhat=-1., rng="philox", seed=5, first_index=k * per,
                           out=[c[k * per:(k + 1) * per] for c in a])
    b = fresh()
    sources.segments("subannulus", [(g[4], g[5], g[3], -1.) for g in geo], per, out=b, seed=5)
    same(a, b)
    c = [t.clone() for t in a]
    # (1) loop of plain calls  (2) loop of per-shell fused programs  (3) one segmented launch
    for k, g in enumerate(geo):
        shell([t[k * per:(k + 1) * per] for t in a], g[0], g[1], g[2])
    for k, g in enumerate(geo):
        with pxf.fused():
            shell([t[k * per:(k + 1) * per] for t in b], g[0], g[1], g[2])
    prog = pxf.compile_segments(c, per, shell, [(g[0], g[1], g[2]) for g in geo])
    assert tuple(prog.table.shape) == (len(shells), 5, 9)
    prog.run()
    same(a, b)
    same(a, c)
    # a different sequence in one segment is rejected
    def bad(r, r0, z0, a_):
        shell(r, r0, z0, a_)
        if r0 > 330.:
            surf.flat(r)
    with pytest.raises(Exception, match="different operation sequence"):
        pxf.compile_segments(c, per, bad, [(g[0], g[1], g[2]) for g in geo])


def test_capture_replays_trace_as_hip_graph():
    """pyxfocus_amd.capture: eager calls + a fused block captured once, replayed on refilled inputs."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from oracle import cases
    rng = np.random.default_rng(21)
    c = cases.annulus_rays(rng, 5000, cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR), 0.)
    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]

    def trace():
        tran.transform(rays, 0, 0, -(cases.Z0 + 2 * cases.LMIR), 0, 0, 0)
        with pxf.fused():
            surf.wolterprimary(rays, cases.R0, cases.Z0)
            tran.reflect(rays)
            surf.woltersecondary(rays, cases.R0, cases.Z0)
            tran.reflect(rays)
        surf.flat(rays)

    trace()
    want = [r.clone() for r in rays]
    for k, t in zip(ALLCOLS, rays):
        t.copy_(torch.from_numpy(c[k]))
    g = pxf.capture(trace)
    for rep in range(3):
        for k, t in zip(ALLCOLS, rays):
            t.copy_(torch.from_numpy(c[k]))
        g.replay()
        torch.cuda.synchronize()
        for w, r in zip(want, rays):
            assert bits_equal(r.cpu().numpy(), w.cpu().numpy()).all(), rep


def _zern_bundle(n, seed):
    rng = np.random.default_rng(seed)
    c = {k: np.zeros(n) for k in ALLCOLS}
    rho, th = 45. * np.sqrt(rng.random(n)), rng.random(n) * 2 * np.pi
    c["x"], c["y"], c["z"] = rho * np.cos(th), rho * np.sin(th), np.full(n, 10.)
    c["l"] = np.full(n, .01)
    c["n"] = -np.sqrt(1 - c["l"] ** 2)
    return c


@pytest.mark.parametrize("nterms,nr", [(36, None), (36, 1.5), (10, None), (66, None), (120, 1.2)])
def test_fused_program_around_a_zernike_trace(nterms, nr):
    """BASELINE config 4 as ONE launch: transform -> zernsurf -> reflect -> flat inside fused() equals
    the four stand-alone launches (the Zernike arithmetic is tolerance-parity, the other operations
    are the same bits given the same inputs), for the instantiated order ranges and the OPD variant;
    every radial order is fused (instantiations for orders <= 4, 7, 10, 15)."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(nterms)
    ro, ao = cases.noll_orders(nterms)
    coeff = rng.normal(0, 1e-3, nterms) / (1 + np.asarray(ro))
    coeff[0] = 0.
    c = _zern_bundle(20001, 5)

    def chain(rays):
        tran.transform(rays, .5, -.25, 0., 1e-3, -2e-3, .1)
        surf.zernsurf(rays, coeff, 50., rorder=ro, aorder=ao, nr=nr)
        tran.reflect(rays)
        surf.flat(rays, nr=nr)

    a, b = dev_rays(c), dev_rays(c)
    chain(a)
    with pxf.fused() as rec:
        chain(b)
    assert rec.launches == 1 and rec.fused_ops == 4
    for k, u, v in zip(ALLCOLS, a, b):
        u, v = u.cpu().numpy(), v.cpu().numpy()
        assert np.array_equal(np.isnan(u), np.isnan(v)), k
        f = ~np.isnan(u)
        # 120 terms (n <= 14): Newton converges slowly on the oblique rays, and two roundings of the same
        # evaluation may stop one step apart -- by the reference's own 1e-10 stopping step
        tol = 1e-12 if nterms <= 66 else 2e-9
        assert np.abs(u[f] - v[f]).max() <= tol * (1 + np.abs(u[f]).max()), (k, np.abs(u[f] - v[f]).max())


def test_fused_zernike_masks_limits_and_liveness():
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    ro, ao = cases.noll_orders(21)
    coeff = np.random.default_rng(3).normal(0, 1e-3, 21)
    coeff[0] = 0.
    c = _zern_bundle(5000, 9)
    # an opd column that is live around a plain (non-OPD) Zernike trace must come through untouched
    a, b = dev_rays(c), dev_rays(c)
    for rays, ctx in ((a, None), (b, pxf.fused)):
        def chain():
            surf.flat(rays, nr=1.3)                       # writes opd
            tran.transform(rays, 0, 0, -10., 0, 0, 0)
            surf.zernsurf(rays, coeff, 50., rorder=ro, aorder=ao)
            tran.reflect(rays)
        if ctx:
            with ctx():
                chain()
        else:
            chain()
    assert bits_equal(a[0].cpu().numpy(), b[0].cpu().numpy()).all() and float(a[0].abs().max()) > 0
    for u, v in zip(a[1:], b[1:]):
        assert np.abs(u.cpu().numpy() - v.cpu().numpy()).max() < 1e-11
    # two Zernike traces in one block -> two launches; 20 operations around one -> split at 16
    r = dev_rays(c)
    with pxf.fused() as rec:
        surf.zernsurf(r, coeff, 50., rorder=ro, aorder=ao)
        tran.reflect(r)
        tran.transform(r, 0, 0, -5., 0, 0, 0)
        surf.zernsurf(r, coeff, 50., rorder=ro, aorder=ao)
    assert rec.launches == 2
    r = dev_rays(c)
    with pxf.fused() as rec:
        surf.zernsurf(r, coeff, 50., rorder=ro, aorder=ao)
        for i in range(20):
            tran.transform(r, 1e-3, 0, 0, 0, 0, 1e-4)
    assert rec.launches == 2 and rec.fused_ops == 21


def test_fused_zernike_with_masks_splits_instead_of_fusing():
    """ind= masks and a Zernike trace do not share a program (register budget): the block still gives the
    stand-alone results, with more launches."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    ro, ao = cases.noll_orders(21)
    coeff = np.random.default_rng(4).normal(0, 1e-3, 21)
    coeff[0] = 0.
    c = _zern_bundle(4000, 10)
    ind = np.random.default_rng(5).random(4000) < .5
    a, b = dev_rays(c), dev_rays(c)

    def chain(r):
        tran.transform(r, .1, 0, 0, 0, 0, 1e-3, ind=ind)
        surf.zernsurf(r, coeff, 50., rorder=ro, aorder=ao)
        tran.reflect(r, ind=ind)
        surf.flat(r)

    chain(a)
    with pxf.fused() as rec:
        chain(b)
    assert rec.launches >= 2
    # (21 terms: the stand-alone launch uses the order-5 instantiation, the program the order-7 one --
    # same arithmetic, roundings of the two may differ in the last bit)
    for k, u, v in zip(ALLCOLS, a, b):
        assert np.abs(u.cpu().numpy() - v.cpu().numpy()).max() <= 1e-12, k


def test_fused_zernike_with_a_heavy_neighbour_splits():
    """Only frame changes, reflect and flat share a program with a Zernike trace; a Wolter surface next
    to it goes into its own program -- same results."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    ro, ao = cases.noll_orders(36)
    coeff = np.random.default_rng(6).normal(0, 1e-3, 36)
    coeff[0] = 0.
    c = _zern_bundle(4000, 12)
    a, b = dev_rays(c), dev_rays(c)

    def chain(r):
        surf.zernsurf(r, coeff, 50., rorder=ro, aorder=ao)
        tran.reflect(r)
        surf.conic(r, 5000., -1.)
        tran.reflect(r)
        surf.flat(r)

    chain(a)
    with pxf.fused() as rec:
        chain(b)
    assert rec.launches == 2 and rec.fused_ops == 5
    same(a, b)


# ---- opt-in tolerance mode (pxf_program_run_fast) ---------------------------------------------------------
def _rms_focal(got, want):
    """RMS / max position error, and the rays whose fp32 `delta` in a flat (surfacesf.f95:4-15) rounded the
    other way: -z/n is rounded to SINGLE precision there, so a ray whose quotient sits within the two
    implementations' ~1e-12 relative difference of a rounding boundary lands one fp32 quantum of the
    propagation distance (~1e-3 mm along the ray, <= ~1e-4 mm sideways) apart.  That happens to about one ray
    in a million per flat -- in the reference against ITSELF too, under a 1-ulp change of an input -- and is
    counted separately: those rays are not less accurate, the reference's quantisation just fell elsewhere."""
    f = np.isfinite(want["x"]) & np.isfinite(want["y"])
    assert np.array_equal(f, np.isfinite(got["x"]) & np.isfinite(got["y"]))
    d = np.sqrt(sum((got[k][f] - want[k][f]) ** 2 for k in "xyz"))
    flip = d > 1e-7
    assert (d[flip] < 2e-3).all()                            # bounded by the fp32 quantum of an ~8 m leg
    rest = d[~flip]
    ok = f.copy()
    ok[f] = ~flip
    return float(np.sqrt((rest ** 2).mean())), float(rest.max()), float(f.mean()), int(flip.sum()), len(d), ok


@pytest.mark.parametrize("config", [2, 3])
def test_fast_program_rms_vs_oracle(port, config):
    """fused(fast=True) on BASELINE configs 2 and 3 against the ORACLE: RMS ray-position error at the
    focal plane < 1e-10 mm (the north-star's bar; the reference's own 1-ulp jitter is 4.7e-11,
    tests/test_oracle.py::test_reference_conditioning_one_ulp), direction cosines and normals to 1e-13;
    one launch; and the default fused() stays bit-identical."""
    import os
    import sys
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from oracle import cases, wrappers
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools import workloads as wl
    chain = wl.config2_chain if config == 2 else wl.config3_chain
    rng = np.random.default_rng(50 + config)
    n = 200_000
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    if config == 2:
        c = cases.annulus_rays(rng, n, rin, rout, 0.)
    else:                                                    # sub-aperture that reaches the grating
        c = {k: np.zeros(n) for k in ALLCOLS}
        rho, th = np.sqrt(rin ** 2 + rng.random(n) * (rout ** 2 - rin ** 2)), rng.random(n) * .5 - .25
        c["x"], c["y"], c["n"] = rho * np.cos(th), rho * np.sin(th), np.full(n, -1.)
    osrc, osurf, otran, oanal = wrappers.bind(port)
    want = [c[k].copy() for k in ALLCOLS]
    chain(want, otran, osurf)
    want = dict(zip(ALLCOLS, want))
    rays = dev_rays(c)
    with pxf.fused(fast=True) as rec:
        chain(rays, tran, surf)
    assert rec.launches == 1 and rec.fast_launches == 1
    got = dict(zip(ALLCOLS, host(rays)))
    rms, mx, valid, flips, nr, ok = _rms_focal(got, want)
    print("config %d fast program: rms position error %.3e mm, max %.3e mm (valid %.3f), fp32-flat flips %d of %d"
          % (config, rms, mx, valid, flips, nr))
    assert valid > .99 and rms < 1e-10 and mx < 2e-9 and flips <= 1e-4 * nr, (rms, mx, flips)
    for k in ("l", "m", "n", "ux", "uy", "uz"):            # (a flipped ray meets the grating ~5e-5 mm aside)
        assert np.abs(got[k][ok] - want[k][ok]).max() < 1e-12, k
    exact = dev_rays(c)
    with pxf.fused() as rec:
        chain(exact, tran, surf)
    assert rec.fast_launches == 0
    if config == 2:
        for k, v in zip(ALLCOLS, host(exact)):
            assert bits_equal(v, want[k]).all(), k


def test_fast_program_falls_back_and_covers_every_supported_op(port):
    """(a) a program with an unsupported operation (conic) or an ind= mask runs through the EXACT kernel
    even under fused(fast=True); the C entry itself refuses it (PXF_ERANGE).  (b) each supported
    operation alone against the oracle on generic (off-axis, tilted) rays, 1e-9 relative: frame changes
    with all three rotations, spoCone hits and misses, radgrat with evanescent rays, reflect, flat."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from oracle import cases
    from pyxfocus_amd import _fused, _lib
    rng = np.random.default_rng(61)
    base = {name: (fn, cols, args) for name, fn, cols, args in cases.cases(num=4001, seed=61)}
    c = base["conic_parab"][1]
    a, b = dev_rays(c), dev_rays(c)
    tran.transform(a, 1., 2., 3., .01, .02, .03)
    surf.conic(a, 2000., -1.)
    with pxf.fused(fast=True) as rec:
        tran.transform(b, 1., 2., 3., .01, .02, .03)
        surf.conic(b, 2000., -1.)
    assert rec.launches == 1 and rec.fast_launches == 0
    same(a, b)
    op = _fused.PxfOp()
    op.code = 3
    ptrs = (_fused.C.c_void_p * 10)(*[t.data_ptr() for t in b])
    assert _lib.lib().pxf_program_run_fast((_fused.PxfOp * 1)(op), 1, ptrs, len(c["x"]), None) == -4      # PXF_ERANGE
    checked = 0
    for name, (fn, cols, args) in base.items():
        short = fn.split(".")[1]
        if short not in ("flat", "transform", "itransform", "reflect", "wolterprimary", "woltersecondary", "radgrat",
                         "spocone"):
            continue
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        t = to_dev(cols)
        with pxf.fused(fast=True) as rec:
            call_gpu(fn, t, args)
        assert rec.fast_launches == 1, name
        got = to_host(t)
        for k in ALLCOLS:
            assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), (name, k)
            inf = np.isinf(want[k])
            assert np.array_equal(got[k][inf], want[k][inf]), (name, k)          # x/0 = +-inf survives
            f = np.isfinite(want[k])
            # positions of the Newton surfaces carry the reference's own along-ray jitter (~1e-8 mm for the
            # secondary); everything else agrees to rounding
            tol = 5e-8 if (short in ("wolterprimary", "woltersecondary") and k in "xyz") else 1e-9 * (1 + np.abs(want[k][f]).max())
            assert np.abs(got[k][f] - want[k][f]).max() <= tol, (name, k, np.abs(got[k][f] - want[k][f]).max())
        checked += 1
    assert checked >= 8


MIRRORS = ("woltersine", "wolterprimll", "woltersecll", "ellipsoidwoltll", "wsprimary", "wssecondary",
           "wsprimaryback", "wssecondaryback")


def _mirror_cases(num, seed):
    return {name: (fn, cols, args) for name, fn, cols, args in cases.cases(num=num, seed=seed) if name in MIRRORS}


@pytest.mark.parametrize("n", [64, 777, 60_001])
def test_fused_program_around_mirrors(n):
    """W-S / Legendre-Legendre / woltersine mirrors INSIDE a program (pxf_program_mirror.hip): small frame
    change -> mirror -> reflect -> frame change back -> flat as ONE launch, bit-identical to the five
    stand-alone launches (the mirror Ops are the same device code: WSOp::apply, ll_trace, WolterSineOp::apply),
    failed W-S rays (position restored, normal kept) included."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    for name, (fn, cols, args) in _mirror_cases(n, 17).items():
        def chain(t):
            rays = [t[k] for k in ALLCOLS]
            tran.transform(rays, 1e-3, -2e-3, 0., 1e-6, -1e-6, 0.)
            call_gpu(fn, t, args)
            tran.reflect(rays)
            tran.itransform(rays, 1e-3, -2e-3, 0., 1e-6, -1e-6, 0.)
            surf.flat(rays)
        a, b = to_dev(cols), to_dev(cols)
        chain(a)
        with pxf.fused() as rec:
            chain(b)
        assert rec.launches == 1 and rec.fused_ops == 5, (name, rec.launches, rec.fused_ops)
        for k in ALLCOLS:
            assert bits_equal(a[k].cpu().numpy(), b[k].cpu().numpy()).all(), (name, k)


def test_fused_mirror_pairs_limits_and_fallbacks():
    """Two mirrors in one program (a W-S pair; two 11 x 11 Legendre-Legendre sets sharing the coefficient pool),
    and everything the mirror program does not take -- `ind=` masks, a 13th operation, matrices that overflow
    the pool, azimuthal orders > 11, a Zernike trace next to a mirror, check=True -- falls back to separate
    launches with the same bits."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import native
    mc = _mirror_cases(4001, 23)
    rng = np.random.default_rng(5)

    def both(cols, chain, launches, ops=None):
        a, b = to_dev(cols), to_dev(cols)
        chain(a, [a[k] for k in ALLCOLS])
        with pxf.fused() as rec:
            chain(b, [b[k] for k in ALLCOLS])
        assert rec.launches == launches and (ops is None or rec.fused_ops == ops), (rec.launches, rec.fused_ops)
        for k in ALLCOLS:
            assert bits_equal(a[k].cpu().numpy(), b[k].cpu().numpy()).all(), k

    # W-S pair: primary -> reflect -> secondary -> reflect -> flat
    fn1, cols, a1 = mc["wsprimary"]
    fn2, _, a2 = mc["wssecondary"]

    def ws_pair(t, rays):
        call_gpu(fn1, t, a1)
        tran.reflect(rays)
        call_gpu(fn2, t, a2)
        tran.reflect(rays)
        surf.flat(rays)
    both(cols, ws_pair, 1, 5)

    # L-L pair with 11 x 11 coefficient sets (what the reference's examples fit): one pool, one launch
    fnp, cols, ap = mc["wolterprimll"]
    fns, _, as_ = mc["woltersecll"]

    def ll_set(ka, kt, seed):
        r = np.random.default_rng(seed)
        ax, az = np.meshgrid(np.arange(ka + 1), np.arange(kt + 1), indexing="ij")
        co = r.normal(0., 2e-5, ax.size) / (1 + ax.flatten() + az.flatten())
        return co, ax.flatten().astype(np.int32), az.flatten().astype(np.int32)

    def ll_pair(sets):
        def chain(t, rays):
            call_gpu(fnp, t, tuple(ap[:-3]) + sets[0])
            tran.reflect(rays)
            tran.transform(rays, 0, 0, 1e-2, 0, 1e-6, 0)
            call_gpu(fns, t, tuple(as_[:-3]) + sets[1])
            tran.reflect(rays)
        return chain
    both(cols, ll_pair((ll_set(10, 10, 1), ll_set(10, 10, 2))), 1, 5)
    both(cols, ll_pair((ll_set(3, 2, 1), ll_set(14, 7, 2))), 1, 5)          # mixed shapes: common column count 8
    both(cols, ll_pair((ll_set(13, 11, 1), ll_set(13, 11, 2))), 2)          # 2 x 14 x 12 cells > pool: two programs
    both(cols, ll_pair((ll_set(4, 12, 1), ll_set(3, 3, 2))), 1, 4)          # azimuthal order 12: that mirror runs stand-alone

    # an ind= mask on the mirror, a 13th operation, a conic in between: separate launches, same bits
    fn, cols, args = mc["wsprimary"]
    msk = torch.from_numpy(rng.random(len(cols["x"])) < .5).cuda()

    def masked(t, rays):
        tran.transform(rays, 0, 0, 1e-3, 0, 0, 0)
        call_gpu(fn, t, args, mask=msk.to(torch.uint8))
        tran.reflect(rays)
    both(cols, masked, 2)

    def long_chain(t, rays):
        for _ in range(6):
            tran.transform(rays, 0, 0, 1e-3, 0, 0, 0)
            tran.itransform(rays, 0, 0, 1e-3, 0, 0, 0)
        call_gpu(fn, t, args)
    both(cols, long_chain, 2, 13)

    def with_conic(t, rays):
        call_gpu(fn, t, args)
        tran.reflect(rays)
        surf.conic(rays, -500., -1.)
        tran.reflect(rays)
    both(cols, with_conic, 2, 4)

    # a Zernike trace and a mirror never share a program
    ro, ao = cases.noll_orders(10)
    zc = rng.normal(0, 1e-4, 10)

    def with_zern(t, rays):
        call_gpu(fn, t, args)
        tran.reflect(rays)
        surf.zernsurf(rays, zc, 2000., rorder=ro, aorder=ao)
    both(cols, with_zern, 2, 3)

    # check=True reads the positions back: flushes before and after
    a = to_dev(cols)
    ra = [a[k] for k in ALLCOLS]
    tran.transform(ra, 0, 0, 1e-3, 0, 0, 0)
    want = surf.wsPrimary(ra, cases.R0, cases.Z0, cases.PSI, check=True)
    b = to_dev(cols)
    rb = [b[k] for k in ALLCOLS]
    with pxf.fused():
        tran.transform(rb, 0, 0, 1e-3, 0, 0, 0)
        got = surf.wsPrimary(rb, cases.R0, cases.Z0, cases.PSI, check=True)
    assert torch.equal(want, got)
    for k in ALLCOLS:
        assert bits_equal(a[k].cpu().numpy(), b[k].cpu().numpy()).all(), k


def test_graze_angle_value_used_inside_a_fused_block_is_valid():
    """`refl = reflectivity(grazeAngle(rays))` inside `with fused():` (sector.py:592-612) must
    read real angles.  A call WITHOUT out= flushes the pending program and launches -- its return value is valid at
    once; only a call WITH out= is recorded and filled when the block runs."""
    import warnings
    import pyxfocus_amd as pxf
    import pyxfocus_amd.analyses as anal
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(5)
    n = 20_001
    c = cases.annulus_rays(rng, n, cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR), 0.)
    a, b, d = dev_rays(c), dev_rays(c), dev_rays(c)
    tran.transform(a, 0, 0, -(cases.Z0 + 2 * cases.LMIR), 0, 0, 0)
    surf.wolterprimary(a, cases.R0, cases.Z0, cases.PSI)
    want1 = anal.grazeAngle(a).cpu().numpy()
    tran.reflect(a)
    surf.woltersecondary(a, cases.R0, cases.Z0, cases.PSI)
    want2 = anal.grazeAngle(a).cpu().numpy()
    tran.reflect(a)
    assert np.abs(want1).min() > 1e-3 and np.abs(want2).min() > 1e-3
    with pxf.fused() as rec:
        tran.transform(b, 0, 0, -(cases.Z0 + 2 * cases.LMIR), 0, 0, 0)
        surf.wolterprimary(b, cases.R0, cases.Z0, cases.PSI)
        g1 = anal.grazeAngle(b)
        w1 = torch.sin(g1).cpu().numpy()                    # consumed INSIDE the block: must be the real thing
        tran.reflect(b)
        surf.woltersecondary(b, cases.R0, cases.Z0, cases.PSI)
        g2 = anal.grazeAngle(b)
        w2 = torch.sin(g2).cpu().numpy()
        tran.reflect(b)
    assert np.allclose(w1, np.sin(want1), rtol=0, atol=1e-15)
    assert np.allclose(w2, np.sin(want2), rtol=0, atol=1e-15)
    assert np.array_equal(g1.cpu().numpy(), want1) and np.array_equal(g2.cpu().numpy(), want2)
    assert rec.launches == 3                                # two flushes ahead of the angle launches + the block end
    same(a, b)
    # with out=: recorded, ONE launch, the columns are filled when the block has run
    o1, o2 = torch.full((n,), -7., dtype=torch.float64, device="cuda"), torch.full((n,), -7., dtype=torch.float64, device="cuda")
    with pxf.fused() as rec:
        tran.transform(d, 0, 0, -(cases.Z0 + 2 * cases.LMIR), 0, 0, 0)
        surf.wolterprimary(d, cases.R0, cases.Z0, cases.PSI)
        assert anal.grazeAngle(d, out=o1) is o1
        tran.reflect(d)
        surf.woltersecondary(d, cases.R0, cases.Z0, cases.PSI)
        anal.grazeAngle(d, out=o2)
        tran.reflect(d)
    assert rec.launches == 1 and rec.fused_ops == 7
    assert np.array_equal(o1.cpu().numpy(), want1) and np.array_equal(o2.cpu().numpy(), want2)
    same(a, d)
    # nesting a fast block into an exact one (or the reverse) joins the OUTER block's mode -- and says so
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        with pxf.fused():
            with pxf.fused(fast=True):
                pass
        with pxf.fused(fast=True):
            with pxf.fused(fast=True):
                pass
    assert len(wlist) == 1 and "OUTER" in str(wlist[0].message)
    # a fast block recording a Wolter-I surface beyond z0 = 12 m warns (jitter grows as z0^2)
    e = dev_rays(c)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        with pxf.fused(fast=True):
            tran.transform(e, 0, 0, -(20000. + 2 * cases.LMIR), 0, 0, 0)
            surf.wolterprimary(e, 400., 20000., 1.)
    assert any("z0" in str(w.message) for w in wlist)
