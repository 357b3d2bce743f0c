"""GPU tests of the PyXFocus-signature layer (transformations / surfaces / analyses / sources)
against vectors captured from the reference's OWN transformations.py + sources.py running over
its compiled Fortran (tests/golden/wrappers.npz, sources.npz; oracle/make_golden.py)."""
import numpy as np
import pytest

from util import ALLCOLS, bits_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def rays_from(g, prefix, dev="cuda"):
    return [torch.from_numpy(g["%s/%s" % (prefix, k)].copy()).to(dev) for k in ALLCOLS]


def host(rays):
    return [r.cpu().numpy() for r in rays]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda")


def assert_rays(got, g, prefix, exact=True, tol=0.):
    for k, a in zip(ALLCOLS, host(got)):
        w = g["%s/%s" % (prefix, k)]
        assert a.shape == w.shape, "%s/%s shape %s vs %s" % (prefix, k, a.shape, w.shape)
        if exact:
            assert bits_equal(a, w).all(), "%s/%s" % (prefix, k)
        else:
            assert (np.isnan(a) == np.isnan(w)).all()
            f = ~np.isnan(w)
            assert np.all(np.abs(a[f] - w[f]) <= tol * (1 + np.abs(w[f]))), "%s/%s max %g" % (
                prefix, k, np.abs(a[f] - w[f]).max())


def test_coords_and_applyT(golden):
    import pyxfocus_amd.transformations as tran
    g = golden["wrappers"]
    rays = rays_from(g, "coords/in")
    orig = tran.copy_rays(rays)
    coords = tran.newCoords()
    for t in g["coords/seq"]:
        tran.transform(rays, *t, coords=coords)
    tran.itransform(rays, *g["coords/iseq"], coords=coords)
    assert_rays(rays, g, "coords/out")
    assert np.abs(np.array(coords) - g["coords/matrices"]).max() < 1e-13
    fwd = tran.applyT(orig, coords)
    assert fwd[0] is orig[0]                      # opd is shared, as in the reference
    assert_rays(fwd, g, "coords/applyT_fwd", exact=False, tol=1e-13)
    inv = tran.applyT(rays, coords, inverse=True)
    assert_rays(inv, g, "coords/applyT_inv", exact=False, tol=1e-13)
    # applyT(orig) == sequential transforms, and the inverse returns to the start
    for a, b in zip(host(fwd)[1:], host(rays)[1:]):
        assert np.abs(a - b).max() < 1e-11
    for a, b in zip(host(inv)[1:], host(orig)[1:]):
        assert np.abs(a - b).max() < 1e-11
    pos = tran.applyTPos(orig[1], orig[2], orig[3], coords).cpu().numpy()
    assert np.abs(pos - g["coords/applyTPos_fwd"]).max() < 1e-10
    pos = tran.applyTPos(rays[1], rays[2], rays[3], coords, inverse=True).cpu().numpy()
    assert np.abs(pos - g["coords/applyTPos_inv"]).max() < 1e-10


@pytest.mark.parametrize("tag", ["mask", "index", "where"])
@pytest.mark.parametrize("as_torch", [False, True])
def test_ind_semantics(golden, tag, as_torch):
    import pyxfocus_amd.transformations as tran
    g = golden["wrappers"]
    rays = rays_from(g, "ind_%s/in" % tag)
    ind = g["ind_%s/ind" % tag]
    if as_torch:
        ind = torch.from_numpy(ind).cuda()
    if tag == "where":
        ind = (ind,)
    tran.transform(rays, 1., 2., 3., 0.1, 0.2, 0.3, ind=ind)
    tran.reflect(rays, ind=ind)
    tran.itransform(rays, -1., 0., 2., 0., 0.5, 0., ind=ind)
    assert_rays(rays, g, "ind_%s/out" % tag)


def test_vignette(golden):
    import pyxfocus_amd.transformations as tran
    g = golden["wrappers"]
    rays = rays_from(g, "vignette/in")
    keep = [r.clone() for r in rays]
    assert_rays(tran.vignette(rays), g, "vignette/default")
    assert_rays(tran.vignette(rays, ind=g["vignette/mask"]), g, "vignette/bymask")
    assert_rays(tran.vignette(rays, ind=torch.from_numpy(g["vignette/mask"]).cuda()), g, "vignette/bymask")
    assert_rays(tran.vignette(rays, ind=g["vignette/index"]), g, "vignette/byindex")
    assert_rays(tran.vignette(rays, ind=np.where(g["vignette/mask"])), g, "vignette/bymask")
    for a, b in zip(rays, keep):      # the input bundle is not modified
        assert torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(), b.nan_to_num())
    with pytest.raises(IndexError):
        tran.vignette(rays, ind=np.array([len(rays[0])]))


@pytest.mark.parametrize("n", [0, 1, 64, 2047, 2048, 2049, 1_000_003])
def test_vignette_sizes_and_order(n):
    """Compaction is stable and exact at tile boundaries, for empty / all / none kept."""
    import pyxfocus_amd.transformations as tran
    gen = np.random.default_rng(n)
    cols = [gen.normal(size=n) for _ in range(10)]
    for frac in (0., 0.31, 1.):
        keep = gen.random(n) < frac if 0 < frac < 1 else np.full(n, bool(frac))
        rays = [torch.from_numpy(c.copy()).cuda() for c in cols]
        out = tran.vignette(rays, ind=keep)
        for a, c in zip(out, cols):
            assert np.array_equal(a.cpu().numpy(), c[keep])


def test_vignette_default_predicate_large():
    import pyxfocus_amd.transformations as tran
    n = 3_000_001
    gen = torch.Generator(device="cuda").manual_seed(3)
    rays = [torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) for _ in range(10)]
    dead = torch.rand(n, device="cuda", generator=gen) < 0.4
    for k in (4, 5, 6):
        rays[k][dead] = 0.
    rays[6][17] = float("nan")
    out = tran.vignette(rays)
    mag = rays[4] ** 2 + rays[5] ** 2 + rays[6] ** 2
    ref = [r[mag > .1] for r in rays]
    for a, b in zip(out, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("tag", ["scalar", "array", "scalar_ind", "array_ind"])
def test_radgrat_dispatch(golden, tag):
    import pyxfocus_amd.transformations as tran
    g = golden["wrappers"]
    rays = rays_from(g, "radgrat_%s/in" % tag)
    wave = g["radgrat_%s/wave" % tag]
    wave = wave if wave.ndim else float(wave)
    ind = g["radgrat_%s/ind" % tag] if tag.endswith("ind") else None
    tran.radgrat(rays, 160. / 11832., 1., wave, ind=ind)
    assert_rays(rays, g, "radgrat_%s/out" % tag, exact=False, tol=1e-13)


def test_pointto(golden):
    import pyxfocus_amd.transformations as tran
    g = golden["wrappers"]
    rays = rays_from(g, "pointto/in")
    old_l = rays[4]
    tran.pointTo(rays, 1., -2., 5000., reverse=1.)
    assert rays[4] is not old_l                       # rebinding, as the reference does
    assert_rays(rays, g, "pointto/out")


def test_surfaces_wrappers_vs_oracle(port):
    """surfaces.* (3-line unpack-and-call wrappers, surfaces.py) against the oracle's native calls."""
    import pyxfocus_amd.surfaces as surf
    from oracle import cases
    rng = np.random.default_rng(21)
    c = cases.primary_input(rng, 5000, offaxis=1e-3)
    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    a = [c[k] for k in ALLCOLS[1:]]
    surf.wolterprimary(rays, 220., 8400.)
    port.woltsurf.wolterprimary(*a, 220., 8400., 1.)
    import pyxfocus_amd.transformations as tran
    tran.reflect(rays)
    port.transformationsf.reflect(*a[3:])
    surf.woltersecondary(rays, 220., 8400.)
    port.woltsurf.woltersecondary(*a, 220., 8400., 1.)
    tran.reflect(rays)
    port.transformationsf.reflect(*a[3:])
    surf.flat(rays, nr=1.)
    port.surfacesf.flatopd(*a, c["opd"], 1.)
    for k, r in zip(ALLCOLS, host(rays)):
        assert bits_equal(r, c[k]).all(), k
    # flat with ind: plain flat on the subset, OPD untouched (surfaces.py:17-24)
    msk = rng.random(5000) < .5
    c2 = {k: v.copy() for k, v in c.items()}
    c2["z"] += 3.
    rays = [torch.from_numpy(c2[k].copy()).cuda() for k in ALLCOLS]
    sub = [np.ascontiguousarray(c2[k][msk]) for k in ALLCOLS[1:]]
    port.surfacesf.flat(*sub)
    surf.flat(rays, ind=msk, nr=1.)
    for k, r, s in zip(ALLCOLS[1:], host(rays)[1:], sub):
        assert bits_equal(r[msk], s).all() and bits_equal(r[~msk], c2[k][~msk]).all()
    assert bits_equal(rays[0].cpu().numpy(), c2["opd"]).all()
    # wolterprimarynode = transform -> primary -> itransform
    c3 = cases.primary_input(rng, 1000)
    c3["x"] += 220.; c3["z"] -= 8400.                 # node-centred frame
    rays = [torch.from_numpy(c3[k].copy()).cuda() for k in ALLCOLS]
    a = [c3[k] for k in ALLCOLS[1:]]
    port.transformationsf.transform(*a, 220., 0., 8400., 0., 0., 0.)
    port.woltsurf.wolterprimary(*a, 220., 8400., 1.)
    port.transformationsf.itransform(*a, 220., 0., 8400., 0., 0., 0.)
    surf.wolterprimarynode(rays, 220., 8400.)
    for k, r in zip(ALLCOLS, host(rays)):
        assert bits_equal(r, c3[k]).all(), k


def test_conic_and_zernsurf_wrappers(port):
    import pyxfocus_amd.surfaces as surf
    from oracle import cases
    for name, fn, cols, args in cases.cases(num=777, seed=5):
        if fn == "surfacesf.conicopd" and name.endswith("hyper"):
            rays = [torch.from_numpy(cols[k].copy()).cuda() for k in ALLCOLS]
            surf.conic(rays, args[0], args[1], nr=args[2])
            cases.call(port, fn, cols, args)
            for k, r in zip(ALLCOLS, host(rays)):
                assert bits_equal(r, cols[k]).all(), k
        if name == "tracezern_36":
            rays = [torch.from_numpy(cols[k].copy()).cuda() for k in ALLCOLS]
            surf.zernsurf(rays, args[0], args[3], rorder=args[1], aorder=args[2])
            cases.call(port, fn, cols, args)
            for k, r in zip(ALLCOLS, host(rays)):
                f = ~np.isnan(cols[k])
                assert (np.isnan(r) == np.isnan(cols[k])).all()
                assert np.abs(r[f] - cols[k][f]).max() < 2e-11, k
            # default ordering == the Noll ordering the test cases use
            ro, ao = surf.zmodes(36)
            assert np.array_equal(ro, args[1]) and np.array_equal(ao, args[2])


def test_analyses_vs_numpy(golden):
    """analyses.* reductions (restated from analyses.py text; np.average formulas) on the golden
    focal-plane bundle and on weighted random data."""
    import pyxfocus_amd.analyses as anal
    g = golden["chain_config2"]
    rays = [torch.from_numpy(g["out/" + k].copy()).cuda() for k in ALLCOLS]
    x, y = g["out/x"], g["out/y"]
    assert abs(anal.rmsCentroid(rays) - float(g["rms_centroid"])) < 1e-18 + 1e-9 * float(g["rms_centroid"])
    cx, cy = anal.centroid(rays)
    assert abs(cx - np.average(x)) < 1e-15 and abs(cy - np.average(y)) < 1e-15
    r = np.sqrt((x - np.average(x)) ** 2 + (y - np.average(y)) ** 2)
    assert abs(anal.hpd(rays) - 2 * np.median(r)) < 1e-18
    rng = np.random.default_rng(9)
    n = 100_001
    c = {k: rng.normal(size=n) for k in ALLCOLS}
    c["n"] = -np.abs(c["n"]) - .5
    w = rng.random(n)
    rays = [torch.from_numpy(c[k]).cuda() for k in ALLCOLS]
    for wt in (None, w):
        x, y, l, m, nn = c["x"], c["y"], c["l"], c["m"], c["n"]
        av = lambda a: np.average(a, weights=wt)
        bx = av(x * l / nn) - av(x) * av(l / nn); ax = av((l / nn) ** 2) - av(l / nn) ** 2
        by = av(y * m / nn) - av(y) * av(m / nn); ay = av((m / nn) ** 2) - av(m / nn) ** 2
        assert abs(anal.analyticImagePlane(rays, weights=wt) - (-(bx + by) / (ax + ay))) < 1e-11
        assert abs(anal.analyticXPlane(rays, weights=wt) - (-bx / ax)) < 1e-11
        assert abs(anal.analyticYPlane(rays, weights=wt) - (-by / ay)) < 1e-11
        cx, cy = av(x), av(y)
        assert abs(anal.rmsCentroid(rays, weights=wt) - np.sqrt(av((x - cx) ** 2 + (y - cy) ** 2))) < 1e-12
        assert abs(anal.rmsX(rays, weights=wt) - np.sqrt(av((x - cx) ** 2))) < 1e-12
        assert abs(anal.rmsY(rays, weights=wt) - np.sqrt(av((y - cy) ** 2))) < 1e-12
    # weighted hpd (argsort + cdf, analyses.py:73-97)
    rr = np.sqrt((c["x"] - np.average(c["x"], weights=w)) ** 2 + (c["y"] - np.average(c["y"], weights=w)) ** 2)
    ind = np.argsort(rr); cdf = np.cumsum(w[ind]); cdf /= cdf.max(); rs = rr[ind]
    want = rs[np.argmin(np.abs(cdf - .75))] - rs[np.argmin(np.abs(cdf - .25))]
    assert abs(anal.hpd(rays, weights=w) - want) < 1e-12


def test_focusI_and_steer():
    """focusI brings an on-axis Wolter pair to best focus; steerX/Y zero the mean tilt."""
    import pyxfocus_amd.analyses as anal
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import conicsolve as con
    r0, z0 = 220., 8400.
    rays = sources.annulus(con.primrad(z0, r0, z0), con.primrad(z0 + 100., r0, z0), 200_000,
                           rng="philox", seed=5)
    tran.transform(rays, 0, 0, -(z0 + 200.), 0, 0, 0)
    surf.wolterprimary(rays, r0, z0)
    tran.reflect(rays)
    surf.woltersecondary(rays, r0, z0)
    tran.reflect(rays)
    tran.transform(rays, 0, 0, 25., 0, 0, 0)         # deliberately 25 mm off focus
    surf.flat(rays)
    blur = anal.rmsCentroid(rays)
    coords = tran.newCoords()
    dz = surf.focusI(rays, coords=coords)
    assert abs(dz + 25.) < 1e-3 and anal.rmsCentroid(rays) < 1e-4 < blur
    assert abs(coords[1][2, 3] + dz) < 1e-9
    tran.transform(rays, 0, 0, 0, 1e-3, -2e-3, 0)
    tran.steerX(rays); tran.steerY(rays)
    assert abs(float(rays[4].mean())) < 1e-6 and abs(float(rays[5].mean())) < 1e-6


def test_sources_numpy_path_on_device(golden):
    import pyxfocus_amd.sources as sources
    g = golden["sources"]
    np.random.seed(int(g["subannulus/seed"]))
    a = g["subannulus/args"]
    rays = sources.subannulus(a[0], a[1], a[2], int(a[3]))
    assert rays[1].is_cuda
    for k, r in zip(ALLCOLS, host(rays)):
        assert bits_equal(r, g["subannulus/" + k]).all()


def test_sources_philox_distribution_and_sharding():
    """Device RNG: same distribution as the reference formulas; a bundle generated in two shards
    (first_index offset) is bit-identical to the single bundle."""
    import pyxfocus_amd.sources as sources
    n = 2_000_000
    rin, rout = 220., 220.6537
    rays = sources.annulus(rin, rout, n, rng="philox", seed=42)
    x, y = rays[1], rays[2]
    r2 = x * x + y * y
    assert float(r2.min()) >= rin ** 2 * (1 - 1e-15) and float(r2.max()) <= rout ** 2 * (1 + 1e-15)
    u = ((r2 - rin ** 2) / (rout ** 2 - rin ** 2)).cpu().numpy()      # uniform in area
    assert abs(u.mean() - .5) < 2e-3 and abs(u.var() - 1 / 12.) < 2e-3
    th = torch.atan2(y, x).cpu().numpy()
    assert abs(th.mean()) < 5e-3 and abs(np.mean(np.cos(2 * th))) < 3e-3
    assert torch.all(rays[6] == -1.) and torch.all(rays[3] == 0.) and torch.all(rays[0] == 0.)
    a = sources.annulus(rin, rout, n // 2, rng="philox", seed=42, first_index=0)
    b = sources.annulus(rin, rout, n - n // 2, rng="philox", seed=42, first_index=n // 2)
    for full, p, q in zip(rays, a, b):
        assert torch.equal(full, torch.cat([p, q]))
    ps = sources.pointsource(0.01, 100_001, rng="philox", seed=1)
    mag = ps[4] ** 2 + ps[5] ** 2 + ps[6] ** 2
    assert float((mag - 1).abs().max()) < 4e-16
    assert float((ps[4] ** 2 + ps[5] ** 2).max()) <= np.sin(0.01) ** 2 * (1 + 1e-12)
    sa = sources.subannulus(100., 110., 0.5, 100_000, rng="philox", seed=2)
    ang = torch.atan2(sa[2], sa[1])
    assert float(ang.abs().max()) <= 0.25 + 1e-12 and torch.all(sa[6] == 1.)
    rb = sources.rectbeam(3., 4., 100_000, rng="philox", seed=3)
    assert float(rb[1].abs().max()) <= 3. and float(rb[2].abs().max()) <= 4.
    cb = sources.circularbeam(50., 100_000, rng="philox", seed=4)
    assert float((cb[1] ** 2 + cb[2] ** 2).max()) <= 2500. * (1 + 1e-12)


def test_cpu_tensors_rejected_loudly():
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    rays = sources.circularbeam(1., 8, device="cpu")
    with pytest.raises(TypeError):
        surf.flat(rays)


@pytest.mark.parametrize("n", [1, 2, 3, 64, 1001, 4096, 1_000_001, 20_000_000])
def test_hpd_radix_select_matches_np_median(n):
    """analyses.hpd = 2*median(rho) via on-device radix select (no sort): exact vs np.median for odd
    and even counts, clustered values (a focal spot: all radii share exponent and leading bits),
    duplicates, and NaN propagation."""
    import pyxfocus_amd.analyses as anal
    rng = np.random.default_rng(n)
    for kind in ("spot", "wide", "dups"):
        if kind == "spot":
            x, y = rng.normal(0, 7e-6, n), rng.normal(0, 7e-6, n)
        elif kind == "wide":
            x, y = rng.normal(0, 1, n) * 10. ** rng.integers(-8, 8, n), rng.normal(0, 1, n)
        else:
            x, y = rng.integers(-3, 4, n).astype(float), rng.integers(-3, 4, n).astype(float)
        rays = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(10)]
        rays[1], rays[2] = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        cx, cy = anal.centroid(rays)
        r = np.sqrt((x - cx) ** 2 + (y - cy) ** 2)
        got = anal.hpd(rays)
        assert got == np.median(r) * 2., (kind, n, got, np.median(r) * 2.)
        assert np.array_equal(anal.rho(rays, cent=True).cpu().numpy(), r)
    if n > 2:
        rays[1][n // 3] = float("nan")
        assert np.isnan(anal.hpd(rays))


def test_graze_and_ind_angle():
    import pyxfocus_amd.analyses as anal
    from oracle import cases
    rng = np.random.default_rng(3)
    c = cases._randrays(rng, 100_003)
    rays = [torch.from_numpy(c[k]).cuda() for k in ALLCOLS]
    dot = c["l"] * c["ux"] + c["m"] * c["uy"] + c["n"] * c["uz"]
    assert np.abs(anal.indAngle(rays).cpu().numpy() - np.arccos(dot)).max() < 1e-14
    assert np.abs(anal.grazeAngle(rays).cpu().numpy() - (np.pi / 2 - np.arccos(dot))).max() < 1e-14
    ind = rng.random(100_003) < .2
    assert np.abs(anal.grazeAngle(rays, ind=ind).cpu().numpy() - (np.pi / 2 - np.arccos(dot[ind]))).max() < 1e-14
    nrm = (0., 0., 1.)
    assert np.abs(anal.indAngle(rays, normal=nrm).cpu().numpy() - np.arccos(c["n"])).max() < 1e-14


def test_batch1_wrappers_vs_oracle(port):
    """surfaces.spoPrimary/spoSecondary/sphere/tanSphere/cyl/cylconic/paraxial*, transformations.grat/
    refract: wrapper-level calls against the oracle's native entry points."""
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from oracle import cases
    rng = np.random.default_rng(31)
    n = 20_001
    c = cases.annulus_rays(rng, n, 599.6, 601.2, 150.)
    c["l"] = rng.normal(0, 2e-4, n); c["m"] = rng.normal(0, 2e-4, n)
    c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    a = [c[k] for k in ALLCOLS[1:]]
    T, S = port.transformationsf, port.surfacesf
    ind = rng.random(n) < .5
    # SPO pair; secondary only on a subset (ind)
    surf.spoPrimary(rays, 600., 12000.)
    port.woltsurf.spocone(*a, 600., .25 * np.arctan((600. + .605 / 2) / 12000.))
    tran.reflect(rays); T.reflect(*a[3:])
    sub = [np.ascontiguousarray(v[ind]) for v in a]
    port.woltsurf.spocone(*sub, 600., .75 * np.arctan((600. + .605 / 2) / 12000.))
    for v, s_ in zip(a, sub):
        v[ind] = s_
    surf.spoSecondary(rays, 600., 12000., ind=ind)
    # linear grating: scalar and per-ray order
    tran.grat(rays, 160., 1., 2.4)
    T.grat(c["x"], c["y"], c["l"], c["m"], c["n"], 160., 1., 2.4)
    orders = rng.choice([-1., 0., 1., 2.], n)
    tran.grat(rays, 250., orders, 1.7)
    T.grat(c["x"], c["y"], c["l"], c["m"], c["n"], 250., orders, np.full(n, 1.7))
    surf.paraxial(rays, 5e3); S.paraxial(*a, 5e3)
    surf.paraxialY(rays, -7e3); S.paraxialy(*a, -7e3)
    # tanSphere = transform + sphere + transform (surfaces.py:91-102), OPD variant
    surf.tanSphere(rays, -800., nr=1.2)
    T.transform(*a, 0., 0., 800., 0., 0., 0.); S.tracesphereopd(c["opd"], *a, -800., 1.2); T.transform(*a, 0., 0., -800., 0., 0., 0.)
    surf.cyl(rays, 3000.); S.tracecyl(*a, 3000.)
    for k, r in zip(ALLCOLS, host(rays)):
        assert bits_equal(r, c[k]).all(), k
    tran.refract(rays, 1., 1.45); T.refract(*a[3:], 1., 1.45)
    for k, r in zip(ALLCOLS, host(rays)):
        f = ~np.isnan(c[k])
        assert (np.isnan(r) == np.isnan(c[k])).all() and np.abs(r[f] - c[k][f]).max() < 1e-14, k
    # cylconic
    for name, fn, cols, args in cases.cases(num=3000, seed=9):
        if name == "cylconic":
            rays = [torch.from_numpy(cols[k].copy()).cuda() for k in ALLCOLS]
            surf.cylconic(rays, *args)
            cases.call(port, fn, cols, args)
            for k, r in zip(ALLCOLS, host(rays)):
                assert bits_equal(r, cols[k]).all(), k


def test_batch2_wrappers_vs_oracle(port):
    """surfaces.woltersine / primaryLL / secondaryLL / ellipsoid* / wsPrimary / wsSecondary against the
    oracle's native entry points (wrapper glue: argument order, psi_eff, woltparam alpha, check=)."""
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import conicsolve as con
    from oracle import cases
    got = {}
    for name, fn, cols, args in cases.cases(num=1500, seed=3):
        if name not in ("woltersine", "wolterprimll", "woltersecll", "ellipsoidwoltll", "wsprimary", "wssecondary"):
            continue
        rays = [torch.from_numpy(cols[k].copy()).cuda() for k in ALLCOLS]
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        if name == "woltersine":
            surf.woltersine(rays, *args)
        elif name == "wolterprimll":
            surf.primaryLL(rays, *args)
        elif name == "woltersecll":
            surf.secondaryLL(rays, *args)
        elif name == "ellipsoidwoltll":
            r0, z0, psi, S, zmax, zmin, dphi, cf, ax, az = args
            surf.ellipsoidPrimaryLL(rays, r0, z0, S, psi, zmax, zmin, dphi, cf, ax, az)
        elif name == "wsprimary":
            fail = surf.wsPrimary(rays, cases.R0, cases.Z0, args[2], check=True)
            moved = (want["x"] != cols["x"]) | (want["y"] != cols["y"]) | (want["z"] != cols["z"])
            assert np.array_equal(fail.cpu().numpy(), ~moved)
            assert abs(con.woltparam(cases.R0, cases.Z0)[0] - args[0]) < 1e-18
        else:
            assert surf.wsSecondary(rays, cases.R0, cases.Z0, args[2]) is None
        tol = 2e-8 if name.startswith("ws") else 1e-10
        for k, r in zip(ALLCOLS, host(rays)):
            f = ~np.isnan(want[k])
            assert (np.isnan(r) == np.isnan(want[k])).all() and np.abs(r[f] - want[k][f]).max() < tol, (name, k)
        got[name] = True
    assert len(got) == 6
    # ellipsoidPrimary = transform -> conic -> itransform ; ellipsoidSecondary = woltersecondary(psi_eff)
    rng = np.random.default_rng(4)
    R0, F, S, psi = 220., 8400., 1e5, 1.
    c = cases.primary_input(rng, 4000)
    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    a = [c[k] for k in ALLCOLS[1:]]
    P, ea, eb, ee, ef = con.ellipsoidFunction(S, psi, R0, F)
    T = port.transformationsf
    T.transform(*a, 0., 0., -(F + ef - P - ea), 0., 0., 0.)
    port.surfacesf.conic(*a, eb ** 2 / ea, -ee ** 2)
    T.itransform(*a, 0., 0., -(F + ef - P - ea), 0., 0., 0.)
    T.reflect(*a[3:])
    psi_eff = np.arctan(R0 / P) / (np.arctan(R0 / F) - np.arctan(R0 / P))
    port.woltsurf.woltersecondary(*a, R0, F, psi_eff)
    surf.ellipsoidPrimary(rays, R0, F, S, psi)
    tran.reflect(rays)
    surf.ellipsoidSecondary(rays, R0, F, S, psi)
    for k, r in zip(ALLCOLS, host(rays)):
        assert bits_equal(r, c[k]).all(), k


def test_batch3_wrappers_vs_oracle(port):
    """surfaces.torus / conicplus / legSurf / zernphase / zernsurfrot / wsPrimaryB / wsSecondaryB."""
    import pyxfocus_amd.surfaces as surf
    from oracle import cases
    seen = set()
    for name, fn, cols, args in cases.cases(num=1200, seed=8):
        if name not in ("torus", "conicplus", "conicplusopd", "legsurf", "zernphase", "tracezernrot",
                        "wsprimaryback", "wssecondaryback"):
            continue
        rays = [torch.from_numpy(cols[k].copy()).cuda() for k in ALLCOLS]
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        if name == "torus":
            surf.torus(rays, *args)
        elif name == "conicplus":
            surf.conicplus(rays, *args)
        elif name == "conicplusopd":
            surf.conicplus(rays, args[0], args[1], args[2], nr=args[3])
        elif name == "legsurf":
            surf.legSurf(rays, *args)
        elif name == "zernphase":
            cf, ro, ao, rad, wave = args
            surf.zernphase(rays, cf, rad, wave, rorder=ro, aorder=ao)
        elif name == "tracezernrot":
            c1, r1, a1, c2, r2, a2, rad, rot = args
            surf.zernsurfrot(rays, c1, c2, rad, rot, rorder1=r1, aorder1=a1, rorder2=r2, aorder2=a2)
        elif name == "wsprimaryback":
            surf.wsPrimaryB(rays, cases.R0, cases.Z0, args[2], args[3])
        else:
            surf.wsSecondaryB(rays, cases.R0, cases.Z0, args[2], args[3])
        exact = name in ("torus", "conicplus", "conicplusopd", "legsurf")
        tol = 2e-8 if name.startswith("ws") else 2e-11
        for k, r in zip(ALLCOLS, host(rays)):
            if exact:
                assert bits_equal(r, want[k]).all(), (name, k)
            else:
                f = ~np.isnan(want[k])
                assert (np.isnan(r) == np.isnan(want[k])).all() and np.abs(r[f] - want[k][f]).max() < tol, (name, k)
        seen.add(name)
    assert len(seen) == 8


def test_lenses_and_oap_vs_oracle_composition(port):
    """lenses.singlet / doublet / singletCyl and surfaces.oapCollimate: the same op sequence through
    the oracle's native calls (lenses.py:7-91, surfaces.py:126-156)."""
    import pyxfocus_amd.lenses as lenses
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from oracle import cases
    rng = np.random.default_rng(12)
    n = 5000
    c = cases._randrays(rng, n)
    c["x"] = rng.uniform(-20, 20, n); c["y"] = rng.uniform(-20, 20, n); c["z"] = np.full(n, -5.)
    c["l"] *= .02; c["m"] *= .02
    c["n"] = np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    T, S = port.transformationsf, port.surfacesf

    def o_transform(a, dz):
        T.transform(*a, 0., 0., -dz, 0., 0., 0.)

    # doublet AC508-250
    w = {k: v.copy() for k, v in c.items()}
    a = [w[k] for k in ALLCOLS[1:]]
    r1, r2, r3, n1, n2, t1, t2 = 137.1, -111.7, 459.2, 1.5150885, 1.6437928, 7.5, 2.
    o_transform(a, r1); S.tracesphereopd(w["opd"], *a, r1, 1.); T.refract(*a[3:], 1., n1)
    o_transform(a, -r1 + t1 + r2); S.tracesphereopd(w["opd"], *a, r2, n1); T.refract(*a[3:], n1, n2)
    o_transform(a, -r2 + t2 - r3); S.tracesphereopd(w["opd"], *a, r3, n2); o_transform(a, r3)
    T.refract(*a[3:], n2, 1.); S.flatopd(*a, w["opd"], 1.)
    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    lenses.AC508_250(rays)
    for k, r in zip(ALLCOLS, host(rays)):
        f = ~np.isnan(w[k])
        assert (np.isnan(r) == np.isnan(w[k])).all() and np.abs(r[f] - w[k][f]).max() < 1e-11, k
    # the doublet focuses a collimated beam near its 250 mm back focal distance
    beam = sources.circularbeam(10., 20_000, rng="philox", seed=3)
    lenses.AC508_250(beam)
    dz = surf.focusI(beam)
    assert 200. < -dz < 300. or 200. < dz < 300.

    # cylindrical singlet with one flat side
    w = {k: v.copy() for k, v in c.items()}
    a = [w[k] for k in ALLCOLS[1:]]
    o_transform(a, 0.); S.flatopd(*a, w["opd"], 1.); T.refract(*a[3:], 1., 1.5150885)
    o_transform(a, 3.2 - 516.8); S.tracecylopd(w["opd"], *a, 516.8, 1.5150885); o_transform(a, 516.8)
    T.refract(*a[3:], 1.5150885, 1.); S.flatopd(*a, w["opd"], 1.)
    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    lenses.LJ1516_L2(rays)
    for k, r in zip(ALLCOLS, host(rays)):
        f = ~np.isnan(w[k])
        assert (np.isnan(r) == np.isnan(w[k])).all() and np.abs(r[f] - w[k][f]).max() < 1e-11, k

    # OAP: a point source at the focus (efl along -z from the nominal intersection) comes out
    # collimated along -y; and the composite equals the same sequence through the oracle
    ps = sources.pointsource(0.05, 50_000, rng="philox", seed=9)
    tran.transform(ps, 0, 0, 500., 0, 0, 0)              # origin -> OAP intersection, source at z = -efl
    w = {k: v.cpu().numpy().copy() for k, v in zip(ALLCOLS, ps)}
    out = surf.oapCollimate(ps, 500., np.pi / 2)
    assert out[0] is ps[0]
    l, m, n_ = [t.cpu().numpy() for t in out[4:7]]
    good = np.isfinite(m)
    assert good.mean() > .99 and np.abs(np.abs(m[good]) - 1.).max() < 1e-9 and np.abs(l[good]).max() < 1e-7
    a = [w[k] for k in ALLCOLS[1:]]
    fp = 500. * (1 + np.cos(np.pi / 2)) / 2
    T.transform(*a, 0., 0., 500., 0., 0., 0.)
    T.transform(*a, 0., 0., 0., -(np.pi - np.pi / 2), 0., 0.)
    T.transform(*a, 0., 0., fp, 0., 0., 0.)
    S.conic(*a, fp * 2, -1.)
    T.reflect(*a[3:])
    for k, r in zip(ALLCOLS[1:], host(ps)[1:]):          # ps itself was traced in place (local frame)
        assert bits_equal(r, w[k]).all(), k


def test_measure_opd_and_rms_point():
    import pyxfocus_amd.analyses as anal
    from oracle import cases
    rng = np.random.default_rng(6)
    c = cases._randrays(rng, 50_001)
    rays = [torch.from_numpy(c[k]).cuda() for k in ALLCOLS]
    pt = (1.5, -2.5, 30.)
    d = np.sqrt((c["x"] - pt[0]) ** 2 + (c["y"] - pt[1]) ** 2 + (c["z"] - pt[2]) ** 2)
    assert np.array_equal(anal.measureOPD(rays, pt).cpu().numpy(), d)
    ten = [0., 1.5, -2.5, 30., 0, 0, 0, 0, 0, 0]
    assert np.array_equal(anal.measureOPD(rays, ten).cpu().numpy(), d)
    assert abs(anal.rmsPoint(rays, pt) - np.sqrt(np.average(d ** 2))) < 1e-11
    w = rng.random(50_001)
    assert abs(anal.rmsPoint(rays, pt, weights=w) - np.sqrt(np.average(d ** 2, weights=w))) < 1e-11


def test_zernconic_extension():
    """surfaces.zernconic (extension, parity unpinned): (1) with zero coefficients it lands on the same
    surface as `conic`; (2) with coefficients it satisfies the surface equation and matches an
    independent fp64 NumPy Newton; (3) normals are unit and orthogonal to numerical tangents."""
    import pyxfocus_amd.surfaces as surf
    from oracle import cases
    rng = np.random.default_rng(77)
    n = 20_000
    R, K, rad = 2000., -.8, 60.
    c = cases._randrays(rng, n, nsign=-1.)
    c["x"] = rng.uniform(-40, 40, n); c["y"] = rng.uniform(-40, 40, n); c["z"] = rng.uniform(20, 30, n)
    c["l"] *= .05; c["m"] *= .05
    c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    ro, ao = surf.zmodes(21)
    # (1) zero coefficients == conic
    a = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    b = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    surf.conic(a, R, K)
    surf.zernconic(b, R, K, np.zeros(21), rad, rorder=ro, aorder=ao)
    for k in (1, 2, 3):
        assert float((a[k] - b[k]).abs().max()) < 1e-10
    # (2) with coefficients: independent Newton in NumPy using the oracle's zernset for S and its gradient
    coeff = rng.normal(0, 2e-3, 21); coeff[0] = 0.
    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    surf.zernconic(rays, R, K, coeff, rad, rorder=ro, aorder=ao, nr=1.)
    x, y, z = [rays[k].cpu().numpy() for k in (1, 2, 3)]
    from oracle import api
    port = api.load_port()
    idx = rng.choice(n, 300, replace=False)
    cc = 1. / R
    for i in idx:
        rho, th = math.hypot(x[i], y[i]), math.atan2(y[i], x[i])
        zv, _, _ = port.zernsurf.zernset(rho / rad, th, ro, ao)
        sag = cc * rho ** 2 / (1 + math.sqrt(1 - (K + 1) * cc ** 2 * rho ** 2)) + float(np.dot(coeff, zv))
        assert abs(z[i] - sag) < 1e-9, (i, z[i] - sag)
    # the ray stayed on its line
    t = (z - c["z"]) / c["n"]
    assert np.abs(c["x"] + c["l"] * t - x).max() < 1e-9 and np.abs(c["y"] + c["m"] * t - y).max() < 1e-9
    # OPD = path length * nr
    assert np.abs(rays[0].cpu().numpy() - c["opd"] - t).max() < 1e-9
    # (3) unit normals, mostly along +z
    ux, uy, uz = [rays[k].cpu().numpy() for k in (7, 8, 9)]
    assert np.abs(ux ** 2 + uy ** 2 + uz ** 2 - 1).max() < 1e-12 and uz.min() > .99


import math  # noqa: E402


def test_device_grid_and_extended_sources(golden):
    """The remaining sources.py generators on the device: grid kinds equal the reference's own
    sources.py output (positions bit for bit, direction cosines to an ulp of sin), random kinds have
    the reference's distribution."""
    import pyxfocus_amd.sources as sources
    g = golden["sources"]
    for tag, fn in (("xslit", lambda a: sources.xslit(a[0], a[1], int(a[2]), on_device=True)),
                    ("rectArray", lambda a: sources.rectArray(a[0], a[1], int(a[2]), on_device=True)),
                    ("fanBeam", lambda a: sources.fanBeam(a[0], a[1], int(a[2]), on_device=True)),
                    ("circFan", lambda a: sources.circFan(a[0], int(a[1]), int(a[2]), on_device=True))):
        rays = fn(g[tag + "/args"])
        for k, r in zip(ALLCOLS, host(rays)):
            w = g["%s/%s" % (tag, k)]
            assert r.shape == w.shape, (tag, k)
            if k in ("x", "y", "z"):
                assert bits_equal(r, w).all(), (tag, k)
            else:
                assert np.abs(r - w).max() < 1e-15, (tag, k)
    # a big grid (10^6 points): end points exact
    full = sources.rectArray(3., 4., 1000, on_device=True)
    assert full[1].shape[0] == 10 ** 6 and float(full[1].max()) == 3. and float(full[2].min()) == -4.
    n = 400_000
    gb = sources.gaussianBeam(0.01, n, rng="philox", seed=3)
    l, m = gb[4].cpu().numpy(), gb[5].cpu().numpy()
    sg = np.sin(0.01) / np.sqrt(2)
    assert abs(l.std() / sg - 1) < 5e-3 and abs(m.std() / sg - 1) < 5e-3 and abs(np.mean(l * m)) / sg ** 2 < 1e-2
    assert float((gb[4] ** 2 + gb[5] ** 2 + gb[6] ** 2 - 1).abs().max()) < 4e-16
    cb = sources.convergingbeam(8400., 220., 221., -.2, .3, n, 1.5, rng="philox", seed=4)
    x, y, z, l, m, nn = [t.cpu().numpy() for t in cb[1:7]]
    rho, th = np.hypot(x, y), np.arctan2(y, x)
    assert rho.min() >= 220. - 1e-9 and rho.max() <= 221. + 1e-9 and th.min() >= -.2 - 1e-12 and th.max() <= .3 + 1e-12
    assert np.all(z == 8400.) and np.abs(l ** 2 + m ** 2 + nn ** 2 - 1).max() < 1e-12
    # rays converge toward the axis: the median pointing error is the Lorentzian scale (1.5 arcsec)
    ang = np.arccos(-nn) - np.arctan(rho / 8400.)
    assert abs(np.median(np.abs(ang)) / (1.5 / 3600 * np.pi / 180) - 1) < 2e-2
    cb2 = sources.convergingbeam2(8400., 200., 220., -10., 10., n, 0., rng="philox", seed=5)
    x, y = cb2[1].cpu().numpy(), cb2[2].cpu().numpy()
    assert x.min() >= 200. and x.max() <= 220. and abs(y.mean()) < .05


def test_wavefront_southwellbin_and_reconstruct(golden):
    """reconstruct.f95 on the device.  Binning: same valid/invalid lenslets, means to 1e-12 (the
    sums are accumulated with atomics).  Reconstruction from the REFERENCE's binned slopes: the
    anti-diagonal Gauss-Seidel sweep performs the reference's updates bit for bit."""
    from oracle import cases
    from pyxfocus_amd import native
    g = golden["wavefront"]
    for case in cases.wavefront_cases():
        tag, xdim, ydim, bs = case[:4]
        x, y, l, m = [dev(v) for v in case[4:]]
        xang, yang, phase = native.southwellbin(x, y, l, m, bs, xdim, ydim)
        for name, t in (("bin_xang", xang), ("bin_yang", yang), ("bin_phase", phase)):
            got, want = t.cpu().numpy(), g["%s/%s" % (tag, name)]
            assert got.shape == want.shape == (xdim, ydim)
            assert np.array_equal(got == 100., want == 100.), (tag, name)
            assert np.abs(got - want).max() <= 1e-12 * max(1., np.abs(want[want != 100.]).max()), (tag, name)
        # analyses.wavefront's call: reconstruct(yang, xang, 1e-12, dx, phase)  (analyses.py:325)
        a, b, ph = [native.fgrid(g["%s/bin_%s" % (tag, k)]) for k in ("yang", "xang", "phase")]
        phasec, sweeps = native.reconstruct(a, b, 1e-12, bs, ph, 10000, return_sweeps=True)
        assert sweeps > 3
        for name, t in (("phasec", phasec), ("rec_xang", a), ("rec_yang", b), ("rec_phase", ph)):
            assert bits_equal(t.cpu().numpy(), g["%s/%s" % (tag, name)]).all(), (tag, name, sweeps)
    # rays outside the grid are dropped, not written out of bounds
    far = dev(np.array([1e9, -1e9, np.nan, 0.25]))
    one = dev(np.array([0., 0., 0., 1e-3]))
    xa, ya, ph = native.southwellbin(far, one * 0, one, one, 1.0, 8, 8)
    assert int((ph != 100.).sum()) == 1 and abs(float(xa[xa != 100.][0]) - np.tan(np.arcsin(1e-3))) < 1e-18


def test_wavefront_large_grid_paths():
    """Grids beyond the LDS-resident sizes: global-atomic binning (> 3072 bins) and the
    global-memory phase grid (> 160 KB), checked against the oracle on the same data."""
    from oracle import api
    from pyxfocus_amd import native
    port = api.load_port()
    rng = np.random.default_rng(5)
    xdim, ydim, bs, n = 150, 160, 0.5, 400_000
    x, y = rng.uniform(-36., 36., n), rng.uniform(-38., 38., n)
    keep = (x / 36.) ** 2 + (y / 38.) ** 2 < 1
    x, y = np.ascontiguousarray(x[keep]), np.ascontiguousarray(y[keep])
    l, m = 1e-4 * np.sin(x / 9.), 2e-4 * np.cos(y / 7.)
    wx, wy, wp = port.reconstruct.southwellbin(x, y, l, m, bs, xdim, ydim)
    gx, gy, gp = native.southwellbin(dev(x), dev(y), dev(l), dev(m), bs, xdim, ydim)
    assert np.array_equal(gp.cpu().numpy(), wp)
    assert np.abs(gx.cpu().numpy() - wx).max() < 1e-15 and np.abs(gy.cpu().numpy() - wy).max() < 1e-15
    a, b, ph = wy.copy(order="F"), wx.copy(order="F"), wp.copy(order="F")
    want = port.reconstruct.reconstruct(a, b, 1e-9, bs, ph, 300)
    ga, gb, gph = native.fgrid(wy), native.fgrid(wx), native.fgrid(wp)
    got, sweeps = native.reconstruct(ga, gb, 1e-9, bs, gph, 300, return_sweeps=True)
    assert sweeps == port.reconstruct.reconstruct.sweeps
    assert bits_equal(got.cpu().numpy(), want).all() and bits_equal(gph.cpu().numpy(), ph).all()


def test_hpdY_matches_numpy():
    import pyxfocus_amd.analyses as anal
    rng = np.random.default_rng(8)
    for n in (1001, 4000):
        cols = [dev(rng.normal(0., 1. + k, n)) for k in range(10)]
        y = cols[2].cpu().numpy()
        want = np.median(np.abs(y - np.average(y))) * 2.
        assert abs(anal.hpdY(cols) - want) < 1e-14
        w = rng.uniform(.5, 2., n)
        cy = np.average(y, weights=w)
        r = np.abs(y - cy)
        ind = np.argsort(r, kind="stable")
        cdf = np.cumsum(w[ind])
        cdf = cdf / cdf.max()
        want = r[ind][np.argmin(np.abs(cdf - .75))] - r[ind][np.argmin(np.abs(cdf - .25))]
        assert abs(anal.hpdY(cols, weights=w) - want) < 1e-12


def _ref_whpd(r, w):
    """The reference's weighted HPD on given radii (analyses.py:73-97)."""
    ind = np.argsort(r, kind="stable")
    rs, cdf = r[ind], np.cumsum(w[ind])
    cdf = cdf / cdf.max()
    return rs[np.argmin(np.abs(cdf - .75))] - rs[np.argmin(np.abs(cdf - .25))]


@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 65537, 1_000_003])
def test_weighted_hpd_by_weighted_radix_select(n):
    """Weighted HPD without a sort: equals the reference's argsort/cumsum/argmin construction on the
    same radii -- uniform, random and heavy-tailed weights, radii spanning many binades, and ties."""
    import pyxfocus_amd.analyses as anal
    rng = np.random.default_rng(n)
    for trial in range(3):
        x = rng.normal(0, 1., n) * np.exp(rng.normal(0, 2., n))
        y = rng.normal(0, 1., n)
        if trial == 2 and n > 3:                        # ties: many identical positions
            x[: n // 2] = x[0]
            y[: n // 2] = y[0]
        w = {0: np.ones(n), 1: rng.uniform(.1, 3., n), 2: rng.pareto(1.5, n) + 1e-3}[trial]
        rays = [dev(np.zeros(n)), dev(x), dev(y)] + [dev(np.zeros(n)) for _ in range(7)]
        cx, cy = anal.centroid(rays, weights=w)
        r = np.sqrt((x - cx) ** 2 + (y - cy) ** 2)
        got = anal.hpd(rays, weights=w)
        want = _ref_whpd(r, w)
        if trial == 2 and n > 3:
            # inside a run of equal radii the order of the reference's sort is arbitrary; what is
            # defined: each quantile is the tied group where the cdf crosses q, or its predecessor
            vals = anal._wquantiles(dev(r), dev(w), (.75, .25))
            uniq = np.unique(r)
            for v, q in zip(vals, (.75, .25)):
                assert v in uniq
                below, upto = w[r < v].sum() / w.sum(), w[r <= v].sum() / w.sum()
                nxt = uniq[np.searchsorted(uniq, v) + 1] if v < uniq[-1] else None
                crossing = below < q <= upto + 1e-12
                pred = nxt is not None and upto < q + 1e-12 and w[r <= nxt].sum() / w.sum() >= q - 1e-12
                assert crossing or pred, (n, q, v, below, upto)
            continue
        assert got == want, (n, trial, got, want)


def test_southwell_module_matches_reference_southwell_py(golden):
    """pyxfocus_amd.southwell.southwell == the reference's southwell.py over its compiled
    reconstruct.f95 (golden from oracle/make_golden.py): NaN handling, padding, sign, bit for bit."""
    from pyxfocus_amd import southwell as sw
    g = golden["design"]
    gx, gy = g["southwell/gx"].copy(), g["southwell/gy"].copy()
    out = sw.southwell(gx, gy, 1e-12, 1.).cpu().numpy()
    want = g["southwell/out"]
    assert out.shape == want.shape and np.array_equal(np.isnan(out), np.isnan(want))
    assert bits_equal(out, want).all()
    assert (gx[np.isnan(want)] == 100.).all()                 # in-place sentinel, as in the reference


@pytest.mark.parametrize("n", [(1 << 22) + 1, 10_000_000])
def test_one_pass_radial_median_is_exact(n):
    """hpd / hpdY on large bundles take the one-pass path (sample bracket + candidate select):
    exactly np.median's value; heavy ties make the bracket overflow and fall back, same answer;
    NaN anywhere -> NaN."""
    import pyxfocus_amd.analyses as anal
    rng = np.random.default_rng(n)
    x = rng.normal(0., 1., n) * np.exp(rng.normal(0., 1., n))
    y = rng.normal(.3, 2., n)
    rays = [dev(np.zeros(1))] * 1 + [dev(x), dev(y)] + [dev(np.zeros(1))] * 7
    cx, cy = anal.centroid(rays)
    want = 2. * np.median(np.sqrt((x - cx) ** 2 + (y - cy) ** 2))
    assert anal._radial_kth(rays, cx, cy, [(n - 1) // 2, n // 2]) is not None       # the fast path held
    assert anal.hpd(rays) == want
    assert anal.hpdY(rays) == 2. * np.median(np.abs(y - cy))
    # other ranks through the same entry (one bracket spans BOTH ranks, so they must be close:
    # far-apart ranks overflow the candidate buffer and report "did not hold"), reversed order too
    r = np.sort(np.sqrt((x - cx) ** 2 + (y - cy) ** 2))
    assert anal._radial_kth(rays, cx, cy, [3 * n // 4 + 2, 3 * n // 4]) == [r[3 * n // 4 + 2], r[3 * n // 4]]
    assert anal._radial_kth(rays, cx, cy, [n // 4]) == [r[n // 4]]
    assert anal._radial_kth(rays, cx, cy, [0]) == [r[0]] and anal._radial_kth(rays, cx, cy, [n - 1]) == [r[-1]]
    assert anal._radial_kth(rays, cx, cy, [3 * n // 4, n // 4]) is None
    # ties: half the rays on one point -> the bracket is one huge tied run
    x2, y2 = x.copy(), y.copy()
    x2[: n // 2 + 5] = 1.25
    y2[: n // 2 + 5] = -.5
    rays2 = [rays[0], dev(x2), dev(y2)] + rays[3:]
    cx2, cy2 = anal.centroid(rays2)
    assert anal.hpd(rays2) == 2. * np.median(np.sqrt((x2 - cx2) ** 2 + (y2 - cy2) ** 2))
    x2[7] = np.nan
    assert np.isnan(anal.hpd([rays[0], dev(x2), dev(y2)] + rays[3:]))


def test_catalogue_lenses_expand_to_the_reference_sequences():
    """The catalogue entries and the two hand-built recipes (edmundCollimator, cylNull) against the
    call sequences of lenses.py:161-246 written out with the reference's numbers."""
    from pyxfocus_amd import lenses
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(31)

    def beam():
        n = 2000
        c = [np.zeros(n) for _ in range(10)]
        c[1], c[2] = rng.uniform(-8, 8, n), rng.uniform(-8, 8, n)
        c[3] = np.full(n, -20.)
        c[4], c[5] = rng.normal(0, 1e-3, n), rng.normal(0, 1e-3, n)
        c[6] = np.sqrt(1 - c[4] ** 2 - c[5] ** 2)
        return [dev(v) for v in c]

    def same(a, b):
        for k, (u, v) in enumerate(zip(a, b)):
            assert bits_equal(u.cpu().numpy(), v.cpu().numpy()).all(), k

    def expect_edmund(r, reverse):
        if reverse:
            lenses.singlet(r, 1131.72, 780.87, 15.42, 1.51501)
            tran.transform(r, 0, 0, .1, 0, 0, 0)
            lenses.singlet(r, -779.37, 2704.01, 10.92, 1.64363)
        else:
            lenses.singlet(r, 2704.01, -779.37, 10.92, 1.64363)
            tran.transform(r, 0, 0, .1, 0, 0, 0)
            lenses.singlet(r, 780.87, 1131.72, 15.42, 1.51501)

    def expect_cylnull(r, reverse):
        if not reverse:
            tran.transform(r, 0, 0, 0, np.pi / 2, 0, 0)
            surf.cylconic(r, .007626, -.575)
            tran.refract(r, 1., 1.51501)
            tran.transform(r, 0, 0, 0, -np.pi / 2, 0, 0)
            tran.transform(r, 0, 0, 50, 0, 0, 0)
            surf.flat(r)
            tran.refract(r, 1.51501, 1.)
        else:
            tran.refract(r, 1., 1.51501)
            tran.transform(r, 0, 0, 50, 0, 0, 0)
            tran.transform(r, 0, 0, 0, np.pi / 2, 0, 0)
            surf.cylconic(r, -.007626, -.575)
            tran.refract(r, 1.51501, 1.)
            tran.transform(r, 0, 0, 0, -np.pi / 2, 0, 0)

    table = [(lenses.AC508_250, lambda r, rev: lenses.doublet(r, 137.1, -111.7, 459.2, 1.5150885, 1.6437928, 7.5, 2., reverse=rev)),
             (lenses.collimator6, lambda r, rev: lenses.singlet(r, 9324., 1124., 20., 1.5150885, reverse=rev)),
             (lenses.LJ1516_L2, lambda r, rev: lenses.singletCyl(r, 0, 516.8, 3.2, 1.5150885, reverse=rev)),
             (lenses.LJ1144_L2, lambda r, rev: lenses.singletCyl(r, 0, 258.4, 3.4, 1.5150885, reverse=rev)),
             (lenses.edmundCollimator, expect_edmund), (lenses.cylNull, expect_cylnull)]
    for fn, expect in table:
        for rev in (False, True):
            a = beam()
            b = [t.clone() for t in a]
            if fn is lenses.cylNull:                 # needs surface normals for its first refraction when reversed
                for r in (a, b):
                    surf.flat(r)
            fn(a, reverse=rev)
            expect(b, rev)
            same(a, b)
    assert lenses.AC508_250.__name__ == "AC508_250" and "AC508" in lenses.AC508_250.__doc__


@pytest.mark.parametrize("nmax", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_zernconic_every_order_lands_on_its_surface(nmax):
    """Each zernconic instantiation (orders 2, 4, 6, 8 unrolled; 10 by run-time loops): the traced
    points satisfy  z = conic sag + Zernike sum  (the sum from the oracle's zernset) and stay on their rays."""
    import pyxfocus_amd.surfaces as surf
    from oracle import api, cases
    port = api.load_port()
    rng = np.random.default_rng(300 + nmax)
    n = 3000
    R, K, rad = 2000., -.8, 60.
    c = cases._randrays(rng, n, nsign=-1.)
    c["x"] = rng.uniform(-40, 40, n); c["y"] = rng.uniform(-40, 40, n); c["z"] = rng.uniform(20, 30, n)
    c["l"] *= .05; c["m"] *= .05
    c["n"] = -np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2)
    nterms = (nmax + 1) * (nmax + 2) // 2
    ro, ao = cases.noll_orders(nterms)
    ro, ao = np.asarray(ro, np.int32), np.asarray(ao, np.int32)
    coeff = rng.normal(0, 2e-3, nterms) / (1 + ro)
    coeff[0] = 0.
    rays = [dev(c[k]) for k in ALLCOLS]
    surf.zernconic(rays, R, K, coeff, rad, rorder=ro, aorder=ao)
    x, y, z = [rays[k].cpu().numpy() for k in (1, 2, 3)]
    cc = 1. / R
    for i in rng.choice(n, 150, replace=False):
        rho, th = math.hypot(x[i], y[i]), math.atan2(y[i], x[i])
        zv, _, _ = port.zernsurf.zernset(rho / rad, th, ro, ao)
        sag = cc * rho ** 2 / (1 + math.sqrt(1 - (K + 1) * cc ** 2 * rho ** 2)) + float(np.dot(coeff, zv))
        assert abs(z[i] - sag) < 1e-9, (nmax, i, z[i] - sag)
    t = (z - c["z"]) / c["n"]
    assert np.abs(c["x"] + c["l"] * t - x).max() < 1e-9 and np.abs(c["y"] + c["m"] * t - y).max() < 1e-9
