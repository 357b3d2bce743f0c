"""GPU parity tests: HIP kernels (through the C ABI) vs the oracle / the reference goldens.

Bar (DESIGN.md "Parity policy"):
  * flat, flatopd, conic, conicopd, wolterprimary[opd], woltersecondary, transform, itransform,
    reflect: BIT-IDENTICAL to the reference (NaN == NaN).
  * radgrat, radgratW: |err| <= 1e-14 + 1e-13|ref| on the direction cosines.
  * tracezern[OPD]: |err| <= 2e-11 + 1e-12|ref| (positions in mm, cosines) -- well inside the
    north-star's 1e-10 mm RMS ray-position budget.
"""
import numpy as np
import pytest

from oracle import cases
from util import ALLCOLS, BITWISE, assert_cols, bits_equal, call_gpu, to_dev, to_host, zernike_order_check

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _golden_cases(g):
    for ent in g["__names__"]:
        name, fn = str(ent).split("|")
        cols = {k: g["%s/in/%s" % (name, k)].copy() for k in ALLCOLS}
        args, i = [], 0
        while "%s/arg/%d" % (name, i) in g:
            a = g["%s/arg/%d" % (name, i)]
            args.append(a if a.ndim else a.item())
            i += 1
        want = {k: g["%s/out/%s" % (name, k)] for k in cases.SIG[fn].split()}
        yield name, fn, cols, tuple(args), want


def test_library_loaded_on_gpu():
    from pyxfocus_amd import _lib
    assert torch.cuda.is_available()
    assert _lib.lib().pxf_abi_version() == 1


def test_golden_native_cases(golden):
    """Every committed reference vector (N=256 per case)."""
    for name, fn, cols, args, want in _golden_cases(golden["native_cases"]):
        t = to_dev(cols)
        call_gpu(fn, t, args)
        got = to_host(t)
        assert_cols(fn, got, want, want.keys(), context="golden:" + name)
        # columns the function does not take must be untouched
        for k in set(ALLCOLS) - set(cases.SIG[fn].split()):
            assert bits_equal(got[k], cols[k]).all()


@pytest.mark.parametrize("num", [5, 6, 63, 2049, 100003])
def test_live_oracle_sizes(port, num):
    """Same seeded inputs through oracle and GPU at ragged sizes (odd counts exercise the
    vector-pair tail, 1 and 2 the degenerate grids)."""
    for name, fn, cols, args in cases.cases(num=num, seed=4242 + num):
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        t = to_dev(cols)
        call_gpu(fn, t, args)
        assert_cols(fn, to_host(t), want, ALLCOLS, context="n=%d:%s" % (num, name))


@pytest.mark.parametrize("num", [1, 2, 3])
def test_degenerate_sizes(port, num):
    from pyxfocus_amd import native
    rng = np.random.default_rng(num)
    c = cases._randrays(rng, num)
    t = to_dev(c)
    a = [c[k] for k in ALLCOLS[1:]]
    d = [t[k] for k in ALLCOLS[1:]]
    for be, cols in ((port, a), (native, d)):
        be.transformationsf.transform(*cols, 1., 2., 3., .1, .2, .3)
        be.surfacesf.flat(*cols)
        be.transformationsf.reflect(*cols[3:])
        be.surfacesf.conic(*cols, 1000., -1.)
    got = to_host(t)
    for k in ALLCOLS:
        assert bits_equal(got[k], c[k]).all(), k


def test_empty_bundle():
    from pyxfocus_amd import native
    e = [torch.empty(0, dtype=torch.float64, device="cuda") for _ in range(10)]
    native.surfacesf.flat(*e[1:])
    native.woltsurf.wolterprimary(*e[1:], 220., 8400., 1.)
    native.transformationsf.transform(*e[1:], 1., 2., 3., .1, .2, .3)
    native.transformationsf.reflect(*e[4:])


def test_masked_equals_gather_scatter(port):
    """`mask` (the wrappers' ind=) == gather -> oracle -> scatter; unselected rays untouched bitwise."""
    rng = np.random.default_rng(7)
    for name, fn, cols, args in cases.cases(num=4097, seed=31):
        n = len(cols["x"])
        msk = rng.random(n) < 0.37
        want = {k: v.copy() for k, v in cols.items()}
        sub = {k: np.ascontiguousarray(v[msk]) for k, v in cols.items()}
        a2 = tuple(a[msk] if (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.shape == (n,)) else a
                   for a in args)
        cases.call(port, fn, sub, a2)
        for k in ALLCOLS:
            want[k][msk] = sub[k]
        t = to_dev(cols)
        call_gpu(fn, t, args, mask=torch.from_numpy(msk).cuda())
        got = to_host(t)
        assert_cols(fn, got, want, ALLCOLS, context="masked:" + name)
        for k in ALLCOLS:
            assert bits_equal(got[k][~msk], cols[k][~msk]).all(), "unselected rays modified: %s %s" % (name, k)


def test_unaligned_views(port):
    """Columns that are views at an odd element offset (8-byte aligned only) take the scalar path."""
    for name, fn, cols, args in cases.cases(num=1500, seed=77):
        if fn not in BITWISE:
            continue
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        big = {k: torch.zeros(len(v) + 3, dtype=torch.float64, device="cuda") for k, v in cols.items()}
        t = {}
        for k, v in cols.items():
            t[k] = big[k][1:1 + len(v)]
            t[k].copy_(torch.from_numpy(v))
            assert t[k].data_ptr() % 16 == 8
        call_gpu(fn, t, args)
        assert_cols(fn, to_host(t), want, ALLCOLS, context="unaligned:" + name)
        for k in big:   # guard elements around the view untouched
            assert float(big[k][0]) == 0. and float(big[k][-1]) == 0. and float(big[k][-2]) == 0.


def test_chain_config2_golden_and_1e6(port, golden):
    """golden bundle bitwise, then 1e6 rays vs the oracle: RMS position error
    (the north-star metric, target < 1e-10 mm) must be EXACTLY 0 because the chain is bit-identical."""
    from pyxfocus_amd import native
    g = golden["chain_config2"]
    t = to_dev({k: g["in/" + k] for k in ALLCOLS})
    cases.chain_config2(native, t)
    got = to_host(t)
    for k in ALLCOLS:
        assert bits_equal(got[k], g["out/" + k]).all(), k
    rng = np.random.default_rng(11)
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    c = cases.annulus_rays(rng, 1_000_000, rin, rout, 0.)
    t = to_dev(c)
    cases.chain_config2(port, c)
    cases.chain_config2(native, t)
    got = to_host(t)
    rms = np.sqrt(np.mean((got["x"] - c["x"]) ** 2 + (got["y"] - c["y"]) ** 2 + (got["z"] - c["z"]) ** 2))
    assert rms == 0.0
    for k in ALLCOLS:
        assert bits_equal(got[k], c[k]).all(), k


def test_offaxis_chain_vs_oracle(port):
    """5 arcmin off-axis: secondary needs 3-8 Newton iterations, divergent per ray."""
    from pyxfocus_amd import native
    rng = np.random.default_rng(12)
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    c = cases.annulus_rays(rng, 200_001, rin, rout, 0.)
    off = 5. / 60 * np.pi / 180
    c["l"][:] = np.sin(off)
    c["n"][:] = -np.sqrt(1 - np.sin(off) ** 2)
    t = to_dev(c)
    cases.chain_config2(port, c)
    cases.chain_config2(native, t)
    got = to_host(t)
    for k in ALLCOLS:
        assert bits_equal(got[k], c[k]).all(), k


def test_transform_roundtrip_property():
    """Size-independent property at a larger N: itransform(transform(rays)) == rays to rounding, |d|=1."""
    from pyxfocus_amd import native
    n = 5_000_000
    gen = torch.Generator(device="cuda").manual_seed(1)
    cols = [torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1 for _ in range(9)]
    orig = [c.clone() for c in cols]
    p = (1.5, -2.5, 3.5, 0.3, -0.2, 0.9)
    native.transformationsf.transform(*cols, *p)
    native.transformationsf.itransform(*cols, *p)
    for a, b in zip(cols, orig):
        assert float((a - b).abs().max()) < 5e-15


def test_full_size_1e9_rays_properties(port):
    """BASELINE config 3 at its full size: 1e9 rays on one GPU (80 GB bundle), one launch per call.
    Size-independent properties + the first 1e5 rays against the oracle bit for bit (the device
    source is keyed on the global ray index, so a 1e5-ray bundle with the same seed IS the head of
    the 1e9-ray bundle)."""
    import math
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import conicsolve as con
    free, total = torch.cuda.mem_get_info()
    n = 1_000_000_000
    if free < 100e9:
        pytest.skip("needs ~90 GB of free HBM")
    r0, z0, L = cases.R0, cases.Z0, cases.LMIR
    rin, rout = float(con.primrad(z0, r0, z0)), float(con.primrad(z0 + L, r0, z0))
    zg, g = 8000., 1.5 * math.pi / 180

    def trace(rays):
        tran.transform(rays, 0, 0, -(z0 + 2 * L), 0, 0, 0)
        surf.wolterprimary(rays, r0, z0)
        tran.reflect(rays)
        surf.woltersecondary(rays, r0, z0)
        tran.reflect(rays)
        tran.transform(rays, 0, -zg * math.sin(g), 0, math.pi / 2 - g, 0, 0)
        surf.flat(rays)
        tran.reflect(rays)
        tran.radgrat(rays, 160. / zg, 1., 2.4)
        tran.itransform(rays, 0, -zg * math.sin(g), 0, math.pi / 2 - g, 0, 0)
        surf.flat(rays)

    rays = sources.subannulus(rin, rout, 0.5, n, zhat=-1., rng="philox", seed=77)
    assert rays[0].shape[0] == n
    trace(rays)
    head = [r[:100_000].cpu().numpy() for r in rays]
    tail = [r[-100_000:].cpu().numpy() for r in rays]
    # properties over ALL 1e9 rays
    assert bool((rays[3] == 0).all())                                   # on the focal plane
    mag = rays[4] * rays[4] + rays[5] * rays[5] + rays[6] * rays[6]
    assert float((mag - 1).abs().max()) < 1e-12                         # unit direction cosines
    assert bool(torch.isfinite(rays[1]).all()) and bool(torch.isfinite(rays[6]).all())
    assert bool((rays[9] == 1).all()) and bool((rays[7] == 0).all())    # flat's normal
    del rays, mag
    torch.cuda.empty_cache()
    # the same first / last 1e5 global indices as small bundles, through the oracle
    for first, got in ((0, head), (n - 100_000, tail)):
        small = sources.subannulus(rin, rout, 0.5, 100_000, zhat=-1., rng="philox", seed=77, first_index=first)
        c = {k: v.cpu().numpy() for k, v in zip(ALLCOLS, small)}
        a = [c[k] for k in ALLCOLS[1:]]
        T = port.transformationsf
        T.transform(*a, 0., 0., z0 + 2 * L, 0., 0., 0.)
        port.woltsurf.wolterprimary(*a, r0, z0, 1.)
        T.reflect(*a[3:])
        port.woltsurf.woltersecondary(*a, r0, z0, 1.)
        T.reflect(*a[3:])
        T.transform(*a, 0., zg * math.sin(g), 0., -(math.pi / 2 - g), 0., 0.)
        port.surfacesf.flat(*a)
        T.reflect(*a[3:])
        xs, ys, ls, ms, ns = (c[k] for k in ("x", "y", "l", "m", "n"))
        T.radgrat(xs, ys, ls, ms, ns, 2.4, 160. / zg, 1.)
        T.itransform(*a, 0., zg * math.sin(g), 0., -(math.pi / 2 - g), 0., 0.)
        port.surfacesf.flat(*a)
        for k, u in zip(ALLCOLS, got):
            # everything up to radgrat is bit-identical; radgrat is 1e-14-level, then propagated ~8 m
            assert np.abs(u - c[k]).max() < 5e-10, (k, np.abs(u - c[k]).max())


def test_full_size_1e9_rays_zernike_properties(port):
    """BASELINE config 4 at its full size: 1e9 rays, the 36-term Zernike surface as ONE program
    (zernsurf -> reflect -> flat).  Size-independent properties over all rays, a strided sample checked
    against an independent evaluation of the surface polynomial (torch, from pxf_zernike_monomials), and the
    first / last 1e5 global ray indices against the oracle."""
    import ctypes as C
    import math
    import pyxfocus_amd as pxf
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from pyxfocus_amd import _lib
    free, total = torch.cuda.mem_get_info()
    n = 1_000_000_000
    if free < 100e9:
        pytest.skip("needs ~90 GB of free HBM")
    rng = np.random.default_rng(4)
    coeff = rng.normal(0., 1e-3, 36)
    coeff[0] = 0.
    ro, ao = cases.noll_orders(36)
    ro, ao = np.ascontiguousarray(ro, np.int32), np.ascontiguousarray(ao, np.int32)

    def prep(rays):
        rays[3].fill_(10.)
        rays[4].fill_(.01)
        rays[6].fill_(-math.sqrt(1 - 1e-4))

    rays = sources.circularbeam(50., n, rng="philox", seed=41)
    prep(rays)
    surf.zernsurf(rays, coeff, 50., rorder=ro, aorder=ao)
    # (1) every ray sits on the surface z = S(x, y): independent Horner evaluation on a strided 1e6-ray sample
    mono = np.zeros(36)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    assert _lib.lib().pxf_zernike_monomials(coeff.ctypes.data_as(dp), ro.ctypes.data_as(ip), ao.ctypes.data_as(ip), 36, 7,
                                            mono.ctypes.data_as(dp)) == 0
    xs, ys, zs = [rays[j][::1000] for j in (1, 2, 3)]
    u, v = xs / 50., ys / 50.
    sv = torch.zeros_like(u)
    N = 7
    for j in range(N, -1, -1):
        off = j * (N + 1) - j * (j - 1) // 2
        p = torch.full_like(u, float(mono[off]))
        for i in range(1, N - j + 1):
            p = p * u + float(mono[off + i])
        sv = sv * v + p
    assert float((zs - sv).abs().max()) < 1e-12
    nrm = rays[7] * rays[7] + rays[8] * rays[8] + rays[9] * rays[9]
    assert float((nrm - 1).abs().max()) < 1e-12 and bool(torch.isfinite(rays[1]).all())
    del nrm, sv, p, u, v
    head = [r[:100_000].cpu().numpy() for r in rays]
    tail = [r[-100_000:].cpu().numpy() for r in rays]
    # (2) the whole config-4 chain as one program == zernsurf (above) + reflect + flat, launch by launch
    tran.reflect(rays)
    surf.flat(rays)
    keep_x, keep_l = rays[1].clone(), rays[4].clone()
    del rays
    torch.cuda.empty_cache()
    rays = sources.circularbeam(50., n, rng="philox", seed=41)
    prep(rays)
    with pxf.fused() as rec:
        surf.zernsurf(rays, coeff, 50., rorder=ro, aorder=ao)
        tran.reflect(rays)
        surf.flat(rays)
    assert rec.launches == 1
    assert float((rays[1] - keep_x).abs().max()) < 1e-12 and float((rays[4] - keep_l).abs().max()) < 1e-14
    assert bool((rays[3] == 0).all())
    del rays, keep_x, keep_l
    torch.cuda.empty_cache()
    # (3) first / last 1e5 global indices through the oracle
    for first, got in ((0, head), (n - 100_000, tail)):
        small = sources.circularbeam(50., 100_000, rng="philox", seed=41, first_index=first)
        prep(small)
        c = {k: t.cpu().numpy() for k, t in zip(ALLCOLS, small)}
        port.zernsurf.tracezern(*[c[k] for k in ALLCOLS[1:]], coeff, ro, ao, 50.)
        for k, u_ in zip(ALLCOLS, got):
            assert np.abs(u_ - c[k]).max() <= 2e-11 + 1e-12 * np.abs(c[k]).max(), (k, np.abs(u_ - c[k]).max())


def test_streams_and_graph_capture(port):
    """Every launch goes to torch's CURRENT stream (no implicit sync, nothing allocated by the
    library), so a trace can run on a side stream and be captured into a HIP graph and replayed."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(41)
    rin, rout = cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR)
    c = cases.annulus_rays(rng, 300_000, rin, rout, 0.)
    want = {k: v.copy() for k, v in c.items()}
    cases.chain_config2(port, want)

    def trace(r):
        tran.transform(r, 0, 0, -(cases.Z0 + 2 * cases.LMIR), 0, 0, 0)
        surf.wolterprimary(r, cases.R0, cases.Z0)
        tran.reflect(r)
        surf.woltersecondary(r, cases.R0, cases.Z0)
        tran.reflect(r)
        surf.flat(r)

    # side stream
    side = torch.cuda.Stream()
    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        trace(rays)
    side.synchronize()
    for k, r in zip(ALLCOLS, rays):
        assert bits_equal(r.cpu().numpy(), want[k]).all(), k
    # graph capture: eager calls and a fused program, replayed on fresh inputs
    static = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            tran.transform(static, 0, 0, -(cases.Z0 + 2 * cases.LMIR), 0, 0, 0)
            with pxf.fused():
                surf.wolterprimary(static, cases.R0, cases.Z0)
                tran.reflect(static)
                surf.woltersecondary(static, cases.R0, cases.Z0)
                tran.reflect(static)
            surf.flat(static)
    torch.cuda.current_stream().wait_stream(cap)
    for rep in range(2):
        for k, t in zip(ALLCOLS, static):
            t.copy_(torch.from_numpy(c[k]))
        g.replay()
        torch.cuda.synchronize()
        for k, r in zip(ALLCOLS, static):
            assert bits_equal(r.cpu().numpy(), want[k]).all(), (rep, k)


def test_argument_validation_through_python_layer():
    from pyxfocus_amd import _lib, native
    import pyxfocus_amd.surfaces as surf
    rays = [torch.zeros(16, dtype=torch.float64, device="cuda") for _ in range(10)]
    with pytest.raises(ValueError, match="differ in length"):
        native.surfacesf.flat(*rays[1:9], torch.zeros(15, dtype=torch.float64, device="cuda"))
    with pytest.raises(TypeError, match="mask"):
        native.surfacesf.flat(*rays[1:], mask=torch.zeros(16, dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        native.surfacesf.flat(*rays[1:], mask=torch.zeros(15, dtype=torch.bool, device="cuda"))
    with pytest.raises(_lib.PxfError, match="range"):
        surf.zernsurf(rays, [1., 1.], 50., rorder=[2, 40], aorder=[0, 0])
    with pytest.raises(_lib.PxfError, match="invalid"):
        surf.zernsurf(rays, [1.], 50., rorder=[2], aorder=[1])           # n - |m| odd: not a Zernike term
    with pytest.raises(IndexError):
        import pyxfocus_amd.transformations as tran
        tran.reflect(rays, ind=np.array([99]))


def test_two_host_threads_two_streams(port):
    """The C ABI is stateless and stream-ordered: two host threads tracing different bundles on
    their own HIP streams at the same time both get the oracle's bits."""
    import threading
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    results, errors = {}, []

    def work(tid):
        try:
            rng = np.random.default_rng(100 + tid)
            c = cases.annulus_rays(rng, 20000 + 1000 * tid, cases.primrad(cases.Z0),
                                   cases.primrad(cases.Z0 + cases.LMIR), 0.)
            want = {k: v.copy() for k, v in c.items()}
            cases.chain_config2(port, want)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rep in range(10):
                    rays = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
                    tran.transform(rays, 0, 0, -(cases.Z0 + 2 * cases.LMIR), 0, 0, 0)
                    surf.wolterprimary(rays, cases.R0, cases.Z0)
                    tran.reflect(rays)
                    surf.woltersecondary(rays, cases.R0, cases.Z0)
                    tran.reflect(rays)
                    surf.flat(rays)
                    st.synchronize()
                    ok = all(bits_equal(r.cpu().numpy(), want[k]).all() for k, r in zip(ALLCOLS, rays))
                    results[(tid, rep)] = ok
        except Exception as e:  # surfaced in the main thread
            errors.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(results) == 20 and all(results.values())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_special_values_bitwise(port, seed):
    """IEEE corner inputs through every bit-exact entry: +-0, denormals, huge and tiny magnitudes,
    +-inf and NaN sprinkled over the ray columns.  The GPU must reproduce the CPU oracle bit for bit
    (sign of zero included; NaN payload aside) -- no flush-to-zero, no fast-math, the reference's
    operation order.  Newton surfaces included: where a corner value never converges both sides stop
    at the same iteration cap (the reference itself would not terminate) with the same bits."""
    rng = np.random.default_rng(900 + seed)
    special = np.array([0., -0., 5e-324, -5e-324, 2.2250738585072014e-308, 1e-300, -1e-300, 1e300, -1e300,
                        np.inf, -np.inf, np.nan, 1., -1., .5, 2., 1e-160, 1e160])
    n_checked = 0
    for name, fn, cols, args in cases.cases(num=512, seed=77 + seed):
        if fn not in BITWISE:
            continue
        cols = {k: v.copy() for k, v in cols.items()}
        for k in ALLCOLS:
            hit = rng.random(cols[k].shape[0]) < .15
            cols[k][hit] = rng.choice(special, int(hit.sum()))
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        t = to_dev(cols)
        call_gpu(fn, t, args)
        got = to_host(t)
        for k in ALLCOLS:
            assert bits_equal(got[k], want[k]).all(), (name, fn, k, int((~bits_equal(got[k], want[k])).sum()))
        n_checked += 1
    assert n_checked >= 25


@pytest.mark.parametrize("kt,ka", [(0, 0), (1, 6), (3, 3), (3, 21), (4, 2), (7, 7), (8, 10), (10, 10),
                                   (11, 19), (11, 20), (15, 15), (16, 5), (20, 11)])
def test_legendre_legendre_order_ranges(port, kt, ka):
    """The Legendre-Legendre kernels are instantiated for azimuthal orders <= 3, <= 7, <= 11, <= 15 (register-
    resident P_j, zero-padded matrix read through scalar loads; at most 32 padded axial rows and 256 cells) and
    for any order (matrix in LDS, regenerated recurrence): every variant against the oracle, with dense random
    coefficient sets up to the given orders (10 x 10 is what the reference's examples fit).  Orders stay <= 21:
    the reference evaluates P_n by an explicit factorial sum (specialFunctions.f95:337-388) that loses digits
    with n (5e-12 mm at order 20, 1e-10 at 30 for these amplitudes) and whose default-INTEGER `2**n` overflows
    at n = 31 (sign flip; division by zero from 32 on) -- the kernels' three-term recurrences do neither, so
    beyond ~order 25 the two part ways and it is the reference that is wrong (tests/soak/ll_debug.py)."""
    rng = np.random.default_rng(1000 * kt + ka)
    base = {name: (fn, cols, args) for name, fn, cols, args in cases.cases(num=300, seed=99)}
    nc = (ka + 1) * (kt + 1)
    ax, az = np.meshgrid(np.arange(ka + 1), np.arange(kt + 1), indexing="ij")
    coeff = rng.normal(0., 2e-5, nc) / (1 + ax.flatten() + az.flatten())
    keep = rng.random(nc) < .8
    keep[-1] = True                                    # the highest orders stay in
    coeff, ax, az = coeff[keep], ax.flatten()[keep].astype(np.int32), az.flatten()[keep].astype(np.int32)
    for name in ("wolterprimll", "woltersecll", "ellipsoidwoltll"):
        fn, cols, args = base[name]
        args = tuple(args[:-3]) + (coeff, ax, az)
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        t = to_dev(cols)
        call_gpu(fn, t, args)
        assert_cols(fn, to_host(t), want, ALLCOLS, context="kt=%d ka=%d:%s" % (kt, ka, name))


@pytest.mark.parametrize("nmax", list(range(1, 16)))
def test_zernike_every_radial_order(port, nmax):
    """One kernel instantiation per maximum radial order 1..15: each against the oracle, for tracezern,
    tracezernOPD, zernphase and the fused-program variant (util.zernike_order_check).  (Regression:
    several unrolled instantiations of orders 9..15 once returned wrong values)"""
    zernike_order_check(port, nmax)


@pytest.mark.parametrize("nmax", list(range(1, 16)))
def test_zernike_every_radial_order_masked(port, nmax):
    """The MASKED instantiations (ind=) are separate code objects: the same sweep with 40 % of the
    rays selected; unselected rays must come back untouched bit for bit."""
    zernike_order_check(port, nmax, masked=True)


def test_nontemporal_instantiations_in_child_process():
    """Every >3.3M-ray launch uses the NT=true instantiation of a kernel (non-temporal loads/stores),
    a different code object from the one small test bundles reach.  use_nt() reads PXF_NT_MIN_RAYS once
    per process, so ONE child process with PXF_NT_MIN_RAYS=0 re-runs the oracle comparisons on the NT
    kernels: all cases (plain + masked) and the Zernike sweep over all 15 orders (plain + masked)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PXF_NT_MIN_RAYS="0")
    r = subprocess.run([sys.executable, os.path.join(here, "nt_sweep.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "nt sweep ok" in r.stdout, r.stdout[-2000:]


def test_tolerance_mirrors_rms_position_parity(port):
    """Parity of the tolerance-parity mirrors the way the north-star states it: RMS ray-position error
    vs the oracle < 1e-10 mm -- observed 0 (woltersine, the Legendre-Legendre family) and 1-4e-12 (all four W-S
    surfaces, no ray further than 4e-11).  The W-S SECONDARIES are the delicate ones, and the cause is the
    reference's own conditioning: `a` in woltsurf.f95:518-519 contains (1 - cos beta) with cos beta ~ 1 - 3e-4, so F
    is QUANTISED in steps of 2.6e-9 mm (one ulp of cos beta) and the converged point equals the reference's only
    if the whole Newton path does, bit for bit -- which is why the kernels restate the C library's sincos / tan and
    form the correctly rounded atan2 (pxf_libm_match.h; round 1 sat at 2e-9 RMS, round 2 at 1.2e-10 with the
    correctly rounded cosine, 2.1e-11 with libm's `cos()`, 1.0e-12 with the `sincos` build the reference binary
    really calls).  Recorded per surface: RMS, median, share of rays further than 1e-10 mm."""
    stats = {}
    for name, fn, cols, args in cases.cases(num=50000, seed=4321):
        if not any(t in fn for t in ("ws", "ll", "woltersine")):
            continue
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        t = to_dev(cols)
        call_gpu(fn, t, args)
        got = to_host(t)
        moved = (want["x"] != cols["x"]) | (want["y"] != cols["y"]) | (want["z"] != cols["z"])
        assert np.array_equal(moved, (got["x"] != cols["x"]) | (got["y"] != cols["y"]) | (got["z"] != cols["z"])), name
        d = np.sqrt(sum((got[k][moved] - want[k][moved]) ** 2 for k in "xyz"))
        stats[name] = (float(np.sqrt((d ** 2).mean())), float(np.median(d)), float((d > 1e-10).mean()), float(d.max()))
        print("%-16s rms %.2e  median %.2e  share>1e-10 %.4f  max %.2e mm" % ((name,) + stats[name]))
    assert len(stats) == 8
    for name, (rms, med, frac, mx) in stats.items():
        if name.startswith("wssecondary"):
            assert rms < 2e-11 and med < 1e-12 and frac < 2e-4 and mx < 2e-8, (name, rms, med, frac, mx)
        else:
            assert rms < 1e-10 and frac == 0., (name, rms, med, frac, mx)


def test_legsurf_and_conicplus_parameter_ranges(port):
    """The host-folded tables of legSurf (orders 0..12, up to 48 terms) and conicplus (1..16 even
    polynomial coefficients) over their whole supported ranges: bit for bit against the oracle, and the
    documented refusals beyond them."""
    from pyxfocus_amd import _lib, native
    base = {name: (fn, cols, args) for name, fn, cols, args in cases.cases(num=400, seed=5)}
    rng = np.random.default_rng(12)
    fn, cols, args = base["legsurf"]
    for nc, omax in ((1, 0), (3, 1), (10, 4), (25, 8), (48, 12)):
        coeff = rng.normal(0, .1, nc)
        xo = rng.integers(0, omax + 1, nc).astype(np.int32)
        yo = rng.integers(0, omax + 1, nc).astype(np.int32)
        xo[-1] = omax
        a = tuple(args[:3]) + (coeff, xo, yo)
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, a)
        t = to_dev(cols)
        call_gpu(fn, t, a)
        got = to_host(t)
        for k in ALLCOLS:
            assert bits_equal(got[k], want[k]).all(), ("legsurf", nc, omax, k)
    t = to_dev(cols)
    with pytest.raises(_lib.PxfError):
        call_gpu(fn, t, tuple(args[:3]) + (np.ones(49), np.zeros(49, np.int32), np.zeros(49, np.int32)))
    with pytest.raises(_lib.PxfError):
        call_gpu(fn, t, tuple(args[:3]) + (np.ones(2), np.array([13, 0], np.int32), np.zeros(2, np.int32)))
    fn, cols, args = base["conicplus"]
    for npc in (1, 2, 5, 9, 16):
        p = rng.normal(0, 1., npc) * 10. ** (-6. - 4. * np.arange(npc))
        a = tuple(args[:2]) + (p,)
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, a)
        t = to_dev(cols)
        call_gpu(fn, t, a)
        got = to_host(t)
        for k in ALLCOLS:
            assert bits_equal(got[k], want[k]).all(), ("conicplus", npc, k)
    with pytest.raises(_lib.PxfError):
        call_gpu(fn, to_dev(cols), tuple(args[:2]) + (np.ones(17) * 1e-30,))


@pytest.mark.parametrize("fn", ["transformationsf.transform", "transformationsf.itransform"])
def test_zero_angle_frame_changes_special_values_bitwise(port, fn):
    """A frame change with zero angles skips its rotations for every vector made of ordinary numbers (and stores only
    what changed); a vector holding a -0, an infinity or a NaN must still go through the reference's arithmetic
    (1*x - 0*y ...: the -0 may come back +0, the NaN reaches its neighbours).  Pure translations, +0 and -0 angles,
    rotations about one axis only -- bundles salted with special values in every column -- stand-alone and inside a
    fused program, bit for bit against the oracle."""
    import pyxfocus_amd as pxf
    from pyxfocus_amd import native
    rng = np.random.default_rng(5)
    n = 4096
    cols = {k: rng.normal(0., 1., n) for k in ALLCOLS}
    specials = [0., -0., np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e300, -1e-300]
    for j, k in enumerate(ALLCOLS[1:]):
        idx = rng.choice(n, 400, replace=False)
        cols[k][idx] = rng.choice(specials, 400)
        cols[k][j::97] = 0.                                    # plenty of +0 (the normals of a fresh source)
    nine = ALLCOLS[1:]
    for args in ((0., 0., -8600., 0., 0., 0.), (1.5, -2.5, 0., -0., 0., -0.), (0., 0., 0., 0., 0., 0.),
                 (0., 0., 3., .01, 0., 0.), (0., 0., 0., 0., 0., .3), (-0., 0., 7., 0., -.2, 0.)):
        want = {k: v.copy() for k, v in cols.items()}
        getattr(port.transformationsf, fn.split(".")[1])(*[want[k] for k in nine], *args)
        for fused in (False, True):
            t = to_dev(cols)
            if fused:
                with pxf.fused():
                    getattr(native.transformationsf, fn.split(".")[1])(*[t[k] for k in nine], *args)
                    native.transformationsf.reflect(*[t[k] for k in "l m n ux uy uz".split()])
                w2 = {k: v.copy() for k, v in want.items()}
                port.transformationsf.reflect(*[w2[k] for k in "l m n ux uy uz".split()])
            else:
                getattr(native.transformationsf, fn.split(".")[1])(*[t[k] for k in nine], *args)
                w2 = want
            got = to_host(t)
            for k in ALLCOLS:
                assert bits_equal(got[k], w2[k]).all(), (fn, args, fused, k, int((~bits_equal(got[k], w2[k])).sum()))


def test_frame_angles_where_libm_sincos_differs_from_sin_and_cos(port):
    """The reference binary forms the cos / sin of a frame angle with ONE call of the C library's `sincos` (both
    flang and gcc merge the pair), and glibc's sincos -- built without FMA -- differs from its sin() / cos() in the
    last bit for about one argument in a thousand.  The host side of the kernels therefore calls sincos too
    (pxf_kernel.h: host_sincos).  Regression (found by a 600-seed soak of the fuzz test): frame changes and SPO cones
    at angles picked so that sincos != (sin, cos) -- bit for bit against the oracle."""
    import ctypes as C
    from pyxfocus_amd import native
    libm = C.CDLL("libm.so.6")
    for f in (libm.sin, libm.cos):
        f.restype, f.argtypes = C.c_double, [C.c_double]
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(251)
    odd = [-0.1610095633704616]                       # the soak's angle
    while len(odd) < 12:
        a = float(rng.uniform(-3.1, 3.1)) if len(odd) % 2 else float(rng.uniform(-.02, .02))
        s, c = C.c_double(), C.c_double()
        libm.sincos(a, C.byref(s), C.byref(c))
        if s.value != libm.sin(a) or c.value != libm.cos(a):
            odd.append(a)
    base = {name: (fn, cols, args) for name, fn, cols, args in cases.cases(num=2048, seed=9)}
    nine = ALLCOLS[1:]
    for i in range(0, 12, 3):
        ang = odd[i:i + 3]
        for fn in ("transform", "itransform"):
            cols = base["transform_general"][1]
            want = {k: v.copy() for k, v in cols.items()}
            getattr(port.transformationsf, fn)(*[want[k] for k in nine], 1.5, -2.5, 30., *ang)
            t = to_dev(cols)
            getattr(native.transformationsf, fn)(*[t[k] for k in nine], 1.5, -2.5, 30., *ang)
            got = to_host(t)
            for k in ALLCOLS:
                assert bits_equal(got[k], want[k]).all(), (fn, ang, k)
    fn, cols, args = base["spocone_prim"]
    for tg in [a for a in odd if abs(a) < .03][:3] + [abs(odd[1]) % .02 + .002]:
        want = {k: v.copy() for k, v in cols.items()}
        port.woltsurf.spocone(*[want[k] for k in nine], args[0], tg)
        t = to_dev(cols)
        native.woltsurf.spocone(*[t[k] for k in nine], args[0], tg)
        got = to_host(t)
        for k in ALLCOLS:
            assert bits_equal(got[k], want[k]).all(), ("spocone", tg, k)


def test_box_libm_is_the_one_the_kernels_restate():
    """The W-S kernels restate glibc 2.35's x86-64 sincos / tan / atan2 (pxf_libm_match.h) and the frame changes take
    their host sin/cos from `sincos`; the oracle the GPU tests compare with runs on THIS box's C library.  If the box
    carries another libm (glibc picks these routines per CPU and per version), say THAT -- instead of letting the W-S
    parity tests fail a quantum off with no hint why.  (The same checks run as CPU tests in tests/test_host.py; this
    runs them on the GPU box, once per suite.)"""
    import json
    import os
    import platform
    import test_host
    prov = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "PROVENANCE.json")))
    here = " ".join(platform.libc_ver())
    try:
        test_host.test_ws_secondary_cos_beta_matches_libm_bitwise()
        test_host.test_ws_back_sincos_atan2_matches_libm_bitwise()
    except AssertionError as e:
        pytest.fail("this box's C library (%s) does not compute sincos/tan/atan2 like the one the kernels restate and the "
                    "fixtures were made with (%s on %s): Wolter-Schwarzschild parity (tests below) is expected to be a "
                    "quantum (5e-9 mm) off for a few rays -- %s" % (here, prov["libc"], prov["cpu"], str(e)[:300]))


def test_shared_reciprocal_divisions_equal_ieee_division_bitwise():
    """csrc/pxf_exactdiv.h: the quotients of a surface normal over its length, y/rho and x/rho of a grating, the two
    roots of a cone over 2A ... share ONE refined reciprocal inside a window of ordinary operands and fall back to `/`
    outside it.  The library's own self-test draws 2^31 operand tuples (random mantissas; exponents across the window,
    around both of its edges, over the whole range; zeros, denormals, infinities, NaNs) and compares every routine
    with `/` as bit patterns: not one result may differ."""
    import torch
    from pyxfocus_amd import _lib, native
    h = _lib.lib()
    dev = torch.device("cuda", 0)
    out = torch.zeros(1, dtype=torch.int64, device=dev)
    total = 0
    for seed, n in ((1, 1 << 29), (0xdecafbad, 1 << 30), (2 ** 63 + 12345, 1 << 29)):
        _lib.check(h.pxf_selftest_exact_division(n, seed, out.data_ptr(), native._stream(dev)), "pxf_selftest_exact_division")
        assert int(out.item()) == 0, (seed, int(out.item()))
        total += n
    assert total == 1 << 31
