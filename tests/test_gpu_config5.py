"""BASELINE configs 1 and 5 through the HIP path against the reference.

config 1: seed 0, pointsource(0.01, 1e5) -> transform -> flat -> reflect: per-column SHA-256 of what
          the reference's own sources.py + Fortran produced (tests/golden/config1.npz).
config 5: the Arcus-style multi-shell trace.  (a) examples/arcus_sector.py -- the call shapes of the
          reference's examples/arcus/sector.py: per-shell SPO pair with reflectivity weights, the
          stacked-grating while loop, focusY -- on pyxfocus_amd vs the fixture the SAME driver produced
          on the reference's own Python modules over its compiled Fortran (config5.npz), and vs the
          oracle run live at another seed / off-axis; (b) BASELINE's Wolter-I wording of config 5:
          the segmented two-bounce program + grating program (what bench.py times) vs the oracle's
          per-shell loop.
Parity bar: everything up to the first `radgrat` is bit-identical; from there positions carry
radgrat's 1e-14 on direction cosines times the ~12 m to the focal plane, asserted as RMS < 1e-10 mm."""
import hashlib
import math
import os
import sys

import numpy as np
import pytest

from util import ALLCOLS, bits_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def host(rays):
    return [r.cpu().numpy() for r in rays]


def test_config1_through_hip_path(golden):
    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    g = golden["config1"]
    np.random.seed(int(g["seed"]))
    rays = sources.pointsource(float(g["ang"]), int(g["num"]))
    assert all(r.is_cuda for r in rays)
    for k, v in zip(ALLCOLS, host(rays)):
        assert hashlib.sha256(v.tobytes()).hexdigest() == str(g["sha_in/" + k]), "source column " + k
    tran.transform(rays, 0, 0, -100, 0, 0, 0)
    surf.flat(rays)
    tran.reflect(rays)
    for k, v in zip(ALLCOLS, host(rays)):
        assert hashlib.sha256(v.tobytes()).hexdigest() == str(g["sha_out/" + k]), "output column " + k
        assert bits_equal(v[:64], g["head_out/" + k]).all()


def _rms_pos(got, want):
    f = np.isfinite(want[1]) & np.isfinite(want[2])
    d2 = sum((got[j][f] - want[j][f]) ** 2 for j in (1, 2, 3))
    return float(np.sqrt(d2.mean())), float(np.sqrt(d2.max()))


def _check_focal(out, want_rays, want_w, want_dz, want_rows, want_coords):
    got = host(out["rays"])
    assert out["grating_rows"] == want_rows
    for j in (1, 2, 3, 4, 5, 6):
        assert np.array_equal(np.isfinite(got[j]), np.isfinite(want_rays[j])), ALLCOLS[j]
    rms, mx = _rms_pos(got, want_rays)
    assert rms < 1e-10 and mx < 1e-9, (rms, mx)
    for j in (4, 5, 6, 7, 8, 9):
        f = np.isfinite(want_rays[j])
        assert np.abs(got[j][f] - want_rays[j][f]).max() < 1e-12, ALLCOLS[j]
    assert abs(out["dz"] - want_dz) < 1e-9 * abs(want_dz), (out["dz"], want_dz)
    assert np.abs(np.array(out["coords"]) - want_coords).max() < 1e-9 * abs(want_dz)
    w = out["weights"].cpu().numpy()
    assert np.abs(w - want_w).max() <= 1e-12 * np.abs(want_w).max()
    return rms, mx


def test_config5_arcus_driver_equals_reference_python(golden):
    import arcus_sector as A
    g = golden["config5"]
    np.random.seed(int(g["par/seed"]))
    out = A.traceArcus(A.Api.amd(), int(g["par/N"]), order=int(g["par/order"]), wave=float(g["par/wave"]),
                       shell_stride=int(g["par/shell_stride"]), modules_per_row=int(g["par/modules_per_row"]))
    want = [g["out/" + k] for k in ALLCOLS]
    rms, mx = _check_focal(out, want, g["weights"], float(g["dz"]), int(g["grating_rows"]), g["coords"])
    print("config5 (reference-python fixture): rms position error %.3e mm, max %.3e mm" % (rms, mx))
    import pyxfocus_amd.analyses as anal
    cx, cy = anal.centroid(out["rays"], weights=out["weights"])
    assert abs(cx - g["centroid_w"][0]) < 1e-9 and abs(cy - g["centroid_w"][1]) < 1e-9
    assert abs(anal.hpd(out["rays"], weights=out["weights"]) - float(g["hpd_w"])) < 1e-9
    assert abs(anal.rmsY(out["rays"], weights=out["weights"]) - float(g["rmsY_w"])) < 1e-9


def test_config5_spo_stage_bit_identical_to_oracle(port):
    """Everything before the gratings (sources, frame rotation, SPO primary/secondary, off-axis kick,
    reflections, collection, move to the grating plane, flat) is bit-identical; the reflectivity
    weights (acos, exp from ocml) agree to 1e-13."""
    import arcus_sector as A
    from oracle import wrappers
    for seed, off in ((3, (0., 0.)), (4, (1.5e-4, -1e-4))):
        kw = dict(offX=off[0], offY=off[1], rows=(0, 3, 7), modules_per_row=2, shell_stride=11)
        np.random.seed(seed)
        a = A.defineSPOaperture(A.Api.amd(), 24, **kw)
        np.random.seed(seed)
        b = A.defineSPOaperture(A.Api(*wrappers.bind(port), np), 24, **kw)
        for k, u, v in zip(ALLCOLS, host(a[0]), b[0]):
            assert bits_equal(u, v).all(), (seed, k)
        assert np.abs(a[1].cpu().numpy() - b[1]).max() <= 1e-13 * np.abs(b[1]).max()
        assert a[2] == b[2] and a[3] == b[3] and np.array_equal(np.array(a[4]), np.array(b[4]))


def test_config5_arcus_driver_equals_live_oracle_offaxis(port):
    import arcus_sector as A
    from oracle import wrappers
    kw = dict(order=-3, wave=2.4, offX=1e-4, offY=-5e-5, rows=(1, 4, 6), modules_per_row=1, shell_stride=5)
    np.random.seed(11)
    out = A.traceArcus(A.Api.amd(), 40, **kw)
    np.random.seed(11)
    ref = A.traceArcus(A.Api(*wrappers.bind(port), np), 40, **kw)
    rms, mx = _check_focal(out, ref["rays"], ref["weights"], ref["dz"], ref["grating_rows"], np.array(ref["coords"]))
    print("config5 (live oracle, off-axis): rms position error %.3e mm, max %.3e mm" % (rms, mx))


def test_config5_segmented_wolter_programs_equal_oracle_loop(port):
    """BASELINE's wording of config 5 (multi-shell WOLTER-I + grating) the way bench.py runs it: ONE
    segmented two-bounce program over all shells (per-shell constants staged in LDS) + ONE grating
    program, against the oracle looping over the shells with the reference's call chain."""
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    from oracle import cases, wrappers
    osrc, osurf, otran, oanal = wrappers.bind(port)
    F, lmir, ZG, GRAZE = 12e3, 50., 8000., 1.5 * math.pi / 180
    radii = list(np.arange(320.443, 376.243, .775)[::9]) + list(np.arange(755.807, 811.607, .775)[::13])
    shells = [(float(r0), math.sqrt(F * F - r0 * r0), (-1) ** i * .0931 * (i % 5), 50. / r0) for i, r0 in enumerate(radii)]
    per = 498                                   # even; shell boundaries fall inside 512-ray workgroup tiles
    n = per * len(shells)
    rng = np.random.default_rng(8)
    cols = {k: np.zeros(n) for k in ALLCOLS}
    for s, (r0, z0, a, w) in enumerate(shells):          # sources.subannulus(rin, rout, w, per, zhat=-1) per shell
        rin, rout = cases.primrad(z0, r0, z0), cases.primrad(z0 + lmir, r0, z0)
        rho = np.sqrt(rin ** 2 + rng.random(per) * (rout ** 2 - rin ** 2))
        th = rng.random(per) * w - w / 2
        sl = slice(s * per, (s + 1) * per)
        cols["x"][sl], cols["y"][sl], cols["n"][sl] = rho * np.cos(th), rho * np.sin(th), -1.
    buf = torch.from_numpy(np.stack([cols[k] for k in ALLCOLS])).cuda()
    rays = [buf[i] for i in range(10)]
    ty, rx = -ZG * math.sin(GRAZE), math.pi / 2 - GRAZE

    def shell(t, sl, r0, z0, a):
        t.transform(sl, 0, 0, -(z0 + 2 * lmir), 0, 0, a)
        (surf if t is tran else osurf).wolterprimary(sl, r0, z0)
        t.reflect(sl)
        (surf if t is tran else osurf).woltersecondary(sl, r0, z0)
        t.reflect(sl)

    def grating(t, s, r):
        t.transform(r, 0, ty, 0, rx, 0, 0)
        s.flat(r)
        t.reflect(r)
        t.radgrat(r, 160. / ZG, 1., 2.4)
        t.itransform(r, 0, ty, 0, rx, 0, 0)
        s.flat(r)

    prog = pxf.compile_segments(rays, per, lambda sl, r0, z0, a: shell(tran, sl, r0, z0, a),
                                [(g[0], g[1], g[2]) for g in shells])
    prog.run()
    want = [cols[k] for k in ALLCOLS]
    for s, (r0, z0, a, w) in enumerate(shells):
        shell(otran, [c[s * per:(s + 1) * per] for c in want], r0, z0, a)
    for k, u, v in zip(ALLCOLS, host(rays), want):
        assert bits_equal(u, v).all(), "two-bounce stage, column " + k       # bit-identical before the grating
    with pxf.fused() as rec:
        grating(tran, surf, rays)
    assert rec.launches == 1
    grating(otran, osurf, want)
    got = host(rays)
    for j in range(1, 10):
        assert np.array_equal(np.isfinite(got[j]), np.isfinite(want[j])), ALLCOLS[j]
    rms, mx = _rms_pos(got, want)
    assert np.isfinite(want[1]).mean() > .9 and rms < 1e-10 and mx < 1e-9, (rms, mx)
    print("config5 (segmented Wolter-I + grating vs oracle loop): rms position error %.3e mm, max %.3e mm" % (rms, mx))


@pytest.mark.parametrize("fast", [False, True])
def test_config5_spo_stage_segmented_equals_shell_loop(fast):
    """The petal's SPO stage as THREE launches (segmented source / segmented SPO-pair program with the two
    graze-angle columns written in-kernel / move + flat) against the reference-shaped Python loop over
    shells (defineSPOaperture) on the same np.random stream: bit-identical rays in the default mode,
    reflectivity weights to 1e-13; in the opt-in tolerance mode positions to 1e-9 mm (cone intersections
    are closed-form: no Newton jitter, only rounding)."""
    import arcus_sector as A
    kw = dict(rows=(0, 2, 5), modules_per_row=2, shell_stride=7)
    N = 180
    np.random.seed(21)
    a = A.defineSPOaperture(A.Api.amd(), N, **kw)
    np.random.seed(21)
    b = A.defineSPOaperture_segmented(N, source="numpy", fast=fast, **kw)
    assert b[5].fast == fast
    for k, u, v in zip(ALLCOLS, host(a[0]), host(b[0])):
        if fast:
            assert np.array_equal(np.isfinite(u), np.isfinite(v)), k
            assert np.abs(u - v).max() <= 1e-9 * (1 + np.abs(u).max()), (k, np.abs(u - v).max())
        else:
            assert bits_equal(u, v).all(), k
    wa, wb = a[1].cpu().numpy(), b[1].cpu().numpy()
    assert np.abs(wa - wb).max() <= 1e-13 * np.abs(wa).max()
    assert a[2] == b[2] and a[3] == b[3] and np.array_equal(np.array(a[4]), np.array(b[4]))
    # device Philox source: the same distribution, shell by shell (mean radius of every shell)
    c = A.defineSPOaperture_segmented(N, source="philox", seed=3, **kw)
    assert np.isfinite(c[0][1].cpu().numpy()).all() and abs(float(c[1].sum()) / float(b[1].sum()) - 1) < 1e-3


def test_config5_fan_array_equals_host_loop_and_reference_fixture(golden):
    """grating.fanArray (every ray walks the fanned grating array in registers: two launches) against
    (a) the reference's stacked-grating `while` loop through the drop-in API (gratArray of
    examples/arcus_sector.py): bit-identical rays, same row count, same coords, same focusY shift;
    (b) the fixture the loop produced on the reference's own Python modules (config5.npz)."""
    import arcus_sector as A
    from pyxfocus_amd import grating
    g = golden["config5"]
    kw = dict(order=int(g["par/order"]), wave=float(g["par/wave"]), shell_stride=int(g["par/shell_stride"]),
              modules_per_row=int(g["par/modules_per_row"]))
    np.random.seed(int(g["par/seed"]))
    a = A.traceArcus(A.Api.amd(), int(g["par/N"]), **kw)
    np.random.seed(int(g["par/seed"]))
    b = A.traceArcus(A.Api.amd(), int(g["par/N"]), fan=grating.fanArray, **kw)
    assert a["grating_rows"] == b["grating_rows"] == int(g["grating_rows"]) and a["dz"] == b["dz"]
    assert np.array_equal(np.array(a["coords"]), np.array(b["coords"]))
    for k, u, v in zip(ALLCOLS, host(a["rays"]), host(b["rays"])):
        assert bits_equal(u, v).all(), k
    _check_focal(b, [g["out/" + k] for k in ALLCOLS], g["weights"], float(g["dz"]), int(g["grating_rows"]), g["coords"])
    # a ray that can never reach a grating is reported, not looped on forever
    rays = [t.clone() for t in b["rays"]]
    rays[1].fill_(0.); rays[2].fill_(0.); rays[4].fill_(0.); rays[5].fill_(0.); rays[6].fill_(1.)
    with pytest.raises(RuntimeError, match="never reached a grating"):
        grating.fanArray(rays, 1e4, 95., 1e-4, .016, -3, 2.4, maxiter=50)
