"""Randomised programs: sequences of the bit-exact operations with random parameters and random
`ind` selections, run (a) one launch per call on the GPU, (b) as one fused program on the GPU and
(c) call by call through the oracle on the CPU.  All three must agree BIT FOR BIT, including the
in-band failure sentinels (zeroed cosines, NaN) that random geometry produces in quantity."""
import math

import numpy as np
import pytest

from oracle import cases
from util import ALLCOLS, bits_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def random_program(rng, n):
    """List of (name, kwargs) steps; every step knows how to run on (surf, tran) and on the oracle."""
    steps = []
    nops = rng.integers(3, 11)
    for _ in range(nops):
        kind = rng.choice(["transform", "itransform", "flat", "flatopd", "reflect", "conic", "conicopd", "wolterprimary",
                           "woltersecondary", "spocone", "sphere", "cyl", "paraxial", "grat", "cylconic_skip"])
        ind = None
        if rng.random() < .35:
            ind = rng.random(n) < rng.uniform(.2, .8)
        t = np.concatenate([rng.uniform(-5, 5, 3), rng.uniform(-.3, .3, 3)])
        # exact zeros (either sign) in the frame parameters: pure translations, rotations about one axis, the
        # identity -- the forms whose zero-angle rotations the kernels skip for ordinary rays and whose unchanged
        # columns they do not rewrite, here on bundles full of failure sentinels (NaN, zeroed cosines, -0)
        if rng.random() < .5:
            z = rng.random(6) < rng.choice([.35, .7, 1.])
            t[z] = rng.choice([0., -0.], int(z.sum()))
        t = tuple(float(v) for v in t)
        par = {"transform": t, "itransform": t, "flat": (), "flatopd": (rng.uniform(1, 2),), "reflect": (),
               "conic": (rng.choice([-1, 1]) * rng.uniform(300, 3000), rng.choice([-1., 0., -2.5, .5])),
               "conicopd": (rng.choice([-1, 1]) * rng.uniform(300, 3000), rng.choice([-1., 0., -2.5, .5]), 1.5),
               "wolterprimary": (rng.uniform(100, 300), rng.uniform(5000, 9000), rng.choice([1., 2.])),
               "woltersecondary": (rng.uniform(100, 300), rng.uniform(5000, 9000), 1.),
               "spocone": (rng.uniform(100, 700), rng.uniform(.002, .02)),
               "sphere": (rng.uniform(200, 2000),), "cyl": (rng.uniform(200, 2000),),
               "paraxial": (rng.uniform(100, 1000),), "grat": (rng.uniform(100, 300), float(rng.integers(-2, 3)), rng.uniform(.5, 5.)),
               "cylconic_skip": ()}[kind]
        if kind == "cylconic_skip":
            continue
        if kind in ("conic", "conicopd", "wolterprimary", "woltersecondary", "sphere", "cyl", "paraxial") and ind is not None:
            ind = None                       # these reference wrappers have no ind= argument
        steps.append((kind, par, ind))
    return steps


def run_gpu(steps, rays, surf, tran):
    for kind, p, ind in steps:
        if kind == "transform":
            tran.transform(rays, *p, ind=ind)
        elif kind == "itransform":
            tran.itransform(rays, *p, ind=ind)
        elif kind == "flat":
            surf.flat(rays, ind=ind)
        elif kind == "flatopd":
            surf.flat(rays, nr=p[0]) if ind is None else surf.flat(rays, ind=ind)
        elif kind == "reflect":
            tran.reflect(rays, ind=ind)
        elif kind == "conic":
            surf.conic(rays, *p)
        elif kind == "conicopd":
            surf.conic(rays, p[0], p[1], nr=p[2])
        elif kind == "wolterprimary":
            surf.wolterprimary(rays, *p)
        elif kind == "woltersecondary":
            surf.woltersecondary(rays, *p)
        elif kind == "spocone":
            surf.spoCone(rays, *p, ind=ind)
        elif kind == "sphere":
            surf.sphere(rays, *p)
        elif kind == "cyl":
            surf.cyl(rays, *p)
        elif kind == "paraxial":
            surf.paraxial(rays, *p)
        elif kind == "grat":
            tran.grat(rays, *p, ind=ind)


def run_oracle(steps, c, port):
    S, W, T = port.surfacesf, port.woltsurf, port.transformationsf

    def on(cols_names, ind, fn):
        """the reference wrappers' ind path: gather -> call -> scatter"""
        if ind is None:
            fn(*[c[k] for k in cols_names])
        else:
            sub = [np.ascontiguousarray(c[k][ind]) for k in cols_names]
            fn(*sub)
            for k, s_ in zip(cols_names, sub):
                c[k][ind] = s_
    nine = ALLCOLS[1:]
    for kind, p, ind in steps:
        if kind == "transform":
            on(nine, ind, lambda *a: T.transform(*a, *[-v for v in p]))
        elif kind == "itransform":
            on(nine, ind, lambda *a: T.itransform(*a, *[-v for v in p]))
        elif kind == "flat":
            on(nine, ind, lambda *a: S.flat(*a))
        elif kind == "flatopd":
            if ind is None:
                S.flatopd(*[c[k] for k in nine], c["opd"], p[0])
            else:
                on(nine, ind, lambda *a: S.flat(*a))
        elif kind == "reflect":
            on(ALLCOLS[4:], ind, lambda *a: T.reflect(*a))
        elif kind == "conic":
            S.conic(*[c[k] for k in nine], *p)
        elif kind == "conicopd":
            S.conicopd(*[c[k] for k in ALLCOLS], *p)
        elif kind == "wolterprimary":
            W.wolterprimary(*[c[k] for k in nine], *p)
        elif kind == "woltersecondary":
            W.woltersecondary(*[c[k] for k in nine], *p)
        elif kind == "spocone":
            on(nine, ind, lambda *a: W.spocone(*a, *p))
        elif kind == "sphere":
            S.tracesphere(*[c[k] for k in nine], *p)
        elif kind == "cyl":
            S.tracecyl(*[c[k] for k in nine], *p)
        elif kind == "paraxial":
            S.paraxial(*[c[k] for k in nine], *p)
        elif kind == "grat":
            on(["x", "y", "l", "m", "n"], ind, lambda *a: T.grat(*a, p[0], p[1], p[2]))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PXF_FUZZ_SEEDS", "12"))))   # (soak: PXF_FUZZ_SEEDS=500)
def test_random_programs_eager_fused_oracle(port, seed):
    import pyxfocus_amd as pxf
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([777, 4096, 20_001]))
    c = cases._randrays(rng, n, nsign=-1.)
    c["z"] = rng.uniform(50, 400, n)
    c["x"] *= .5; c["y"] *= .5
    steps = random_program(rng, n)
    a = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    b = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
    run_gpu(steps, a, surf, tran)
    with pxf.fused() as rec:
        run_gpu(steps, b, surf, tran)
    assert rec.launches == 1
    run_oracle(steps, c, port)
    kinds = [s[0] for s in steps]
    for k, u, v in zip(ALLCOLS, a, b):
        hu, hv = u.cpu().numpy(), v.cpu().numpy()
        assert bits_equal(hu, hv).all(), ("fused != eager", k, kinds)
        assert bits_equal(hu, c[k]).all(), ("gpu != oracle", k, kinds, int((~bits_equal(hu, c[k])).sum()))
