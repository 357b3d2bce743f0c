"""dev: radgrat / radgratW / grat / refract with random parameters, GPU vs the oracle at the tests' tolerances."""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import api, cases
from util import ALLCOLS, assert_cols, call_gpu, to_dev, to_host
port = api.load_port()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
n = 4000
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    hub = float(rng.uniform(2000., 20000.)); order = float(rng.integers(-4, 5)); wave = float(rng.uniform(.3, 6.)); per = float(rng.uniform(100., 400.))
    c = cases._randrays(rng, n)
    c["x"] = rng.uniform(-60, 60, n); c["y"] = hub + rng.uniform(-60, 60, n)
    g = float(rng.uniform(.5, 3.)) * math.pi / 180
    c["l"] = rng.normal(0, 1e-3, n); c["m"] = np.full(n, -math.cos(g)) + rng.uniform(-5e-5, 5e-5, n)
    c["n"] = np.sqrt(1 - c["l"] ** 2 - c["m"] ** 2) * rng.choice([-1., 1.], n)
    jobs = [("transformationsf.radgrat", c, (wave, per / hub, order))]
    c2 = {k: v.copy() for k, v in c.items()}; c2["y"] *= rng.choice([-1., 1.], n)
    jobs.append(("transformationsf.radgratw", c2, (rng.uniform(.3, 6., n), per / hub, order)))
    c3 = cases._randrays(rng, n); c3["l"] = rng.uniform(-.01, .01, n); c3["m"] = rng.uniform(-.9999, -.9990, n)
    c3["n"] = np.sqrt(1 - c3["l"] ** 2 - c3["m"] ** 2) * rng.choice([-1., 1.], n)
    jobs.append(("transformationsf.grat", c3, (per, rng.choice([-2., -1., 0., 1., 3.], n), rng.uniform(.5, 6., n))))
    c4 = cases._randrays(rng, n); c4["ux"], c4["uy"], c4["uz"] = cases._unit(rng, n, spread=.2)
    n1, n2 = float(rng.uniform(1., 1.8)), float(rng.uniform(1., 1.8))
    jobs.append(("transformationsf.refract", c4, (n1, n2)))
    for fn, cols, args in jobs:
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        t = to_dev(cols); call_gpu(fn, t, args)
        try:
            assert_cols(fn, to_host(t), want, ALLCOLS, context="soak")
        except AssertionError as e:
            bad += 1; print("MISMATCH", fn, [a if np.isscalar(a) else "arr" for a in args], str(e)[:160])
print("grating soak: %d mismatches" % bad)
