"""dev: the W-S secondary design of param_soak.py that showed a 400 mm discrepancy -- are those rays the ones that use
up the 26 allowed steps (restored to their start on one side, converged on the other)?"""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import api, cases
from util import ALLCOLS, call_gpu, to_dev, to_host
port = api.load_port()
NINE = ALLCOLS[1:]
alpha, z0, psi = 0.0024221038813583475, 13621.094538785126, 2.
r0 = z0 * math.tan(4 * alpha)
for seed in range(6):
    rng = np.random.default_rng(seed)
    rin, rout = cases.primrad(z0, r0, z0, 1.), cases.primrad(z0 + 150., r0, z0, 1.)
    c = cases.annulus_rays(rng, 20000, rin, rout, z0 + 300.)
    a = [c[k] for k in NINE]
    port.woltsurf.wsprimary(*a, alpha, z0, psi); port.transformationsf.reflect(*a[3:])
    want = {k: v.copy() for k, v in c.items()}
    port.woltsurf.wssecondary(*[want[k] for k in NINE], alpha, z0, psi)
    t = to_dev(c); call_gpu("woltsurf.wssecondary", t, (alpha, z0, psi)); got = to_host(t)
    d = np.sqrt(sum((got[k] - want[k]) ** 2 for k in "xyz"))
    rest_w = (want["x"] == c["x"]) & (want["z"] == c["z"]); rest_g = (got["x"] == c["x"]) & (got["z"] == c["z"])
    big = d > 1e-6
    print("seed %d: restored (did not converge in 26 steps): oracle %d, gpu %d of %d; rays > 1e-6 mm apart: %d, of which restored on exactly one side: %d; rms of the rest %.2e" % (
        seed, rest_w.sum(), rest_g.sum(), len(d), big.sum(), (big & (rest_w ^ rest_g)).sum(), np.sqrt((d[~big] ** 2).mean())))
