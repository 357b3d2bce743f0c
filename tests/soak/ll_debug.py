"""dev: LL register form vs oracle for a sweep of (kt, ka)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import api, cases
from util import ALLCOLS, call_gpu, to_dev, to_host
port = api.load_port()
base = {name: (fn, cols, args) for name, fn, cols, args in cases.cases(num=300, seed=99)}
for kt, ka in [(3, 8), (3, 15), (3, 20), (3, 25), (3, 29), (3, 30), (3, 31), (7, 31), (1, 31), (0, 31)]:
    rng = np.random.default_rng(1000 * kt + ka)
    nc = (ka + 1) * (kt + 1)
    ax, az = np.meshgrid(np.arange(ka + 1), np.arange(kt + 1), indexing="ij")
    coeff = rng.normal(0., 2e-5, nc) / (1 + ax.flatten() + az.flatten())
    ax, az = ax.flatten().astype(np.int32), az.flatten().astype(np.int32)
    for name in ("wolterprimll",):
        fn, cols, args = base[name]
        args = tuple(args[:-3]) + (coeff, ax, az)
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        t = to_dev(cols)
        call_gpu(fn, t, args)
        got = to_host(t)
        d = np.sqrt(sum((got[k] - want[k]) ** 2 for k in "xyz"))
        print(kt, ka, name, "max %.2e" % np.nanmax(d), "n>1e-10:", int((d > 1e-10).sum()), "worst ray", int(np.nanargmax(d)))
