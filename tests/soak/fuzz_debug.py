"""dev: replay one seed of tests/test_gpu_fuzz.py step by step, GPU (eager) vs oracle, and report the first step
after which they differ."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import api, cases
from util import ALLCOLS, bits_equal
import test_gpu_fuzz as F
import pyxfocus_amd.surfaces as surf, pyxfocus_amd.transformations as tran
seed = int(sys.argv[1])
port = api.load_port()
rng = np.random.default_rng(1000 + seed)
n = int(rng.choice([777, 4096, 20_001]))
c = cases._randrays(rng, n, nsign=-1.)
c["z"] = rng.uniform(50, 400, n)
c["x"] *= .5; c["y"] *= .5
steps = F.random_program(rng, n)
a = [torch.from_numpy(c[k].copy()).cuda() for k in ALLCOLS]
for i, st in enumerate(steps):
    F.run_gpu([st], a, surf, tran)
    F.run_oracle([st], c, port)
    bad = {k: int((~bits_equal(u.cpu().numpy(), c[k])).sum()) for k, u in zip(ALLCOLS, a)}
    print(i, st[0], [("%.17g" % v) for v in st[1]][:6], "ind" if st[2] is not None else "", {k: v for k, v in bad.items() if v})
    if any(bad.values()):
        k = [k for k, v in bad.items() if v][0]
        u = a[ALLCOLS.index(k)].cpu().numpy(); j = np.where(~bits_equal(u, c[k]))[0][:3]
        for jj in j: print("   ray", jj, k, repr(u[jj]), repr(c[k][jj]), {q: repr(c[q][jj]) for q in "xyzlmn"})
        break
