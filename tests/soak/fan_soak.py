"""dev: random fanned-grating geometries through grating.fanArray and through the host loop it replaces (the API-level
`while` loop of tests/test_gpu_fan.py), bit for bit: hub distance, grating length, graze angle, sign of the pitch,
grazing threshold (inside or below the graze-angle range), number of rows, diffraction order / wavelength, with and
without special values salted into the bundle.  usage: fan_soak.py [iterations] [seed]"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyxfocus_amd.surfaces as surf  # noqa: E402
import pyxfocus_amd.transformations as tran  # noqa: E402
from pyxfocus_amd import grating  # noqa: E402
from test_gpu_fan import fan_loop, fan_rays  # noqa: E402
from util import bits_equal  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(iters):
    H = float(rng.uniform(500., 15000.))
    Lg = float(rng.uniform(20., 150.))
    inc = float(rng.uniform(.8, 3.))
    sgn = float(rng.choice([1., 1., -1.]))
    thr = float(rng.choice([.001, 1e-4, math.sin(math.radians(inc)) * rng.uniform(.5, .9)]))
    kmax = int(rng.integers(3, 70))
    salt = bool(rng.integers(0, 2))
    n = int(rng.choice([4097, 60_000, 300_001]))
    wave = float(rng.uniform(.05, .4))
    order = int(rng.choice([1, -1]))
    ang = sgn * Lg * math.sin(math.radians(inc)) / H * .95
    cols = fan_rays(np.random.default_rng(int(rng.integers(1 << 30))), n, H, Lg, ang, math.radians(inc), kmax, salt)
    a = [torch.from_numpy(c.copy()).cuda() for c in cols]
    b = [torch.from_numpy(c.copy()).cuda() for c in cols]
    try:
        rows_a = fan_loop(torch, tran, surf, a, H, Lg, ang, 160. / H, order, wave, thr, 4 * kmax + 60)
    except AssertionError as e:                      # a geometry in which some ray never lands: not a comparison case
        print("skip", it, str(e)[:60])
        continue
    rows_b = grating.fanArray(b, H, Lg, ang, 160. / H, order, wave, thr=thr, maxiter=4 * kmax + 60)
    diff = {k: int((~bits_equal(a[k].cpu().numpy(), b[k].cpu().numpy())).sum()) for k in range(1, 10)}
    if rows_a != rows_b or any(diff.values()):
        bad += 1
        print("MISMATCH", it, dict(H=H, Lg=Lg, inc=inc, sgn=sgn, thr=thr, kmax=kmax, salt=salt, n=n), rows_a, rows_b,
              {k: v for k, v in diff.items() if v})
print("fan soak: %d iterations, %d mismatches" % (iters, bad))
