"""dev: the device grid sources (xslit, rectArray, fanBeam, circFan) against the NumPy path of the same functions
(= the reference's own sources.py semantics, bit-equal by tests/test_host.py) for random sizes and extents."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pyxfocus_amd.sources as S
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    n = int(rng.choice([1, 2, 3, 7, 64, 513, 1000, int(rng.integers(2, 1500))]))
    a, b = float(rng.uniform(.1, 50.)), float(rng.uniform(.1, 50.))
    for tag, dev, host, exact in (
            ("xslit", lambda: S.xslit(-a, b, n * 7, on_device=True), lambda: S.xslit(-a, b, n * 7), range(10)),
            ("rectArray", lambda: S.rectArray(a, b, n, on_device=True), lambda: S.rectArray(a, b, n), (0, 1, 2, 3, 7, 8, 9)),
            ("fanBeam", lambda: S.fanBeam(a * 1e-3, b * 1e-3, n, on_device=True), lambda: S.fanBeam(a * 1e-3, b * 1e-3, n), (0, 1, 2, 3, 7, 8, 9)),
            ("circFan", lambda: S.circFan(a * 1e-3, max(n % 40, 1), max(n % 17, 1), on_device=True),
             lambda: S.circFan(a * 1e-3, max(n % 40, 1), max(n % 17, 1)), (0, 1, 2, 3, 7, 8, 9))):
        if tag == "xslit" and n * 7 < 2 or tag != "xslit" and n < 2:
            continue
        d, h = dev(), host()
        for k in range(10):
            u, v = d[k].cpu().numpy(), h[k].cpu().numpy()
            if u.shape != v.shape:
                bad += 1; print("SHAPE", tag, n, k, u.shape, v.shape); break
            if k in exact:
                if not np.array_equal(u, v):
                    bad += 1; print("MISMATCH", tag, n, a, b, "col", k, int((u != v).sum())); break
            elif not np.abs(u - v).max() <= 4e-16:                       # direction cosines: sin/cos of ocml vs numpy
                bad += 1; print("TOL", tag, n, "col", k, np.abs(u - v).max()); break
print("source soak: %d mismatches" % bad)
