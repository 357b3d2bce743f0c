"""dev: weighted hpd vs the reference's argsort / cumsum / argmin on the same radii; on a mismatch, show which sorted
neighbours were picked."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pyxfocus_amd.analyses as anal
nbad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1_000_000, 5_000_000))
    sc = float(rng.choice([1e-6, 1., 1e3]))
    x = rng.normal(0, 1, n) * sc + float(rng.choice([0., 5., -300.])); y = rng.normal(0, 2, n)
    w = rng.random(n) + .01
    rays = [torch.zeros(1, dtype=torch.float64, device="cuda")] * 10
    rays = list(rays); rays[1] = torch.from_numpy(x).cuda(); rays[2] = torch.from_numpy(y).cuda()
    cxw, cyw = np.average(x, weights=w), np.average(y, weights=w)
    gcx, gcy = anal.centroid(rays, weights=torch.from_numpy(w).cuda())
    r = np.sqrt((x - cxw) ** 2 + (y - cyw) ** 2)
    o = np.argsort(r, kind="stable"); rs = r[o]; cdf = np.cumsum(w[o]); cdf = cdf / cdf.max()
    i75, i25 = int(np.argmin(np.abs(cdf - .75))), int(np.argmin(np.abs(cdf - .25)))
    want = rs[i75] - rs[i25]
    got = anal.hpd(rays, weights=torch.from_numpy(w).cuda())
    if abs(got - want) > 1e-9 * max(1., abs(want)):
        nbad += 1
        # radii about the GPU's centroid
        r2 = np.sqrt((x - gcx) ** 2 + (y - gcy) ** 2); o2 = np.argsort(r2, kind="stable"); rs2 = r2[o2]
        c2 = np.cumsum(w[o2]); c2 = c2 / c2.max(); j75, j25 = int(np.argmin(np.abs(c2 - .75))), int(np.argmin(np.abs(c2 - .25)))
        print("seed", seed, "n", n, "scale", sc, "got %.13g want %.13g" % (got, want), "centroid diff", gcx - cxw, gcy - cyw,
              "| with the GPU's centroid NumPy gives %.13g" % (rs2[j75] - rs2[j25]),
              "| neighbours at 75%%: %.13g %.13g %.13g" % tuple(rs[i75 - 1:i75 + 2]), "cdf gaps", abs(cdf[i75] - .75), abs(cdf[i75 - 1] - .75), abs(cdf[i75 + 1] - .75))
print("weighted hpd: %d mismatches" % nbad)
