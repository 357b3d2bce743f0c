"""dev: config-4 program (zernsurf, reflect, flat) timing, 1e8 rays; run with and without PXF_PROGRAM_NO_CHAIN=1."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pyxfocus_amd as pxf
import pyxfocus_amd.sources as sources, pyxfocus_amd.surfaces as surf, pyxfocus_amd.transformations as tran
from tools import workloads as wl
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
par = wl.config4_setup(surf.zmodes)
ms = []
for r in range(6):
    rays = sources.circularbeam(50., n, rng="philox", seed=r)
    wl.config4_prepare(rays)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    with pxf.fused():
        wl.config4_chain(rays, tran, surf, par)
    b.record(); torch.cuda.synchronize()
    if r: ms.append(a.elapsed_time(b))
    del rays
print("config4 program ms per %g rays:" % n, round(float(np.median(ms)), 3), "chain off" if os.environ.get("PXF_PROGRAM_NO_CHAIN") else "straight line")
