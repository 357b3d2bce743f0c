"""The BASELINE.json workloads as call chains against the PyXFocus API (`tran`, `surf` = any pair of
namespaces with the reference's signatures: pyxfocus_amd's modules on device tensors, or
oracle/wrappers.py on NumPy columns for the CPU checker).  Used by bench.py, tools/bench_configs.py
and the parity legs; SURVEY 8(d) defines the chains and their algorithmic bytes per ray.
"""
import math

import numpy as np

R0, Z0, PSI, LMIR = 220., 8400., 1., 100.
GRAZE, ZG = 1.5 * math.pi / 180, 8000.        # grating incidence angle; grating plane above the focus [mm]
HBM_PEAK_GBS = 8000.0                         # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is achievable
FP64_VECTOR_PEAK_TFLOPS = 78.6                # MI355X fp64 vector peak

# (launch name, algorithmic bytes per ray) -- SURVEY 8(d): columns read + columns written, fp64.
# The chain's leading `transform(rays, 0, 0, -(z0 + 2L), 0, 0, 0)` is a pure translation along z: all nine numbers
# are read and go through the reference's arithmetic, but only z comes back different, and the kernel stores only
# what changed (TransformOp, pxf_ops_transforms.h): 72 B read + 8 B written = 80 B/ray, where SURVEY 8(d) booked
# the general 144 (the PMC traffic confirms both: profiles/r02d_config{2,3}_kernels.md).
SURVEY_BYTES = {"config2": 576, "config3": 1192}     # SURVEY 8(d)'s own figures, every transform booked at 144
CONFIG2 = [("transform", 80), ("wolterprimary", 96), ("reflect", 72), ("woltersecondary", 96), ("reflect", 72),
           ("flat", 96)]
# the grating stage's frame changes rotate about x only: x, l, ux come back unchanged and are not rewritten (120 B)
CONFIG3_TAIL = [("transform(g)", 120), ("flat(g)", 96), ("reflect(g)", 72), ("radgrat", 64), ("itransform(g)", 120),
                ("flat(focal)", 96)]
CONFIG3 = CONFIG2 + CONFIG3_TAIL
CONFIG4 = [("zernsurf", 96), ("reflect", 72), ("flat", 96)]
BYTES = {"config2": 512, "config3": 1080, "config4": 264}
# one launch: 6 columns in (x y z l m n), 9 out; the Zernike program reads the same six
FUSED_BYTES = {"config2": 120, "config3": 120, "config4": 120}


def _noop(i):
    pass


def config2_chain(rays, tran, surf, mark=_noop):
    """transform -> wolterprimary -> reflect -> woltersecondary -> reflect -> flat.  `mark(i)` is
    called before launch i and once after the last (HIP-event hooks)."""
    mark(0)
    tran.transform(rays, 0, 0, -(Z0 + 2 * LMIR), 0, 0, 0)
    mark(1)
    surf.wolterprimary(rays, R0, Z0, PSI)
    mark(2)
    tran.reflect(rays)
    mark(3)
    surf.woltersecondary(rays, R0, Z0, PSI)
    mark(4)
    tran.reflect(rays)
    mark(5)
    surf.flat(rays)
    mark(6)


def grating_stage(rays, tran, surf, mark=_noop, first=0):
    """One radial grating between the mirrors and the focus (sector.py:652-704 reduced to a single
    grating): frame to the grating (hub = focus), flat, reflect, radgrat, frame back, flat."""
    ty, rx = -ZG * math.sin(GRAZE), math.pi / 2 - GRAZE
    mark(first + 0)
    tran.transform(rays, 0, ty, 0, rx, 0, 0)
    mark(first + 1)
    surf.flat(rays)
    mark(first + 2)
    tran.reflect(rays)
    mark(first + 3)
    tran.radgrat(rays, 160. / ZG, 1., 2.4)
    mark(first + 4)
    tran.itransform(rays, 0, ty, 0, rx, 0, 0)
    mark(first + 5)
    surf.flat(rays)
    mark(first + 6)


def config3_chain(rays, tran, surf, mark=_noop):
    config2_chain(rays, tran, surf, mark)
    grating_stage(rays, tran, surf, mark, first=6)


def config4_setup(zmodes, nterms=36, seed=4):
    """36 Noll-ordered coefficients ~N(0, 1e-3) mm (piston 0), radius 50 mm.
    `zmodes(nterms)` -> (rorder, aorder): pyxfocus_amd.surfaces.zmodes, or oracle.cases.noll_orders."""
    rng = np.random.default_rng(seed)
    coeff = rng.normal(0., 1e-3, nterms)
    coeff[0] = 0.
    ro, ao = zmodes(nterms)
    return coeff, 50., ro, ao


def config4_prepare(rays):
    """circularbeam(50) raised to z = 10 mm, tilted by l = 0.01, travelling in -z."""
    for j, v in ((3, 10.), (4, 0.01), (6, -math.sqrt(1 - 1e-4))):
        rays[j][...] = v


def config4_chain(rays, tran, surf, par, mark=_noop):
    coeff, rad, ro, ao = par
    mark(0)
    surf.zernsurf(rays, coeff, rad, rorder=ro, aorder=ao)
    mark(1)
    tran.reflect(rays)
    mark(2)
    surf.flat(rays)
    mark(3)


def annulus_radii():
    from math import sqrt
    # conicsolve.primrad (conicsolve.py:7-15) at the two ends of the primary; same libm scalars as the reference
    alpha = .25 * math.atan(R0 / Z0)
    thetah = 2 * (1 + 2 * PSI) / (1 + PSI) * alpha
    thetap = 2 * PSI / (1 + PSI) * alpha
    p = Z0 * math.tan(4 * alpha) * math.tan(thetap)
    d = Z0 * math.tan(4 * alpha) * math.tan(4 * alpha - thetah)
    e = math.cos(4 * alpha) * (1 + math.tan(4 * alpha) * math.tan(thetah))
    f = lambda z: sqrt(p ** 2 + 2 * p * z + (4 * e ** 2 * p * d) / (e ** 2 - 1))
    return f(Z0), f(Z0 + LMIR)
