"""Drop-in for PyXFocus ``sources.py``: ray bundles ``[opd, x, y, z, l, m, n, ux, uy, uz]``.

Two generation paths, same distributions:

* default (``rng="numpy"``) -- the reference's algorithm verbatim in behaviour: draws come from
  NumPy's GLOBAL legacy state (``np.random.rand``, rho first then theta; sources.py:40-41), so
  ``np.random.seed(k)`` reproduces the reference bundle bit for bit; the columns are then moved to
  the device.  Right for host-sized bundles and for parity runs.
* ``rng="philox"`` -- generated ON the MI355X by pxf_source (counter-based Philox4x32-10 keyed on
  ``seed`` and the GLOBAL ray index ``first_index + i``): no host memory, no PCIe traffic, and a
  bundle sharded over N GPUs equals the single-GPU bundle.  Available for pointsource,
  circularbeam, annulus, subannulus and rectbeam; needed for the 1e8-1e9-ray configurations.

``device`` defaults to the current ROCm device; ``device="cpu"`` returns host tensors (useful to
inspect a source -- no trace function accepts them, there is no CPU compute path).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

__all__ = ["pointsource", "circularbeam", "annulus", "subannulus", "xslit", "rectArray", "convergingbeam",
           "convergingbeam2", "rectbeam", "gaussianBeam", "fanBeam", "circFan"]


def _device(device):
    if device is None:
        if not torch.cuda.is_available():
            raise _lib.PxfError("pyxfocus_amd.sources: no ROCm device visible (pass device='cpu' only to "
                                "inspect a source; tracing needs an MI355X)")
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def _pack(cols, device):
    dev = _device(device)
    # np.array(copy) per column: several columns are built from one `zero` array and must not alias
    return [torch.from_numpy(np.array(c, dtype=np.float64, order="C", copy=True)).to(dev) for c in cols]


def _philox(kind, num, p, seed, first_index, device, out=None, total=None):
    num = int(num)
    if out is not None:
        # generate INTO an existing bundle (e.g. one shell's slice of a multi-shell bundle)
        rays = list(out)
        if len(rays) != 10 or any((not t.is_cuda) or t.dtype != torch.float64 or t.dim() != 1 or
                                  t.shape[0] != num or not t.is_contiguous() for t in rays):
            raise ValueError("out= must be ten contiguous float64 device columns of length num")
        dev = rays[0].device
    else:
        dev = _device(device)
        if dev.type != "cuda":
            raise _lib.PxfError("rng='philox' generates on the GPU; device must be a ROCm device")
        if num % 2:   # keep every column 16-byte aligned for the 2-rays-per-lane kernels
            rays = [torch.empty(num, dtype=torch.float64, device=dev) for _ in range(10)]
        else:
            buf = torch.empty((10, num), dtype=torch.float64, device=dev)   # one allocation, ten SoA columns
            rays = [buf[i] for i in range(10)]
    arr = (C.c_void_p * 10)(*[r.data_ptr() for r in rays])
    st = torch.cuda.current_stream(dev).cuda_stream
    if _lib.SRC_KINDS[kind] >= 5:       # extended kinds: parameters as a host array, bundle length `total`
        par = (C.c_double * 7)(*([float(v) for v in p] + [0.] * (7 - len(p))))
        with torch.cuda.device(dev):
            rc = _lib.lib().pxf_source_ex(_lib.SRC_KINDS[kind], arr, num, int(total if total is not None else num),
                                          par, int(seed) & (2 ** 64 - 1), int(first_index), st)
        _lib.check(rc, "pxf_source_ex")
        return rays
    p = list(p) + [0.] * (4 - len(p))
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_source(_lib.SRC_KINDS[kind], arr, num, *[float(v) for v in p],
                                   int(seed) & (2 ** 64 - 1), int(first_index), st)
    _lib.check(rc, "pxf_source")
    return rays


def segments(kind, params, rays_per_seg, out, seed=0, first_index=0):
    """Fill a whole multi-segment bundle in ONE launch: segment s (rays [s*rays_per_seg,
    (s+1)*rays_per_seg) of `out`) is a `kind` source ("annulus", "subannulus", "pointsource",
    "circularbeam", "rectbeam") with parameters params[s] = (p0, p1, p2, p3) as documented for
    pxf_source (e.g. subannulus: rin, rout, dphi, zhat).  Device Philox keyed on the global ray index."""
    rays = list(out)
    num = rays[0].shape[0]
    tab = torch.as_tensor(np.asarray(params, dtype=np.float64).reshape(-1, 4)).to(rays[0].device)
    if tab.shape[0] * int(rays_per_seg) != num:
        raise ValueError("len(params) * rays_per_seg must equal the bundle length")
    arr = (C.c_void_p * 10)(*[r.data_ptr() for r in rays])
    dev = rays[0].device
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_source_segments(_lib.SRC_KINDS[kind], arr, num, tab.data_ptr(), tab.shape[0],
                                            int(rays_per_seg), int(seed) & (2 ** 64 - 1), int(first_index),
                                            torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "pxf_source_segments")
    return rays


def pointsource(ang, num, device=None, rng="numpy", seed=0, first_index=0, out=None):
    """Point source with half-angle divergence `ang`; rays point in +z (sources.py:20-53)."""
    if rng == "philox":
        return _philox("pointsource", num, (ang,), seed, first_index, device, out=out)
    rho = np.sqrt(np.random.rand(num)) * np.sin(ang)
    theta = np.random.rand(num) * 2 * np.pi
    l = rho * np.cos(theta)
    m = rho * np.sin(theta)
    n = np.sqrt(1. - l ** 2 - m ** 2)
    zero = np.repeat(0., num)
    return _pack([zero, zero, zero, zero, l, m, n, zero, zero, zero], device)


def circularbeam(rad, num, device=None, rng="numpy", seed=0, first_index=0, out=None):
    """Uniform circular beam of radius `rad`; rays point in +z (sources.py:56-88)."""
    if rng == "philox":
        return _philox("circularbeam", num, (rad,), seed, first_index, device, out=out)
    rho = np.sqrt(np.random.rand(num)) * rad
    theta = np.random.rand(num) * 2 * np.pi
    zero = np.repeat(0., num)
    return _pack([zero, rho * np.cos(theta), rho * np.sin(theta), zero, zero, zero, np.repeat(1., num),
                  zero, zero, zero], device)


def annulus(rin, rout, num, zhat=-1., device=None, rng="numpy", seed=0, first_index=0, out=None):
    """Annulus of rays, uniform in area; default pointing -z (sources.py:91-127)."""
    if rng == "philox":
        return _philox("annulus", num, (rin, rout, 0., zhat), seed, first_index, device, out=out)
    rho = np.sqrt(rin ** 2 + np.random.rand(num) * (rout ** 2 - rin ** 2))
    theta = np.random.rand(num) * 2 * np.pi
    zero = np.repeat(0., num)
    return _pack([zero, rho * np.cos(theta), rho * np.sin(theta), zero, zero, zero, np.repeat(zhat, num),
                  zero, zero, zero], device)


def subannulus(rin, rout, dphi, num, zhat=1., device=None, rng="numpy", seed=0, first_index=0, out=None):
    """Sub-apertured annulus of full angular width dphi centred on +x; default pointing +z
    (sources.py:130-170)."""
    if rng == "philox":
        return _philox("subannulus", num, (rin, rout, dphi, zhat), seed, first_index, device, out=out)
    rho = np.sqrt(rin ** 2 + np.random.rand(num) * (rout ** 2 - rin ** 2))
    theta = np.random.rand(num) * dphi - dphi / 2.
    zero = np.repeat(0., num)
    return _pack([zero, rho * np.cos(theta), rho * np.sin(theta), zero, zero, zero, np.repeat(zhat, num),
                  zero, zero, zero], device)


def xslit(xin, xout, num, zhat=-1., device=None, on_device=False):
    """Slit of rays along x (sources.py:173-207).  on_device=True builds the grid on the GPU
    (same values: np.linspace is reproduced bit for bit)."""
    if on_device:
        return _philox("xslit", num, (xin, xout, 0., zhat), 0, 0, device)
    zero = np.repeat(0., num)
    return _pack([zero, np.linspace(xin, xout, num), zero, zero, zero, zero, np.repeat(zhat, num),
                  zero, zero, zero], device)


def rectArray(xsize, ysize, num, device=None, on_device=False):
    """Regular num x num grid over [-xsize,xsize] x [-ysize,ysize]; rays point +z (sources.py:210-247)."""
    if on_device:
        return _philox("rectArray", num ** 2, (xsize, ysize, num), 0, 0, device)
    x, y = np.meshgrid(np.linspace(-xsize, xsize, num), np.linspace(-ysize, ysize, num))
    zero = np.repeat(0., num ** 2)
    return _pack([zero, x.flatten(), y.flatten(), zero, zero, zero, np.repeat(1., num ** 2),
                  zero, zero, zero], device)


def convergingbeam(zset, rin, rout, tmin, tmax, num, lscat, device=None, rng="numpy", seed=0, first_index=0):
    """Converging sub-apertured annulus beam placed at z=zset, with Lorentzian angular scatter
    `lscat` in arcsec (sources.py:250-297)."""
    if rng == "philox":
        return _philox("convergingbeam", num, (zset, rin, rout, tmin, tmax, lscat), seed, first_index, device)
    rho = np.sqrt(rin ** 2 + np.random.rand(num) * (rout ** 2 - rin ** 2))
    theta = tmin + np.random.rand(num) * (tmax - tmin)
    x = rho * np.cos(theta)
    y = rho * np.sin(theta)
    z = np.repeat(zset, num)
    lscat = lscat * np.tan((np.random.rand(num) - .5) * np.pi)
    lscat = lscat / 60 ** 2 * np.pi / 180.
    n = -np.cos(np.arctan(rho / zset) + lscat)
    l = -np.sqrt(1 - n ** 2) * np.cos(theta)
    m = -np.sqrt(1 - n ** 2) * np.sin(theta)
    zero = np.repeat(0., num)
    return _pack([zero, x, y, z, l, m, n, zero, zero, zero], device)


def convergingbeam2(zset, xmin, xmax, ymin, ymax, num, lscat, device=None, rng="numpy", seed=0, first_index=0):
    """Converging rectangular beam placed at z=zset (sources.py:300-345)."""
    if rng == "philox":
        return _philox("convergingbeam2", num, (zset, xmin, xmax, ymin, ymax, lscat), seed, first_index, device)
    x = xmin + np.random.rand(num) * (xmax - xmin)
    y = ymin + np.random.rand(num) * (ymax - ymin)
    rho = np.sqrt(x ** 2 + y ** 2)
    theta = np.arctan2(y, x)
    z = np.repeat(zset, num)
    lscat = lscat * np.tan((np.random.rand(num) - .5) * np.pi)
    lscat = lscat / 60 ** 2 * np.pi / 180.
    n = -np.cos(np.arctan(rho / zset) + lscat)
    l = -np.sqrt(1 - n ** 2) * np.cos(theta)
    m = -np.sqrt(1 - n ** 2) * np.sin(theta)
    zero = np.repeat(0., num)
    return _pack([zero, x, y, z, l, m, n, zero, zero, zero], device)


def rectbeam(xhalfwidth, yhalfwidth, num, device=None, rng="numpy", seed=0, first_index=0, out=None):
    """Uniform rectangular beam; rays point +z (sources.py:348-377)."""
    if rng == "philox":
        return _philox("rectbeam", num, (xhalfwidth, yhalfwidth), seed, first_index, device, out=out)
    x = (np.random.rand(num) - .5) * 2 * xhalfwidth
    y = (np.random.rand(num) - .5) * 2 * yhalfwidth
    zero = np.repeat(0., num)
    return _pack([zero, x, y, zero, zero, zero, np.repeat(1., num), zero, zero, zero], device)


def gaussianBeam(ang, num, device=None, rng="numpy", seed=0, first_index=0):
    """Point source with Gaussian angular divergence, 1-sigma `ang` (sources.py:380-413)."""
    if rng == "philox":
        return _philox("gaussianBeam", num, (ang,), seed, first_index, device)
    l = np.random.randn(num) * np.sin(ang) / np.sqrt(2)
    m = np.random.randn(num) * np.sin(ang) / np.sqrt(2)
    n = np.sqrt(1. - l ** 2 - m ** 2)
    zero = np.repeat(0., num)
    return _pack([zero, zero, zero, zero, l, m, n, zero, zero, zero], device)


def fanBeam(xang, yang, num, device=None, on_device=False):
    """XY fan of rays on a regular direction-cosine grid (sources.py:416-438)."""
    if on_device:
        return _philox("fanBeam", num ** 2, (xang, yang, num), 0, 0, device)
    xa, ya = np.meshgrid(np.linspace(-xang, xang, num), np.linspace(-yang, yang, num))
    l = np.sin(xa.flatten())
    m = np.sin(ya.flatten())
    n = np.sqrt(1. - l ** 2 - m ** 2)
    zero = np.repeat(0., np.size(l))
    return _pack([zero, zero, zero, zero, l, m, n, zero, zero, zero], device)


def circFan(halfang, rings, arms, device=None, on_device=False):
    """Uniformly spaced rays on a diverging cone (sources.py:441-471)."""
    if on_device:
        return _philox("circFan", rings * arms, (halfang, rings, arms), 0, 0, device)
    rad = np.linspace(0, halfang, rings)
    az = np.linspace(0, 2 * np.pi, arms + 1)[0:-1]
    rr, aa = np.meshgrid(rad, az)
    xa = (np.sin(rr) * np.cos(aa)).flatten()
    ya = (np.sin(rr) * np.sin(aa)).flatten()
    l = np.sin(xa)
    m = np.sin(ya)
    n = np.sqrt(1. - l ** 2 - m ** 2)
    zero = np.repeat(0., np.size(l))
    return _pack([zero, zero, zero, zero, l, m, n, zero, zero, zero], device)
