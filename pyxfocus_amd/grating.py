"""Blaze / Littrow design helpers of PyXFocus ``grating.py`` (grating.py:5-38): scalar (or array)
formulas on the host.  (``transformations.radgrat`` / ``grat`` are the per-ray grating kernels.)

Everything derives from one quantity, the order term q = m*wave/d: the Littrow yaw is
asin(q / (2 cos inc)), and the diffracted direction cosines follow from the yawed geometry."""
import numpy as np


def _littrow(inc, wave, m, d):
    """(yaw, beta, alpha): Littrow yaw and the direction cosines along / across the grooves."""
    c = np.cos(inc)
    yaw_ = np.arcsin(m * wave / 2 / d / c)
    return yaw_, c * np.cos(yaw_), c * np.sin(yaw_) - m * wave / d


def blazeYaw(inc, wave, m, d):
    """Yaw that puts wavelength `wave` of order m (groove period d) at blaze for incidence `inc`."""
    return _littrow(inc, wave, m, d)[0]


def litBetaAlpha(inc, wave, m, d):
    """(beta, alpha) direction cosines at the Littrow condition."""
    return _littrow(inc, wave, m, d)[1:]


def blazeAngle(inc, wave, m, d):
    """Blaze angle for wavelength `wave` at order m, incidence `inc`, groove period d."""
    _, beta, alpha = _littrow(inc, wave, m, d)
    return np.arcsin(alpha / np.cos(np.arcsin(beta)))


def blazeAngle2(inc, wave, m, d):
    """The same from the closed form asin(a / sqrt(sin^2(inc) + a^2)), a = m wave / 2d."""
    a = m * wave / 2 / d
    return np.arcsin(a / np.sqrt(1 - np.cos(inc) ** 2 + a ** 2))


def eta(phi0, theta0):
    """Graze angle from the polar pair (phi0, theta0)."""
    return np.arcsin(np.cos(phi0) * np.cos(theta0))


def yaw(phi0, theta0):
    """Yaw from the polar pair (phi0, theta0)."""
    return np.arctan(np.sin(theta0) / np.tan(phi0))


def fanArray(rays, hubdist, length, ang, dpermm, order, wave, thr=.001, coords=None, maxiter=100000):
    """A fanned array of radial gratings about a common hub, per ray, on the device (EXTENSION: the reference
    has no such function -- it is the `while` loop of its examples/arcus/sector.py:663-704, `gratArray`).

    Equivalent, bit for bit, to this loop through the drop-in API (rays already in the frame of the first
    grating, hub at the origin)::

        indg = abs(arcsin(n)) > thr;  flat(rays, ind=indg)
        rho = -sqrt(x**2 + y**2) * sign(y);  ind = (rho > hubdist) & (rho < length + hubdist)
        ind2 = prev = ind;  i = 0
        while prev.sum() < len(x):
            i += 1
            reflect(rays, ind=ind2);  radgrat(rays, dpermm, order, wave, ind=ind2)
            transform(rays, 0, 0, 0, ang, 0, 0, coords=coords)
            flat(rays, ind=~prev & (abs(arcsin(n)) > thr))
            rho = ...;  ind = ...;  ind2 = ~prev & ind;  prev = prev | ind
        reflect(rays, ind=ind2);  radgrat(rays, dpermm, order, wave, ind=ind2)

    which costs a dozen passes over the bundle and a host synchronisation per grating row (~226 rows for the
    Arcus petal); here every ray walks the fan in registers with the same device code (two launches).
    Returns i, the number of rows walked (the caller undoes `ang * i` of rotation, sector.py:718).

    A ray that never lands on a grating (a dead ray: zeroed or NaN direction cosines from an upstream miss) makes
    the reference's loop spin forever; here it raises RuntimeError after `maxiter` rows -- and by then the search has
    rotated every ray in place by its own number of rows: THE BUNDLE IS UNDEFINED after that error (and `coords`
    is not updated).  Vignette dead rays first (`transformations.vignette(rays)`), as the reference's drivers do
    before their grating stage (sector.py:629-634), or trace on a copy."""
    import ctypes as C

    import torch

    from . import _fused as fused
    from . import _lib, native
    from . import transformations as tran
    fused.flush()
    cols = list(rays[1:10])
    num, dev = native._check_cols(cols, "fanArray")
    it_hit = torch.empty(max(num, 1), dtype=torch.int32, device=dev)
    scal = torch.zeros(2, dtype=torch.int64, device=dev)           # [imax (low int32) | nmiss (uint64)]
    ptrs = (C.c_void_p * 10)(None, *[t.data_ptr() for t in cols])
    h, st = _lib.lib(), native._stream(dev)
    with torch.cuda.device(dev):
        _lib.check(h.pxf_fan_search(ptrs, num, float(hubdist), float(length), float(ang), float(thr), int(maxiter),
                                    it_hit.data_ptr(), scal.data_ptr(), scal[1:].data_ptr(), st), "pxf_fan_search")
        imax, nmiss = [int(v) for v in scal.tolist()]                         # the one sync: the row count is data-dependent
        imax &= 0xffffffff
        if nmiss:
            raise RuntimeError("fanArray: %d rays never reached a grating within %d rows; the bundle is undefined now "
                               "(vignette dead rays before the grating stage)" % (nmiss, maxiter))
        _lib.check(h.pxf_fan_finish(ptrs, num, float(ang), float(dpermm), float(order), float(wave), it_hit.data_ptr(),
                                    imax, st), "pxf_fan_finish")
    if coords is not None:
        for _ in range(imax):
            tran.update_coords(coords, 0, 0, 0, ang, 0, 0)
    return imax
