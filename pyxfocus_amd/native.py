"""f2py-shaped front end of the HIP library, on torch tensors.

The four namespaces carry the names of the reference's f2py extension modules
(``PyXFocus.surfacesf / woltsurf / zernsurf / transformationsf``, compiletrace.sh:1-6) and the same
call signatures: SoA columns first, then scalars; columns are mutated in place;
``num`` is implied by ``len(x)``.  Columns must be 1-D contiguous ``torch.float64`` tensors on
one ROCm device (f2py raises for non-contiguous / wrong-dtype ``intent(inout)`` arrays; so do we).

Extra keyword on every call: ``mask`` -- a ``torch.bool``/``uint8`` tensor of the same length; rays
with mask 0 are left untouched (the reference wrappers' ``ind=`` path, done in-kernel).
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _fused as fused
from . import _lib


_F64 = torch.float64
_ONE = (1,)


def _bad_cols(cols, what):
    """Slow path of _check_cols: say exactly what is wrong (or accept what the fast test was too
    strict about)."""
    t0 = cols[0]
    for t in cols:
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s: ray columns must be torch tensors, got %s" % (what, type(t).__name__))
        if t.dtype != torch.float64 or t.dim() != 1 or not t.is_contiguous():
            raise TypeError("%s: ray columns must be 1-D contiguous float64 tensors" % what)
        if not t.is_cuda:
            raise TypeError("%s: ray columns must live on a ROCm device (no CPU path exists)" % what)
        if t.device != t0.device:
            raise ValueError("%s: ray columns live on different devices" % what)
        if t.shape[0] != t0.shape[0]:
            raise ValueError("%s: ray columns differ in length" % what)
    return t0.shape[0], t0.device        # e.g. a one-element view with a non-unit stride: contiguous after all


def _check_cols(cols, what):
    """Every pointer handed to a kernel is validated on EVERY call (a wrong device / stride / length
    would be a GPU fault, not an exception): float64, 1-D, unit stride, one ROCm device, one length.
    Written for the host issue rate: three cheap calls per column, diagnosis only on failure."""
    t0 = cols[0]
    try:
        num, dev = t0.numel(), t0.device
        if dev.type != "cuda":
            return _bad_cols(cols, what)
        for t in cols:
            # a 1-D tensor with stride (1,) is contiguous
            if t.dtype is not _F64 or t.stride() != _ONE or t.numel() != num or t.device != dev:
                return _bad_cols(cols, what)
    except AttributeError:
        return _bad_cols(cols, what)
    return num, dev


def _mask_ptr(mask, num, dev, what):
    if mask is None:
        return None
    if not isinstance(mask, torch.Tensor) or mask.dtype not in (torch.bool, torch.uint8):
        raise TypeError("%s: mask must be a torch.bool / torch.uint8 tensor" % what)
    if mask.device != dev or mask.dim() != 1 or mask.shape[0] != num or not mask.is_contiguous():
        raise ValueError("%s: mask must be a contiguous 1-D tensor of the rays' length and device" % what)
    return mask.data_ptr()


_FN = {}   # name -> bound ctypes function (argtypes already set by _lib.lib())


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream      # what torch's own extensions use (~0.3 us)
except AttributeError:                                     # pragma: no cover - other torch builds
    _raw_stream = None


def _stream(dev):
    """The HIP stream torch is currently issuing on for `dev` (hipStream_t as an integer)."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


def _call(name, cols, tail, mask=None, aux_last=False):
    """cols -> device pointers, then num, then `tail` scalars, then mask, stream.
    Inside a `pyxfocus_amd.fused()` block the call is recorded instead (see _fused.py).
    aux_last: the last entry of `cols` is an extra per-ray column that the C entry takes AFTER the
    scalars (pxf_grat's order*wave column)."""
    num, dev = _check_cols(cols, name)
    mp = _mask_ptr(mask, num, dev, name)
    rec = fused.active()
    if rec is not None:
        if num == 0:
            return
        if rec.try_add(name, cols, tail, mask):
            return
        rec.flush()                      # cannot join the pending program: run it, then go on
        if rec.try_add(name, cols, tail, mask):
            return
    f = _FN.get(name)
    if f is None:
        f = _FN[name] = getattr(_lib.lib(), name)
    ptrs = [t.data_ptr() for t in cols]
    extra = [ptrs.pop()] if aux_last else []
    if dev.index == torch.cuda.current_device():          # the common case: skip the device guard (~6 us)
        rc = f(*ptrs, num, *tail, *extra, mp, _stream(dev))
    else:
        with torch.cuda.device(dev):
            rc = f(*ptrs, num, *tail, *extra, mp, _stream(dev))
    if rc:
        _lib.check(rc, name)


# ---- surfacesf ------------------------------------------------------------------------------
def flat(x, y, z, l, m, n, ux, uy, uz, mask=None):
    _call("pxf_flat", (x, y, z, l, m, n, ux, uy, uz), (), mask)


def flatopd(x, y, z, l, m, n, ux, uy, uz, opd, nr, mask=None):
    _call("pxf_flatopd", (x, y, z, l, m, n, ux, uy, uz, opd), (float(nr),), mask)


def conic(x, y, z, l, m, n, ux, uy, uz, R, K, mask=None):
    _call("pxf_conic", (x, y, z, l, m, n, ux, uy, uz), (float(R), float(K)), mask)


def conicopd(opd, x, y, z, l, m, n, ux, uy, uz, R, K, nr, mask=None):
    _call("pxf_conicopd", (opd, x, y, z, l, m, n, ux, uy, uz), (float(R), float(K), float(nr)), mask)


def tracesphere(x, y, z, l, m, n, ux, uy, uz, rad, mask=None):
    _call("pxf_tracesphere", (x, y, z, l, m, n, ux, uy, uz), (float(rad),), mask)


def tracesphereopd(opd, x, y, z, l, m, n, ux, uy, uz, rad, nr, mask=None):
    _call("pxf_tracesphereopd", (opd, x, y, z, l, m, n, ux, uy, uz), (float(rad), float(nr)), mask)


def tracecyl(x, y, z, l, m, n, ux, uy, uz, rad, mask=None):
    _call("pxf_tracecyl", (x, y, z, l, m, n, ux, uy, uz), (float(rad),), mask)


def tracecylopd(opd, x, y, z, l, m, n, ux, uy, uz, rad, nr, mask=None):
    _call("pxf_tracecylopd", (opd, x, y, z, l, m, n, ux, uy, uz), (float(rad), float(nr)), mask)


def cylconic(x, y, z, l, m, n, ux, uy, uz, rad, k, mask=None):
    _call("pxf_cylconic", (x, y, z, l, m, n, ux, uy, uz), (float(rad), float(k)), mask)


def paraxial(x, y, z, l, m, n, ux, uy, uz, F, mask=None):
    # f2py signature takes all nine columns; the loop touches x, y, l, m only (surfacesf.f95:431-436)
    _check_cols((x, y, z, l, m, n, ux, uy, uz), "pxf_paraxial")
    _call("pxf_paraxial", (x, y, l, m), (float(F),), mask)


def paraxialy(x, y, z, l, m, n, ux, uy, uz, F, mask=None):
    _check_cols((x, y, z, l, m, n, ux, uy, uz), "pxf_paraxialy")
    _call("pxf_paraxialy", (y, m), (float(F),), mask)


def torus(x, y, z, l, m, n, ux, uy, uz, rin, rout, mask=None):
    fused.flush()
    _call("pxf_torus", (x, y, z, l, m, n, ux, uy, uz), (float(rin), float(rout)), mask)


def _f64host(a):
    k = np.ascontiguousarray(np.asarray(a if not isinstance(a, torch.Tensor) else a.cpu(), dtype=np.float64)).reshape(-1)
    return k, k.ctypes.data_as(_lib._DP)


def _i32host(a):
    k = np.ascontiguousarray(np.asarray(a if not isinstance(a, torch.Tensor) else a.cpu()), dtype=np.int32).reshape(-1)
    return k, k.ctypes.data_as(_lib._IP)


def conicplus(x, y, z, l, m, n, ux, uy, uz, R, K, p, mask=None):
    pk, pp = _f64host(p)
    fused.flush()
    _call("pxf_conicplus", (x, y, z, l, m, n, ux, uy, uz), (float(R), float(K), pp, len(pk)), mask)


def conicplusopd(opd, x, y, z, l, m, n, ux, uy, uz, R, K, p, nr, mask=None):
    pk, pp = _f64host(p)
    fused.flush()
    _call("pxf_conicplusopd", (opd, x, y, z, l, m, n, ux, uy, uz), (float(R), float(K), pp, len(pk), float(nr)), mask)


def legsurf(x, y, z, l, m, n, ux, uy, uz, xwidth, ywidth, order, coeff, xo, yo, mask=None):
    _check_cols((x, y, z, l, m, n, ux, uy, uz), "pxf_legsurf")
    ck, cp = _f64host(coeff)
    xk, xp = _i32host(xo)
    yk, yp = _i32host(yo)
    if not (len(ck) == len(xk) == len(yk)):
        raise ValueError("legsurf: coeff, xo, yo must have equal length")
    fused.flush()
    _call("pxf_legsurf", (x, y, l, m, n), (float(xwidth), float(ywidth), float(order), cp, xp, yp, len(ck)), mask)


# ---- woltsurf -------------------------------------------------------------------------------
def spocone(x, y, z, l, m, n, ux, uy, uz, R0, tg, mask=None):
    _call("pxf_spocone", (x, y, z, l, m, n, ux, uy, uz), (float(R0), float(tg)), mask)



def wolterprimary(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, mask=None):
    _call("pxf_wolterprimary", (x, y, z, l, m, n, ux, uy, uz), (float(r0), float(z0), float(psi)), mask)


def wolterprimaryopd(opd, x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, nr, mask=None):
    _call("pxf_wolterprimaryopd", (opd, x, y, z, l, m, n, ux, uy, uz),
          (float(r0), float(z0), float(psi), float(nr)), mask)


def woltersecondary(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, mask=None):
    _call("pxf_woltersecondary", (x, y, z, l, m, n, ux, uy, uz), (float(r0), float(z0), float(psi)), mask)


def woltersine(x, y, z, l, m, n, ux, uy, uz, r0, z0, amp, freq, mask=None):
    _call("pxf_woltersine", (x, y, z, l, m, n, ux, uy, uz), tuple(float(v) for v in (r0, z0, amp, freq)), mask)


def _ll_tables(coeff, axial, az):
    ck = np.ascontiguousarray(np.asarray(coeff if not isinstance(coeff, torch.Tensor) else coeff.cpu(),
                                         dtype=np.float64)).reshape(-1)
    ak = np.ascontiguousarray(np.asarray(axial), dtype=np.int32).reshape(-1)
    zk = np.ascontiguousarray(np.asarray(az), dtype=np.int32).reshape(-1)
    if not (len(ck) == len(ak) == len(zk)):
        raise ValueError("coeff, axial, az must have equal length")
    return (ck, ak, zk), (ck.ctypes.data_as(_lib._DP), ak.ctypes.data_as(_lib._IP), zk.ctypes.data_as(_lib._IP), len(ck))


def _record_ll(name, cols, scalars, tables, S, mask):
    """Inside a fused() block: record the Legendre-Legendre mirror into the pending program (True), making room
    for it by flushing if need be; False = run it as a stand-alone launch."""
    rec = fused.active()
    if rec is None:
        return False
    num, dev = _check_cols(cols, name)
    _mask_ptr(mask, num, dev, name)
    if num == 0:
        return True
    if rec.try_add_ll(name, cols, scalars, tables, S, mask):
        return True
    rec.flush()
    return rec.try_add_ll(name, cols, scalars, tables, S, mask)


def wolterprimll(x, y, z, l, m, n, ux, uy, uz, r0, z0, zmax, zmin, dphi, coeff, axial, az, mask=None):
    tables = _ll_tables(coeff, axial, az)
    cols, sc = (x, y, z, l, m, n, ux, uy, uz), tuple(float(v) for v in (r0, z0, zmax, zmin, dphi))
    if _record_ll("pxf_wolterprimll", cols, sc, tables, 0., mask):
        return
    fused.flush()
    _call("pxf_wolterprimll", cols, sc + tables[1], mask)


def woltersecll(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, zmax, zmin, dphi, coeff, axial, az, mask=None):
    tables = _ll_tables(coeff, axial, az)
    cols, sc = (x, y, z, l, m, n, ux, uy, uz), tuple(float(v) for v in (r0, z0, psi, zmax, zmin, dphi))
    if _record_ll("pxf_woltersecll", cols, sc, tables, 0., mask):
        return
    fused.flush()
    _call("pxf_woltersecll", cols, sc + tables[1], mask)


def ellipsoidwoltll(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, S, zmax, zmin, dphi, coeff, axial, az, mask=None):
    tables = _ll_tables(coeff, axial, az)
    cols = (x, y, z, l, m, n, ux, uy, uz)
    if _record_ll("pxf_ellipsoidwoltll", cols, tuple(float(v) for v in (r0, z0, psi, zmax, zmin, dphi)), tables, S, mask):
        return
    fused.flush()
    _call("pxf_ellipsoidwoltll", cols, tuple(float(v) for v in (r0, z0, psi, S, zmax, zmin, dphi)) + tables[1], mask)


def wsprimary(x, y, z, l, m, n, ux, uy, uz, alpha, z0, psi, mask=None):
    _call("pxf_wsprimary", (x, y, z, l, m, n, ux, uy, uz), (float(alpha), float(z0), float(psi)), mask)


def wssecondary(x, y, z, l, m, n, ux, uy, uz, alpha, z0, psi, mask=None):
    _call("pxf_wssecondary", (x, y, z, l, m, n, ux, uy, uz), (float(alpha), float(z0), float(psi)), mask)


def wsprimaryback(x, y, z, l, m, n, ux, uy, uz, alpha, z0, psi, thick, mask=None):
    _call("pxf_wsprimaryback", (x, y, z, l, m, n, ux, uy, uz), tuple(float(v) for v in (alpha, z0, psi, thick)), mask)


def wssecondaryback(x, y, z, l, m, n, ux, uy, uz, alpha, z0, psi, thick, mask=None):
    _call("pxf_wssecondaryback", (x, y, z, l, m, n, ux, uy, uz), tuple(float(v) for v in (alpha, z0, psi, thick)), mask)


def wolter_params(r0, z0, psi=1.):
    o = [C.c_double() for _ in range(3)]
    _lib.lib().pxf_wolter_params(float(r0), float(z0), float(psi), *[C.byref(v) for v in o])
    return tuple(v.value for v in o)


# ---- zernsurf -------------------------------------------------------------------------------
def _zern_tables(coeff, rorder, aorder):
    ck = np.ascontiguousarray(np.asarray(coeff if not isinstance(coeff, torch.Tensor) else coeff.cpu(),
                                         dtype=np.float64))
    rk = np.ascontiguousarray(np.asarray(rorder), dtype=np.int32)   # f2py casts int64 -> int32 silently
    ak = np.ascontiguousarray(np.asarray(aorder), dtype=np.int32)
    if not (ck.ndim == rk.ndim == ak.ndim == 1 and len(ck) == len(rk) == len(ak)):
        raise ValueError("tracezern: coeff, rorder, aorder must be 1-D and of equal length")
    return (ck, rk, ak, ck.ctypes.data_as(_lib._DP), rk.ctypes.data_as(_lib._IP), ak.ctypes.data_as(_lib._IP))


def _record_zern(name, cols, tables, rad, nr, mask):
    """Inside a fused() block: record the Zernike trace into the pending program (True), or make
    room for it by flushing; False = run it as a stand-alone launch."""
    rec = fused.active()
    if rec is None:
        return False
    num, dev = _check_cols(cols, name)
    _mask_ptr(mask, num, dev, name)
    if num == 0:
        return True
    if rec.try_add_zern(name, cols, tables, rad, nr, mask):
        return True
    rec.flush()
    return rec.try_add_zern(name, cols, tables, rad, nr, mask)


def tracezern(x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad, mask=None):
    tables = _zern_tables(coeff, rorder, aorder)
    cols = (x, y, z, l, m, n, ux, uy, uz)
    if _record_zern("pxf_tracezern", cols, tables, rad, 0., mask):
        return
    ck, rk, ak, cp, rp, ap = tables
    fused.flush()
    _call("pxf_tracezern", cols, (cp, rp, ap, len(ck), float(rad)), mask)


def tracezernopd(opd, x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad, nr, mask=None):
    tables = _zern_tables(coeff, rorder, aorder)
    cols = (opd, x, y, z, l, m, n, ux, uy, uz)
    if _record_zern("pxf_tracezernopd", cols, tables, rad, nr, mask):
        return
    ck, rk, ak, cp, rp, ap = tables
    fused.flush()
    _call("pxf_tracezernopd", cols, (cp, rp, ap, len(ck), float(rad), float(nr)), mask)


def tracezernconic(opd, x, y, z, l, m, n, ux, uy, uz, R, K, coeff, rorder, aorder, rad, nr=0., mask=None):
    """EXTENSION (no f2py counterpart): conic of radius R, constant K plus a Zernike sag.  opd may be None."""
    cols = (x, y, z, l, m, n, ux, uy, uz)
    num, dev = _check_cols(cols if opd is None else (opd,) + cols, "pxf_tracezernconic")
    ck, rk, ak, cp, rp, ap = _zern_tables(coeff, rorder, aorder)
    mp = _mask_ptr(mask, num, dev, "pxf_tracezernconic")
    fused.flush()
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_tracezernconic(None if opd is None else opd.data_ptr(), *[t.data_ptr() for t in cols],
                                           num, float(R), float(K), cp, rp, ap, len(ck), float(rad), float(nr),
                                           mp, _stream(dev))
    _lib.check(rc, "pxf_tracezernconic")


def zernphase(opd, x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad, wave, mask=None):
    _check_cols((opd, x, y, z, l, m, n, ux, uy, uz), "pxf_zernphase")
    ck, rk, ak, cp, rp, ap = _zern_tables(coeff, rorder, aorder)
    fused.flush()
    _call("pxf_zernphase", (opd, x, y, l, m, n), (cp, rp, ap, len(ck), float(rad), float(wave)), mask)


def tracezernrot(x, y, z, l, m, n, ux, uy, uz, coeff1, rorder1, aorder1, coeff2, rorder2, aorder2, rad, rot,
                 mask=None):
    c1, r1, a1, c1p, r1p, a1p = _zern_tables(coeff1, rorder1, aorder1)
    c2, r2, a2, c2p, r2p, a2p = _zern_tables(coeff2, rorder2, aorder2)
    fused.flush()
    _call("pxf_tracezernrot", (x, y, z, l, m, n, ux, uy, uz),
          (c1p, r1p, a1p, len(c1), c2p, r2p, a2p, len(c2), float(rad), float(rot)), mask)


# ---- transformationsf -----------------------------------------------------------------------
def transform(x, y, z, l, m, n, ux, uy, uz, tx, ty, tz, rx, ry, rz, mask=None):
    _call("pxf_transform", (x, y, z, l, m, n, ux, uy, uz),
          tuple(float(v) for v in (tx, ty, tz, rx, ry, rz)), mask)


def itransform(x, y, z, l, m, n, ux, uy, uz, tx, ty, tz, rx, ry, rz, mask=None):
    _call("pxf_itransform", (x, y, z, l, m, n, ux, uy, uz),
          tuple(float(v) for v in (tx, ty, tz, rx, ry, rz)), mask)


def reflect(l, m, n, ux, uy, uz, mask=None):
    _call("pxf_reflect", (l, m, n, ux, uy, uz), (), mask)


def radgrat(x, y, l, m, n, wave, dpermm, order, mask=None):
    # f2py signature (x,y,l,m,n,wave,dpermm,order); the C entry takes wave before num
    num, dev = _check_cols((x, y, l, m, n), "pxf_radgrat")
    mp = _mask_ptr(mask, num, dev, "pxf_radgrat")
    rec = fused.active()
    if rec is not None and num:
        tail = (float(wave), float(dpermm), float(order))
        if rec.try_add("pxf_radgrat", (x, y, l, m, n), tail, mask):
            return
        rec.flush()
        if rec.try_add("pxf_radgrat", (x, y, l, m, n), tail, mask):
            return
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_radgrat(x.data_ptr(), y.data_ptr(), l.data_ptr(), m.data_ptr(), n.data_ptr(),
                                    float(wave), num, float(dpermm), float(order), mp, _stream(dev))
    _lib.check(rc, "pxf_radgrat")


def grat(x, y, l, m, n, d, order, wave, mask=None):
    """f2py signature grat(x,y,l,m,n,d,order,wave) (transformationsf.f95:277).  `order` and `wave`
    may each be a scalar or a per-ray column; only their product enters the loop."""
    num, dev = _check_cols((x, y, l, m, n), "pxf_grat")
    if isinstance(order, torch.Tensor) or isinstance(wave, torch.Tensor):
        ow = (torch.as_tensor(order, dtype=torch.float64, device=dev) *
              torch.as_tensor(wave, dtype=torch.float64, device=dev)).expand(num).contiguous()
        _call("pxf_grat", (l, m, n, ow), (float(d), 0.), mask, aux_last=True)
    else:
        _call("pxf_grat", (l, m, n), (float(d), float(order) * float(wave), None), mask)


def refract(l, m, n, ux, uy, uz, n1, n2, mask=None):
    _call("pxf_refract", (l, m, n, ux, uy, uz), (float(n1), float(n2)), mask)


def radgratw(x, y, l, m, n, wave, dpermm, order, mask=None):
    _call("pxf_radgratw", (x, y, l, m, n, wave), (float(dpermm), float(order)), mask)


# ---- reconstruct.f95 (f2py module `reconstruct`) -------------------------------------------------
def _fgrid(xdim, ydim, dev):
    """(xdim, ydim) float64 tensor laid out like a Fortran array (xdim fastest)."""
    return torch.empty((ydim, xdim), dtype=torch.float64, device=dev).t()


def _check_fgrids(what, *g):
    shp = g[0].shape
    for t in g:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.dim() == 2):
            raise TypeError("%s: grids must be 2-D float64 tensors on a ROCm device" % what)
        if t.shape != shp or t.device != g[0].device or not t.t().is_contiguous():
            raise ValueError("%s: grids must share one (xdim, ydim) shape and be Fortran-ordered "
                             "(see native.fgrid)" % what)
    return int(shp[0]), int(shp[1])


def fgrid(a, device=None):
    """Device copy of a 2-D array in the Fortran order the reconstruct entries expect."""
    t = torch.as_tensor(a, dtype=torch.float64)
    out = _fgrid(t.shape[0], t.shape[1], device if device is not None else
                 (t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())))
    out.copy_(t)
    return out


def southwellbin(x, y, l, m, binsize, xdim, ydim):
    """reconstruct.southwellbin(x,y,l,m,binsize,xdim,ydim) -> (xang, yang, phase)   reconstruct.f95:139"""
    fused.flush()
    num, dev = _check_cols((x, y, l, m), "pxf_southwellbin")
    xdim, ydim = int(xdim), int(ydim)
    if xdim < 1 or ydim < 1:
        raise ValueError("southwellbin: xdim, ydim must be positive")
    out = [_fgrid(xdim, ydim, dev) for _ in range(3)]
    h = _lib.lib()
    ws = torch.empty(max(1, h.pxf_southwellbin_workspace_bytes(xdim, ydim)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = h.pxf_southwellbin(x.data_ptr(), y.data_ptr(), l.data_ptr(), m.data_ptr(), num, float(binsize),
                                out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), xdim, ydim,
                                ws.data_ptr(), _stream(dev))
    _lib.check(rc, "pxf_southwellbin")
    return tuple(out)


def reconstruct(xang, yang, criteria, h, phase, maxiter, return_sweeps=False):
    """reconstruct.reconstruct(xang,yang,criteria,h,phase,maxiter) -> phasec   reconstruct.f95:1.
    xang, yang, phase are updated in place, as the f2py intent(inout) arrays are."""
    fused.flush()
    xdim, ydim = _check_fgrids("pxf_reconstruct", xang, yang, phase)
    dev = xang.device
    phasec = _fgrid(xdim, ydim, dev)
    sw = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_reconstruct(xang.data_ptr(), yang.data_ptr(), xdim, ydim, float(criteria), float(h),
                                        phase.data_ptr(), phasec.data_ptr(), int(maxiter), sw.data_ptr(),
                                        _stream(dev))
    _lib.check(rc, "pxf_reconstruct")
    return (phasec, int(sw.item())) if return_sweeps else phasec


reconstruct_mod = SimpleNamespace(southwellbin=southwellbin, reconstruct=reconstruct)
surfacesf = SimpleNamespace(flat=flat, flatopd=flatopd, conic=conic, conicopd=conicopd,
                            tracesphere=tracesphere, tracesphereopd=tracesphereopd, tracecyl=tracecyl,
                            tracecylopd=tracecylopd, cylconic=cylconic, paraxial=paraxial, paraxialy=paraxialy,
                            torus=torus, conicplus=conicplus, conicplusopd=conicplusopd, legsurf=legsurf)
woltsurf = SimpleNamespace(wolterprimary=wolterprimary, wolterprimaryopd=wolterprimaryopd,
                           woltersecondary=woltersecondary, spocone=spocone, woltersine=woltersine,
                           wolterprimll=wolterprimll, woltersecll=woltersecll, ellipsoidwoltll=ellipsoidwoltll,
                           wsprimary=wsprimary, wssecondary=wssecondary, wsprimaryback=wsprimaryback,
                           wssecondaryback=wssecondaryback)
zernsurf = SimpleNamespace(tracezern=tracezern, tracezernopd=tracezernopd, zernphase=zernphase,
                           tracezernrot=tracezernrot, tracezernconic=tracezernconic)
transformationsf = SimpleNamespace(transform=transform, itransform=itransform, reflect=reflect,
                                   radgrat=radgrat, radgratw=radgratw, grat=grat, refract=refract)
