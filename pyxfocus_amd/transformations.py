"""Drop-in for PyXFocus ``transformations.py`` on ROCm tensors.

Same function names, argument order, defaults and return conventions as the reference
(transformations.py:7-289).  ``rays`` is the reference's ten-element list
``[opd,x,y,z,l,m,n,ux,uy,uz]`` whose elements are 1-D contiguous ``torch.float64`` tensors on one
MI355X; calls mutate them in place and return ``None`` unless the reference returns a new object
(``vignette``, ``applyT``, ``applyTPos``, ``copy_rays``).

``ind`` may be what the reference accepts for fancy indexing -- a boolean mask, an integer index
array, or an ``np.where`` tuple -- as NumPy arrays or torch tensors.  The reference gathers the
selected rays, runs the Fortran on the copy and scatters back (transformations.py:20-27); here
``ind`` becomes a byte mask and the kernel simply skips unselected rays, which gives the same
result (repeated indices included: every copy computes the same value).

Only the 4x4 ``coords`` bookkeeping runs on the host (16 numbers per call); everything that
touches a ray is a HIP kernel behind include/pxf.h.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _fused as fused
from . import _lib, native
from .native import transformationsf as tran

__all__ = ["copy_rays", "transform", "itransform", "steerX", "steerY", "pointTo", "reflect", "refract", "grat",
           "radgrat",
           "vignette", "newCoords", "rotationM", "translationM", "applyT", "applyTPos"]


# ---- ind handling ----------------------------------------------------------------------------
def _as_tensor(a, dev):
    if isinstance(a, torch.Tensor):
        return a.to(dev)
    return torch.as_tensor(np.asarray(a), device=dev)


def ind_to_mask(ind, num, dev):
    """None | bool mask | int index array | np.where tuple  ->  None | uint8/bool device mask."""
    if ind is None:
        return None
    if isinstance(ind, tuple):  # np.where(...) / torch.where(...) result
        if len(ind) != 1:
            raise IndexError("too many indices for array: ray columns are 1-dimensional")
        ind = ind[0]
    t = _as_tensor(ind, dev)
    if t.dtype == torch.bool:
        if t.dim() != 1 or t.shape[0] != num:
            raise IndexError("boolean index did not match ray count %d" % num)
        return t.contiguous()
    if t.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
        idx = t.to(torch.int64).contiguous().reshape(-1)
        if idx.numel() and (int(idx.max()) >= num or int(idx.min()) < -num):
            raise IndexError("index out of bounds for %d rays" % num)
        mask = torch.empty(num, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().pxf_mask_from_index(idx.data_ptr(), idx.numel(), num, mask.data_ptr(),
                                                native._stream(dev))
        _lib.check(rc, "pxf_mask_from_index")
        return mask
    raise IndexError("arrays used as indices must be of integer (or boolean) type")


def _dev(rays):
    return rays[1].device


def copy_rays(rays):
    return [rays[i].clone() for i in range(len(rays))]


# ---- frame changes ---------------------------------------------------------------------------
def transform(rays, dx, dy, dz, rx, ry, rz, ind=None, coords=None):
    """Coordinate transformation. translations are done first, then Rx,Ry,Rz
    coords[0] - global to local rotation only
    coords[1] - global to local rotations and translations
    coords[2] - local to global rotations only
    coords[3] - local to global rotations and translations
    (transformations.py:11-44)
    """
    x, y, z, l, m, n, ux, uy, uz = rays[1:]
    mask = ind_to_mask(ind, x.shape[0], _dev(rays))
    tran.transform(x, y, z, l, m, n, ux, uy, uz, -dx, -dy, -dz, -rx, -ry, -rz, mask=mask)
    if coords is not None:
        update_coords(coords, dx, dy, dz, rx, ry, rz)
    return


def update_coords(coords, dx, dy, dz, rx, ry, rz):
    """The 4x4 bookkeeping of one transform() (transformations.py:32-42), without touching any ray."""
    rotm = rotationM(rx, ry, rz)
    tranm = translationM(dx, dy, dz)
    rotmi = rotationM(rx, ry, rz, inverse=True)
    tranmi = translationM(-dx, -dy, -dz)
    coords[0] = np.dot(rotm, coords[0])
    coords[1] = np.dot(np.dot(rotm, tranm), coords[1])
    coords[2] = np.dot(coords[2], rotmi)
    coords[3] = np.dot(coords[3], np.dot(tranmi, rotmi))


def itransform(rays, dx, dy, dz, rx, ry, rz, coords=None, ind=None):
    """Inverse of coordinate transformations. -rz,-ry,-rx then translations.
    (transformations.py:47-77; note the keyword order coords, ind)"""
    x, y, z, l, m, n, ux, uy, uz = rays[1:]
    mask = ind_to_mask(ind, x.shape[0], _dev(rays))
    tran.itransform(x, y, z, l, m, n, ux, uy, uz, -dx, -dy, -dz, -rx, -ry, -rz, mask=mask)
    if coords is not None:
        rotm = rotationM(rx, ry, rz, inverse=True)
        tranm = translationM(-dx, -dy, -dz)
        rotmi = rotationM(rx, ry, rz)
        tranmi = translationM(dx, dy, dz)
        coords[0] = np.dot(rotm, coords[0])
        coords[1] = np.dot(np.dot(tranm, rotm), coords[1])
        coords[2] = np.dot(coords[2], rotmi)
        coords[3] = np.dot(coords[3], np.dot(rotmi, tranmi))
    return


def _mean(col, group=None):
    from . import analyses
    s = analyses._moments([None, col, col], None, dirs=False, group=group)   # np.mean(col): S x / S 1
    return s[1] / s[0]


def steerY(rays, coords=None, group=None):
    """Rotate reference frame for zero mean y tilt (transformations.py:79-83).
    group= (ray-sharded runs): the mean is all-reduced, every rank applies the same rotations."""
    while abs(_mean(rays[5], group)) > 1e-6:
        transform(rays, 0, 0, 0, -_mean(rays[5], group), 0, 0, coords=coords)
    return


def steerX(rays, coords=None, group=None):
    """Rotate reference frame for zero mean x tilt (transformations.py:85-89); group= as steerY."""
    while abs(_mean(rays[4], group)) > 1e-6:
        transform(rays, 0, 0, 0, 0, -_mean(rays[4], group), 0, coords=coords)
    return


def pointTo(rays, x0, y0, z0, reverse=-1.):
    """Direct all ray direction cosines toward (x0,y0,z0); reverse=1. points them away.
    The reference rebinds rays[4:7] to new arrays (transformations.py:96-99); so do we."""
    fused.flush()
    x, y, z = rays[1:4]
    num, dev = native._check_cols((x, y, z), "pointTo")
    out = [torch.empty_like(x) for _ in range(3)]
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_pointto(x.data_ptr(), y.data_ptr(), z.data_ptr(), *[o.data_ptr() for o in out],
                                    num, float(x0), float(y0), float(z0), float(reverse), native._stream(dev))
    _lib.check(rc, "pxf_pointto")
    rays[4], rays[5], rays[6] = out
    return


def reflect(rays, ind=None):
    """Reflect rays based on surface normal (transformations.py:102-112)"""
    l, m, n, ux, uy, uz = rays[4:]
    tran.reflect(l, m, n, ux, uy, uz, mask=ind_to_mask(ind, l.shape[0], _dev(rays)))
    return


def refract(rays, n1, n2):
    """Refract rays at the current surface normal from index n1 into index n2
    (transformations.py:114-121)."""
    l, m, n, ux, uy, uz = rays[4:]
    tran.refract(l, m, n, ux, uy, uz, n1, n2)
    return


def grat(rays, d, order, wave, ind=None):
    """Linear grating with groove direction +y and period d; evanescence zeroes the direction
    cosines (transformations.py:200-212).  The reference's Fortran declares `order` and `wave`
    as per-ray arrays while its wrapper documents scalars; here each may be a
    scalar or an array with one entry per ray."""
    x, y, z, l, m, n = rays[1:7]
    dev = _dev(rays)
    mask = ind_to_mask(ind, x.shape[0], dev)
    arr = [isinstance(v, (np.ndarray, torch.Tensor, list, tuple)) for v in (order, wave)]
    if any(arr):
        order, wave = [_as_tensor(v, dev).to(torch.float64) if a else v for v, a in zip((order, wave), arr)]
    tran.grat(x, y, l, m, n, d, order, wave, mask=mask)
    return


def radgrat(rays, dpermm, order, wave, ind=None):
    """Infinite radial grating in the x-y plane, grooves converging at the origin hub
    (transformations.py:124-172).  A scalar `wave` runs radgrat (sign of n kept), an array
    `wave` (np.ndarray or tensor, one wavelength per ray) runs radgratW (sign of n from sign(y))
    -- the reference dispatches on type(wave)==np.ndarray (:146-149)."""
    x, y, z, l, m, n = rays[1:7]
    mask = ind_to_mask(ind, x.shape[0], _dev(rays))
    if isinstance(wave, (np.ndarray, torch.Tensor)):
        w = _as_tensor(wave, _dev(rays)).to(torch.float64).contiguous()
        if w.dim() != 1 or w.shape[0] != x.shape[0]:
            raise ValueError("radgrat: wave array must have one entry per ray")
        tran.radgratw(x, y, l, m, n, w, dpermm, order, mask=mask)
    else:
        tran.radgrat(x, y, l, m, n, wave, dpermm, order, mask=mask)
    return


# ---- vignette --------------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def vignette(rays, ind=None):
    """Remove vignetted rays from memory; ind is array of "good" indices, all others are removed.
    Default: keep rays with l^2+m^2+n^2 > .1 (transformations.py:214-225).  Returns a NEW list."""
    fused.flush()
    cols = list(rays[:10])
    num, dev = native._check_cols(cols, "vignette")
    h = _lib.lib()
    st = native._stream(dev)
    if isinstance(ind, tuple):
        if len(ind) != 1:
            raise IndexError("too many indices for array: ray columns are 1-dimensional")
        ind = ind[0]
    with torch.cuda.device(dev):
        idx = None
        if ind is None:
            mask = torch.empty(num, dtype=torch.uint8, device=dev)
            _lib.check(h.pxf_vignette_mask(rays[4].data_ptr(), rays[5].data_ptr(), rays[6].data_ptr(), num,
                                           mask.data_ptr(), st), "pxf_vignette_mask")
        else:
            t = _as_tensor(ind, dev)
            if t.dtype == torch.bool:
                if t.dim() != 1 or t.shape[0] != num:
                    raise IndexError("boolean index did not match ray count %d" % num)
                mask = t.contiguous()
            else:
                if t.dtype not in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
                    raise IndexError("arrays used as indices must be of integer (or boolean) type")
                idx = t.to(torch.int64).contiguous().reshape(-1)
        if idx is not None:      # fancy indexing with an integer array: order/repeats preserved
            if idx.numel() and (int(idx.max()) >= num or int(idx.min()) < -num):
                raise IndexError("index out of bounds for %d rays" % num)
            out = [torch.empty(idx.numel(), dtype=torch.float64, device=dev) for _ in cols]
            _lib.check(h.pxf_gather(_ptr_array(cols), _ptr_array(out), len(cols), idx.data_ptr(),
                                    idx.numel(), num, st), "pxf_gather")
            return out
        ws = torch.empty(max(1, h.pxf_compact_workspace_bytes(num)), dtype=torch.uint8, device=dev)
        total = torch.empty(1, dtype=torch.int64, device=dev)
        _lib.check(h.pxf_compact_scan(mask.data_ptr(), num, ws.data_ptr(), total.data_ptr(), st),
                   "pxf_compact_scan")
        k = int(total.item())          # the one unavoidable sync: the result length is data-dependent
        out = [torch.empty(k, dtype=torch.float64, device=dev) for _ in cols]
        if k:
            _lib.check(h.pxf_compact_scatter(_ptr_array(cols), _ptr_array(out), len(cols), num,
                                             mask.data_ptr(), ws.data_ptr(), st), "pxf_compact_scatter")
    return out


# ---- 4x4 bookkeeping (host; 16 numbers) --------------------------------------------------------
def _rotation_matrix(angle, axis):
    """Rotation by `angle` about unit axis 0/1/2 as a 4x4 (Rodrigues form, the same construction
    the reference's transformMod.rotation_matrix uses: R = c I + (1-c) d d^T + s [d]x)."""
    d = np.zeros(3)
    d[axis] = 1.0
    c, s = math.cos(angle), math.sin(angle)
    R = np.diag([c, c, c])
    R += np.outer(d, d) * (1.0 - c)
    d = d * s
    R += np.array([[0.0, -d[2], d[1]], [d[2], 0.0, -d[0]], [-d[1], d[0], 0.0]])
    M = np.identity(4)
    M[:3, :3] = R
    return M


def newCoords():
    """Return identity matrices to establish a coordinate system (transformations.py:228-232).
    Unlike the reference (four references to ONE matrix) these are four separate copies; the
    reference only ever rebinds the entries, so the difference is unobservable."""
    return [np.identity(4) for _ in range(4)]


def rotationM(rx, ry, rz, inverse=False):
    """Rotation matrix, rotations applied in X,Y,Z order, angles negated to be consistent with
    transform (transformations.py:234-248)."""
    if inverse is True:
        rx, ry, rz = -rx, -ry, -rz
    r1 = _rotation_matrix(-rx, 0)
    r2 = _rotation_matrix(-ry, 1)
    r3 = _rotation_matrix(-rz, 2)
    if inverse is True:
        return np.dot(r1, np.dot(r2, r3))
    return np.dot(r3, np.dot(r2, r1))


def translationM(tx, ty, tz):
    """Translation matrix with negated values, consistent with transform (transformations.py:250-255)."""
    M = np.identity(4)
    M[:3, 3] = [-tx, -ty, -tz]
    return M


def _m16(mat):
    a = np.ascontiguousarray(np.asarray(mat, dtype=np.float64))
    if a.shape != (4, 4):
        raise ValueError("coords entries must be 4x4 matrices")
    return a, a.ctypes.data_as(_lib._DP)


def skew(vec):
    """Skew-symmetric cross-product matrix of a 3-vector (transformations.py:291-292)."""
    return np.array([[0, -vec[2], vec[1]], [vec[2], 0, -vec[0]], [-vec[1], vec[0], 0]])


def applyT(rays, coords, inverse=False):
    """Apply transformation matrix to raylist; only rotations to direction cosines and normals.
    inverse=True goes back to the global system.  Returns a NEW list sharing rays[0]
    (transformations.py:257-280)."""
    fused.flush()
    i = 2 if inverse is True else 0
    cols = list(rays[1:10])
    num, dev = native._check_cols(cols, "applyT")
    mpos, ppos = _m16(coords[i + 1])
    mdir, pdir = _m16(coords[i])
    out = [torch.empty_like(c) for c in cols]
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_applyt(*[c.data_ptr() for c in cols], *[o.data_ptr() for o in out], num,
                                   ppos, pdir, native._stream(dev))
    _lib.check(rc, "pxf_applyt")
    return [rays[0]] + out


def applyTPos(x, y, z, coords, inverse=False):
    """Apply transformation to list of points (transformations.py:282-289).  Returns a (3, N) tensor."""
    i = 2 if inverse is True else 0
    if not isinstance(x, torch.Tensor):
        # a handful of host points (scalars or NumPy arrays, e.g. applyTPos(0,0,0,coords) to locate
        # an origin): the reference's own product on the host, (3, N) ndarray
        pos = np.array([np.ravel(np.asarray(v, dtype=float)) for v in (x, y, z)])
        pos = np.vstack([pos, np.ones(pos.shape[1])])
        return np.dot(np.asarray(coords[i + 1], dtype=float), pos)[:3]
    fused.flush()
    num, dev = native._check_cols((x, y, z), "applyTPos")
    mpos, ppos = _m16(coords[i + 1])
    out = torch.empty((3, num), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_applytpos(x.data_ptr(), y.data_ptr(), z.data_ptr(), out[0].data_ptr(),
                                      out[1].data_ptr(), out[2].data_ptr(), num, ppos, native._stream(dev))
    _lib.check(rc, "pxf_applytpos")
    return out
