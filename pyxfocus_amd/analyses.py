"""Drop-in for the focal-plane part of PyXFocus ``analyses.py`` (analyses.py:16-97, 118-187).

All per-ray sums are one fused HIP reduction (pxf_moments / pxf_central_moments); the scalar
algebra that turns sums into centroid / RMS / best-focus shift is the reference's, evaluated on
the host.  Functions return Python floats (this is the point where the stream is synchronised).
``weights`` may be None, a NumPy array or a tensor with one entry per ray.

In a sharded (multi-GPU) run pass ``group=`` (a torch.distributed process group, or True for the
default group): the twelve partial sums are all-reduced over RCCL before the scalar algebra, so
every rank returns the global metric without gathering any ray.
"""
import math

import numpy as np
import torch

from . import _fused as fused
from . import _lib, native

_ws_cache = {}


def _workspace(dev):
    # one scratch buffer per (device, stream): reductions issued on different streams may overlap
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _ws_cache:
        nbytes = _lib.lib().pxf_reduce_workspace_bytes()
        _ws_cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return _ws_cache[key]


def _weights(weights, num, dev):
    if weights is None:
        return None
    w = weights if isinstance(weights, torch.Tensor) else torch.as_tensor(np.asarray(weights, dtype=np.float64))
    w = w.to(device=dev, dtype=torch.float64).contiguous()
    if w.dim() != 1 or w.shape[0] != num:
        raise ValueError("weights must have one entry per ray")
    return w


def _allreduce(out, group):
    if group is None or group is False:
        return out
    import torch.distributed as dist
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=None if group is True else group)
    return out


def _moments(rays, weights, dirs=True, group=None):
    """The twelve weighted sums of include/pxf.h (PXF_NMOM) as a Python list of floats."""
    fused.flush()
    x, y = rays[1], rays[2]
    cols = (x, y) + ((rays[4], rays[5], rays[6]) if dirs else ())
    num, dev = native._check_cols(cols, "analyses")
    w = _weights(weights, num, dev)
    out = torch.zeros(_lib.PXF_NMOM, dtype=torch.float64, device=dev)
    p = [c.data_ptr() for c in cols] + [None] * (5 - len(cols))
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_moments(*p, None if w is None else w.data_ptr(), num, out.data_ptr(),
                                    _workspace(dev).data_ptr(), native._stream(dev))
    _lib.check(rc, "pxf_moments")
    return _allreduce(out, group).tolist()


def _central(rays, weights, cx, cy, group=None):
    fused.flush()
    x, y = rays[1], rays[2]
    num, dev = native._check_cols((x, y), "analyses")
    w = _weights(weights, num, dev)
    out = torch.zeros(4, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_central_moments(x.data_ptr(), y.data_ptr(), None if w is None else w.data_ptr(),
                                            num, float(cx), float(cy), out.data_ptr(),
                                            _workspace(dev).data_ptr(), native._stream(dev))
    _lib.check(rc, "pxf_central_moments")
    return _allreduce(out, group).tolist()


def centroid(rays, weights=None, group=None):
    """Centroid of the rays in the xy plane (analyses.py:16-22)."""
    s = _moments(rays, weights, dirs=False, group=group)
    return s[1] / s[0], s[2] / s[0]


def rmsCentroid(rays, weights=None, group=None):
    """RMS of rays from centroid in the xy plane (analyses.py:24-30) -- the spot-size metric."""
    cx, cy = centroid(rays, weights=weights, group=group)
    s = _central(rays, weights, cx, cy, group=group)
    return math.sqrt(s[1] / s[0])


def measureOPD(rays, point):
    """Distance in mm from every ray to `point` -- a 3-vector, or a 10-element single-ray list whose
    entries 1..3 are used (analyses.py:231-244).  Returns a device tensor."""
    fused.flush()
    x, y, z = rays[1:4]
    num, dev = native._check_cols((x, y, z), "measureOPD")
    p = [float(v) for v in (point[1:4] if len(point) == 10 else point[:3])]
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_dist3(x.data_ptr(), y.data_ptr(), z.data_ptr(), *p, out.data_ptr(), num,
                                  native._stream(dev))
    _lib.check(rc, "pxf_dist3")
    return out


def rmsPoint(rays, point, weights=None):
    """RMS of the rays about a point (analyses.py:32-44)."""
    d = measureOPD(rays, point)
    if weights is None:
        return math.sqrt(float((d * d).mean()))
    w = _weights(weights, d.shape[0], d.device)
    return math.sqrt(float((w * d * d).sum() / w.sum()))


def rmsX(rays, weights=None, group=None):
    """RMS from centroid in the X direction (analyses.py:46-51)."""
    cx, cy = centroid(rays, weights=weights, group=group)
    s = _central(rays, weights, cx, cy, group=group)
    return math.sqrt(s[2] / s[0])


def rmsY(rays, weights=None, group=None):
    """RMS from centroid in the Y direction (analyses.py:53-58)."""
    cx, cy = centroid(rays, weights=weights, group=group)
    s = _central(rays, weights, cx, cy, group=group)
    return math.sqrt(s[3] / s[0])


def rho(rays, weights=None, cent=False):
    """Distance from the centroid (cent=True) or the origin for all rays (analyses.py:60-71)."""
    fused.flush()
    x, y = rays[1:3]
    num, dev = native._check_cols((x, y), "rho")
    cx, cy = centroid(rays, weights=weights) if cent is True else (0., 0.)
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_rho(x.data_ptr(), y.data_ptr(), float(cx), float(cy), out.data_ptr(), num,
                                native._stream(dev))
    _lib.check(rc, "pxf_rho")
    return out


def rhocdf(rays, weights=None, cent=True):
    """Radial CDF of the ray distribution (analyses.py:73-86).  (argsort + cumsum on the device
    via torch: the one analysis here that genuinely needs a full sort.)"""
    r = rho(rays, weights=weights, cent=cent)
    w = torch.ones_like(r) if weights is None else _weights(weights, r.shape[0], r.device)
    r, ind = torch.sort(r, stable=True)
    cdf = torch.cumsum(w[ind], 0)
    return r, cdf / cdf.max()


def _kth(v, ranks, group=None):
    """Order statistics of a device column by radix select (no sort); returns Python floats.
    With `group` the column is this rank's shard of a distributed column and `ranks` are GLOBAL:
    per pass the 32 KB digit histogram is all-reduced, every rank picks the same bin, and after six
    passes all ranks hold the exact global values -- no ray leaves its GPU."""
    import ctypes as C
    num, dev = native._check_cols((v,), "kth_values")
    h = _lib.lib()
    ws = torch.empty(h.pxf_select_workspace_bytes(), dtype=torch.uint8, device=dev)
    out = torch.empty(len(ranks), dtype=torch.float64, device=dev)
    kk = (C.c_int64 * len(ranks))(*[int(k) for k in ranks])
    st = native._stream(dev)
    if group is None:
        with torch.cuda.device(dev):
            rc = h.pxf_kth_values(v.data_ptr(), num, kk, len(ranks), out.data_ptr(), ws.data_ptr(), st)
        _lib.check(rc, "pxf_kth_values")
        return out.tolist()
    import torch.distributed as dist
    grp = None if group is True else group
    off = [C.c_int64() for _ in range(3)]
    h.pxf_select_state_layout(*[C.byref(o) for o in off])
    hist = ws[off[0].value:off[0].value + off[1].value].view(torch.int64)
    nan = ws[off[2].value:off[2].value + 8].view(torch.int64)
    nk = len(ranks)
    with torch.cuda.device(dev):
        _lib.check(h.pxf_select_begin(kk, nk, ws.data_ptr(), st), "pxf_select_begin")
        for p in range(h.pxf_select_passes()):
            _lib.check(h.pxf_select_hist(v.data_ptr(), num, nk, ws.data_ptr(), p, st), "pxf_select_hist")
            dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=grp)
            if p == 0:
                dist.all_reduce(nan, op=dist.ReduceOp.SUM, group=grp)
            _lib.check(h.pxf_select_pick(nk, ws.data_ptr(), p, out.data_ptr(), st), "pxf_select_pick")
    return out.tolist()


def _wquantiles(v, w, qs, group=None):
    """r[argmin |cdf - q|] of the reference's sorted, weight-cumulated column for each q in qs
    (analyses.py:73-97), by weighted radix select on the device -- no sort, no 2.4 GB of sort
    temporaries per 1e8 rays.  group=: v, w are this rank's shard; weight histograms are all-reduced."""
    import ctypes as C
    num, dev = native._check_cols((v, w), "wquantiles")
    h = _lib.lib()
    ws = torch.empty(h.pxf_wselect_workspace_bytes(), dtype=torch.uint8, device=dev)
    out = torch.empty(len(qs), dtype=torch.float64, device=dev)
    qq = (C.c_double * len(qs))(*[float(q) for q in qs])
    st, nq = native._stream(dev), len(qs)
    if group is None:
        if num == 0:
            return [float("nan")] * nq
        with torch.cuda.device(dev):
            rc = h.pxf_wquantiles(v.data_ptr(), w.data_ptr(), num, qq, nq, out.data_ptr(), ws.data_ptr(), st)
        _lib.check(rc, "pxf_wquantiles")
        return out.tolist()
    import torch.distributed as dist
    grp = None if group is True else group
    off = [C.c_int64() for _ in range(4)]
    h.pxf_wselect_state_layout(*[C.byref(o) for o in off])
    hist = ws[off[0].value:off[0].value + off[1].value].view(torch.float64)
    prev = ws[off[2].value:off[2].value + 16].view(torch.int64)
    cnt = ws[off[3].value:off[3].value + 16].view(torch.int64)
    with torch.cuda.device(dev):
        _lib.check(h.pxf_wselect_begin(qq, nq, ws.data_ptr(), st), "pxf_wselect_begin")
        for p in range(h.pxf_select_passes()):
            _lib.check(h.pxf_wselect_hist(v.data_ptr(), w.data_ptr(), num, nq, ws.data_ptr(), p, st), "pxf_wselect_hist")
            dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=grp)
            _lib.check(h.pxf_wselect_pick(nq, ws.data_ptr(), p, st), "pxf_wselect_pick")
        _lib.check(h.pxf_wselect_neighbors(v.data_ptr(), num, nq, ws.data_ptr(), st), "pxf_wselect_neighbors")
        dist.all_reduce(prev, op=dist.ReduceOp.MAX, group=grp)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=grp)
        _lib.check(h.pxf_wselect_finish(nq, ws.data_ptr(), out.data_ptr(), st), "pxf_wselect_finish")
    return out.tolist()


_FAST_MEDIAN_MIN = 1 << 22      # below this the six-pass select is already launch-bound


def _radial_kth(rays, cx, cy, ranks, mode=0):
    """Order statistics of the radii about (cx, cy) (mode 1: of |y - cy|) in ONE pass over x, y
    (pxf_radial_kth: sample bracket + candidate select); None when the bracket did not hold."""
    import ctypes as C
    x, y = rays[1:3]
    num, dev = native._check_cols((x, y), "radial_kth")
    h = _lib.lib()
    ws = torch.empty(h.pxf_radial_kth_workspace_bytes(num), dtype=torch.uint8, device=dev)
    out = torch.empty(len(ranks), dtype=torch.float64, device=dev)
    ok = torch.zeros(1, dtype=torch.int32, device=dev)
    kk = (C.c_int64 * len(ranks))(*[int(k) for k in ranks])
    with torch.cuda.device(dev):
        rc = h.pxf_radial_kth(x.data_ptr(), y.data_ptr(), float(cx), float(cy), mode, num, kk, len(ranks),
                              out.data_ptr(), ok.data_ptr(), ws.data_ptr(), native._stream(dev))
    _lib.check(rc, "pxf_radial_kth")
    return out.tolist() if int(ok.item()) == 1 else None


def _rho_about(rays, cx, cy):
    x, y = rays[1:3]
    num, dev = native._check_cols((x, y), "rho")
    r = torch.empty_like(x)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().pxf_rho(x.data_ptr(), y.data_ptr(), float(cx), float(cy), r.data_ptr(), num,
                                      native._stream(dev)), "pxf_rho")
    return r


def hpd(rays, weights=None, cent=True, group=None):
    """HPD: twice the median radius from the centroid; weighted: the 25-75 % CDF span
    (analyses.py:88-97; like the reference, `cent` is ignored and the centroid is always used).
    Neither form sorts: the median is a radix select, the weighted span a weighted radix select
    (positive weights).  group=: the rays are this rank's shard; centroid by all-reduced moments,
    order statistics by all-reduced histograms -> the exact HPD of the whole bundle on every rank."""
    fused.flush()
    if weights is not None:
        cx, cy = centroid(rays, weights=weights, group=group)
        r = _rho_about(rays, cx, cy)
        w = _weights(weights, r.shape[0], r.device)
        hi, lo = _wquantiles(r, w, (.75, .25), group=group)
        return hi - lo
    if group is None:
        n = rays[1].shape[0]
        if n >= _FAST_MEDIAN_MIN:
            cx, cy = centroid(rays)
            got = _radial_kth(rays, cx, cy, [(n - 1) // 2, n // 2])
            if got is not None:
                return (got[0] + got[1]) * .5 * 2.
        r = rho(rays, weights=None, cent=True)
    else:
        import torch.distributed as dist
        cx, cy = centroid(rays, group=group)
        r = _rho_about(rays, cx, cy)
        cnt = torch.tensor([r.shape[0]], dtype=torch.int64, device=r.device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=None if group is True else group)
        n = int(cnt.item())
    if n == 0:
        return float("nan")
    lo, hi = _kth(r, [(n - 1) // 2, n // 2], group=group)        # np.median: mean of the two middle values
    return (lo + hi) * .5 * 2.


def hpdY(rays, weights=None):
    """HPD in the y direction: twice the median |y - <y>|; weighted: the 25-75 % CDF span
    (analyses.py:99-116)."""
    fused.flush()
    y = rays[2]
    cy = centroid(rays, weights=weights)[1]
    n = y.shape[0]
    if weights is not None:
        r = torch.abs(y - cy)
        hi, lo = _wquantiles(r, _weights(weights, n, r.device), (.75, .25))
        return hi - lo
    if n == 0:
        return float("nan")
    if n >= _FAST_MEDIAN_MIN:
        got = _radial_kth(rays, 0., cy, [(n - 1) // 2, n // 2], mode=1)
        if got is not None:
            return (got[0] + got[1]) * .5 * 2.
    lo, hi = _kth(torch.abs(y - cy), [(n - 1) // 2, n // 2])
    return (lo + hi) * .5 * 2.


def _griddata_device(px, py, pv, qx, qy, method):
    """scipy.interpolate.griddata((px, py), pv, (qx, qy), method) for method 'nearest' / 'linear' on the device
    (pxf_griddata): px, py, pv device columns, qx, qy NumPy arrays of one shape; returns a device tensor of that shape."""
    import ctypes as C
    if method not in ("nearest", "linear"):
        raise ValueError("on_device=True supports method='nearest' and 'linear' (cubic is scipy's Clough-Tocher "
                         "scheme, a global fit over the whole triangulation: use the default host path)")
    dev = px.device
    ok = torch.isfinite(px) & torch.isfinite(py)
    if not bool(ok.all()):
        px, py, pv = px[ok], py[ok], pv[ok]
    n = int(px.shape[0])
    q = np.ascontiguousarray(np.stack([np.asarray(qx, dtype=np.float64).ravel(), np.asarray(qy, dtype=np.float64).ravel()]))
    m = q.shape[1]
    out = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
    if n == 0 or m == 0:
        return out.reshape(np.shape(qx))
    x0, x1, y0, y1 = float(px.min()), float(px.max()), float(py.min()), float(py.max())
    g = int(min(4096, max(1, math.ceil(math.sqrt(n / 2.)))))          # ~2 points per cell
    wx, wy = max(x1 - x0, 1e-300), max(y1 - y0, 1e-300)
    icx, icy = g / wx, g / wy
    ci = torch.clamp(((px - x0) * icx).floor().to(torch.int64), 0, g - 1)
    cj = torch.clamp(((py - y0) * icy).floor().to(torch.int64), 0, g - 1)
    cid, order = torch.sort(cj * g + ci, stable=True)
    xs, ys, vs = px[order].contiguous(), py[order].contiguous(), pv.to(torch.float64)[order].contiguous()
    start = torch.searchsorted(cid, torch.arange(g * g + 1, device=dev, dtype=torch.int64)).to(torch.int32)
    ang = torch.arange(64, dtype=torch.float64, device=dev) * (2 * math.pi / 64)
    supt = torch.full((64,), -float("inf"), dtype=torch.float64, device=dev)
    for lo in range(0, n, 1 << 22):                       # 64 directional maxima in slabs of 4M rays (bounded temporaries)
        sl = slice(lo, lo + (1 << 22))
        supt = torch.maximum(supt, (torch.cos(ang)[:, None] * px[sl][None, :] + torch.sin(ang)[:, None] * py[sl][None, :]).max(dim=1).values)
    sup = (C.c_double * 64)(*supt.tolist())
    qd = torch.from_numpy(q).to(dev)
    ws = torch.empty(_lib.lib().pxf_griddata_workspace_bytes(m), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_griddata(xs.data_ptr(), ys.data_ptr(), vs.data_ptr(), n, start.data_ptr(), g, g, x0, y0, icx, icy,
                                     sup, qd[0].data_ptr(), qd[1].data_ptr(), m, 1 if method == "linear" else 0,
                                     out.data_ptr(), ws.data_ptr(), native._stream(dev))
    _lib.check(rc, "pxf_griddata")
    return out.reshape(np.shape(qx))


def interpolateVec(rays, I, Nx, Ny, xr=None, yr=None, method='linear', polar=False, interpVec=None, on_device=False):
    """Interpolate ray column `I` (or `interpVec`) onto an Nx x Ny grid spanning the rays' x/y extent
    (analyses.py:189-230).  Returns (grid values, dx, dy) as NumPy objects, like the reference.

    Default: the interpolation itself is `scipy.interpolate.griddata` -- a Delaunay triangulation from a third-party
    library the reference calls too -- on the HOST, on copies of the three columns involved: sharing the reference's
    own call is what makes the result identical to it.

    ``on_device=True`` (an addition): nothing but the Nx x Ny result leaves the GPU.  Methods 'nearest' and 'linear'
    (the Delaunay triangle around every grid node found by a per-node simplex walk, pxf_griddata); the values are a
    device tensor and equal scipy's to the rounding of the barycentric weights for rays in general position (rays on
    an exact regular grid are co-circular in fours: any Delaunay triangulation of them is as valid as Qhull's pick)."""
    if on_device:
        fused.flush()
        x, y = rays[1], rays[2]
        v = rays[I] if interpVec is None else interpVec
        if not isinstance(v, torch.Tensor):
            v = torch.as_tensor(np.asarray(v, dtype=np.float64), device=x.device)
        if xr is None:
            xr = [float(x.min()), float(x.max())]
            yr = [float(y.min()), float(y.max())]
        gridx, gridy = np.meshgrid(np.linspace(xr[0], xr[1], Nx), np.linspace(yr[0], yr[1], Ny))
        dx = np.diff(gridx)[0][0]
        dy = np.diff(np.transpose(gridy))[0][0]
        if polar is True:
            rho_ = torch.sqrt(x ** 2 + y ** 2)
            rhog = np.sqrt(gridx ** 2 + gridy ** 2)
            r1 = _griddata_device(rho_, torch.atan2(y, x) * rho_, v, rhog, rhog * np.arctan2(gridy, gridx), method)
            r2 = _griddata_device(rho_, torch.atan2(x, y) * rho_, v, rhog, rhog * np.arctan2(gridx, gridy), method)
            n1, n2 = torch.isnan(r1), torch.isnan(r2)                 # np.nanmedian of two: their mean, or the one that is not NaN
            res = torch.where(n1, r2, torch.where(n2, r1, (r1 + r2) / 2))
        else:
            res = _griddata_device(x, y, v, gridx, gridy, method)
        return res, dx, dy
    from scipy.interpolate import griddata
    fused.flush()
    x = rays[1].detach().cpu().numpy()
    y = rays[2].detach().cpu().numpy()
    if interpVec is None:
        interpVec = rays[I]
    v = interpVec.detach().cpu().numpy() if isinstance(interpVec, torch.Tensor) else np.asarray(interpVec)
    if xr is None:
        xr = [x.min(), x.max()]
        yr = [y.min(), y.max()]
    gridx, gridy = np.meshgrid(np.linspace(xr[0], xr[1], Nx), np.linspace(yr[0], yr[1], Ny))
    dx = np.diff(gridx)[0][0]
    dy = np.diff(np.transpose(gridy))[0][0]
    if polar is True:
        rho_ = np.sqrt(x ** 2 + y ** 2)
        theta1, theta2 = np.arctan2(y, x), np.arctan2(x, y)
        rhog = np.sqrt(gridx ** 2 + gridy ** 2)
        azg1, azg2 = rhog * np.arctan2(gridy, gridx), rhog * np.arctan2(gridx, gridy)
        res1 = griddata((rho_, theta1 * rho_), v, (rhog, azg1), method=method)
        res2 = griddata((rho_, theta2 * rho_), v, (rhog, azg2), method=method)
        res = np.nanmedian([res1, res2], axis=0)
    else:
        res = griddata((x, y), v, (gridx, gridy), method=method)
    return res, dx, dy


def wavefront(rays, Nx, Ny, method='cubic', polar=False, maxiter=10000, on_device=False):
    """Interpolate the beam slopes onto an Nx x Ny grid and integrate the wavefront (Southwell SOR on the
    device, `native.reconstruct`); rays assumed in the xy plane (analyses.py:305-334).  Returns
    (phase, xslope, yslope) as NumPy arrays with NaN where there are no rays.

    PARITY UNPINNED -- the reference function cannot run as shipped: it calls
    `reconstruct.reconstruct(y, x, 1e-12, dx, phase)` without the `maxiter` argument the Fortran requires
    (reconstruct.f95:1-3; its own southwell.py passes it) and pads with `utilities.imaging.man.padRect`, which
    is not vendored.  What it evidently intends is built from pinned pieces: `interpolateVec` (the reference's
    own griddata call), a one-cell border of the reconstructor's NaN sentinel (the `[1:-1, 1:-1]` the
    reference strips afterwards; `southwell.padArrays` does the same), criteria 1e-12, step dx, and
    `native.reconstruct`, which is bit-identical to the compiled reconstruct.f95.  One deliberate difference from
    the reference text: a cell counts as empty when EITHER slope is NaN (the reference marks `phase` where the x slope
    is NaN only, analyses.py:318-320); both slope maps come from the same griddata triangulation, so the two masks
    coincide whenever the inputs are finite.

    ``on_device=True`` (an addition; method 'linear' or 'nearest'): interpolation (pxf_griddata), border, masks and
    reconstruction all stay in HBM and the three results are device tensors."""
    from . import native
    if on_device:
        # the whole chain in HBM: device interpolation (method 'linear' / 'nearest'), the sentinel border and the NaN
        # bookkeeping as tensor operations, the reconstructor as before; returns DEVICE tensors
        ys, dx, dy = interpolateVec(rays, 5, Nx, Ny, method=method, polar=polar, on_device=True)
        xs, dx, dy = interpolateVec(rays, 4, Nx, Ny, method=method, polar=polar, on_device=True)
        pad = torch.nn.functional.pad
        xs, ys = pad(xs, (1, 1, 1, 1), value=float("nan")), pad(ys, (1, 1, 1, 1), value=float("nan"))
        bad = torch.isnan(xs) | torch.isnan(ys)
        hundred = torch.full_like(xs, 100.)
        gx, gy = native.fgrid(torch.where(bad, hundred, xs)), native.fgrid(torch.where(bad, hundred, ys))
        gp = native.fgrid(torch.where(bad, hundred, torch.zeros_like(xs)))
        out = native.reconstruct(gy, gx, 1e-12, float(dx), gp, int(maxiter))
        nanv = torch.full_like(xs, float("nan"))
        out = torch.where(out == 100, nanv, out)
        return out[1:-1, 1:-1], torch.where(bad, nanv, xs)[1:-1, 1:-1], torch.where(bad, nanv, ys)[1:-1, 1:-1]
    ys, dx, dy = interpolateVec(rays, 5, Nx, Ny, method=method, polar=polar)
    xs, dx, dy = interpolateVec(rays, 4, Nx, Ny, method=method, polar=polar)
    xs = np.pad(np.asarray(xs, dtype=np.float64), 1, constant_values=np.nan)
    ys = np.pad(np.asarray(ys, dtype=np.float64), 1, constant_values=np.nan)
    bad = np.isnan(xs) | np.isnan(ys)
    phase = np.zeros(xs.shape)
    phase[bad] = 100.
    xs[bad] = 100.
    ys[bad] = 100.
    dev = rays[1].device
    gx, gy, gp = native.fgrid(xs, dev), native.fgrid(ys, dev), native.fgrid(phase, dev)
    out = native.reconstruct(gy, gx, 1e-12, float(dx), gp, int(maxiter)).cpu().numpy().copy()
    out[out == 100] = np.nan
    xs[xs == 100] = np.nan
    ys[ys == 100] = np.nan
    return out[1:-1, 1:-1], xs[1:-1, 1:-1], ys[1:-1, 1:-1]


def measurePower(rays, Nx, Ny, method='linear'):
    """Radius of curvature of the wavefront along x and y from the central slices of the interpolated
    direction cosines; the sign equals the z transform that goes to focus (analyses.py:336-351).

    PARITY UNPINNED -- the reference indexes with `Ny/2` / `Nx/2` (true division: a TypeError under Python 3)
    and strips NaNs with the un-vendored `utilities.imaging.man.stripnans`; this is the same computation with
    integer division and a plain NaN filter, on the reference's own `interpolateVec` grids."""
    l, dx, dy = interpolateVec(rays, 4, Nx, Ny, method=method)
    m = interpolateVec(rays, 5, Nx, Ny, method=method)[0]
    xsl = l[Ny // 2]
    ysl = m[:, Nx // 2]
    xsl = xsl[~np.isnan(xsl)]
    ysl = ysl[~np.isnan(ysl)]
    xpow = 1 / np.gradient(xsl, dx)[min(Nx // 2, len(xsl) - 1)]
    ypow = 1 / np.gradient(ysl, dy)[min(Ny // 2, len(ysl) - 1)]
    return -xpow, -ypow


def radialGrad(x, y, hubscale, yaw, hubdist):
    """x and y derivatives of a radial-grating phase function; hubscale = lambda / (nn/mm)
    (analyses.py:403-428, including its quirk: the yaw-rotated y2 is overwritten by y + hubdist).
    Works on torch tensors (device) or NumPy arrays."""
    xp = torch if isinstance(x, torch.Tensor) else np
    x2 = x * math.cos(yaw) + y * math.sin(yaw)
    y2 = y + hubdist
    rho_ = xp.sqrt(x2 ** 2 + y2 ** 2)
    theta = (torch.atan2(-x2, y2) if xp is torch else np.arctan2(-x2, y2))
    gx = hubscale * (xp.cos(theta) / rho_)
    gy = -hubscale * (xp.sin(theta) / rho_)
    return gx, gy


def sellmeier(wave, B, C):
    """Refractive index from Sellmeier coefficients (analyses.py:430-446); host NumPy."""
    wave = np.reshape(np.asarray(wave, dtype=float), [np.size(wave), 1])
    B = np.reshape(np.asarray(B, dtype=float), [1, np.size(B)])
    C = np.reshape(np.asarray(C, dtype=float), [1, np.size(C)])
    num = np.dot(wave ** 2, B)
    den = wave ** 2 - C
    return np.sqrt(1 + np.sum(num / den, axis=1))


def _image_plane(s):
    W = s[0]
    bx = s[5] / W - (s[1] / W) * (s[3] / W)
    ax = s[7] / W - (s[3] / W) ** 2
    by = s[6] / W - (s[2] / W) * (s[4] / W)
    ay = s[8] / W - (s[4] / W) ** 2
    return -(bx + by) / (ax + ay)


def _y_plane(s):
    W = s[0]
    by = s[6] / W - (s[2] / W) * (s[4] / W)
    ay = s[8] / W - (s[4] / W) ** 2
    return -by / ay


def _x_plane(s):
    W = s[0]
    bx = s[5] / W - (s[1] / W) * (s[3] / W)
    ax = s[7] / W - (s[3] / W) ** 2
    return -bx / ax


def analyticImagePlane(rays, weights=None, group=None):
    """Best-focus shift by the analytic method of Ron Elsner's paper (analyses.py:118-133)."""
    return _image_plane(_moments(rays, weights, group=group))


def analyticYPlane(rays, weights=None, group=None):
    """analyses.py:135-144"""
    return _y_plane(_moments(rays, weights, group=group))


def analyticXPlane(rays, weights=None, group=None):
    """analyses.py:146-155"""
    return _x_plane(_moments(rays, weights, group=group))


# the scalar algebra behind each best-focus function, for callers that already hold the twelve sums (surfaces.focus)
analyticImagePlane._from_sums = _image_plane
analyticYPlane._from_sums = _y_plane
analyticXPlane._from_sums = _x_plane


def _focus_step(rays, dz, weights, nr, want_sums, group=None):
    """transform(rays,0,0,dz,0,0,0) -> flat(rays, nr=nr) -> the twelve sums of the result (want_sums), as ONE
    launch (pxf_transform_flat_moments): rays and sums have the bits of the three separate calls."""
    fused.flush()
    num, dev = native._check_cols(tuple(rays), "focus")
    w = _weights(weights, num, dev) if want_sums else None
    out = torch.zeros(_lib.PXF_NMOM, dtype=torch.float64, device=dev) if want_sums else None
    p = [c.data_ptr() for c in rays]
    if nr is None:
        p[0] = None
    with torch.cuda.device(dev):
        # (the wrapper negates its arguments: transformations.py:29; focus passes integer zeros, whose negation is +0)
        rc = _lib.lib().pxf_transform_flat_moments(*p, None if w is None else w.data_ptr(), num, 0., 0., -float(dz),
                                                   0., 0., 0., 0. if nr is None else float(nr),
                                                   None if out is None else out.data_ptr(),
                                                   _workspace(dev).data_ptr(), native._stream(dev))
    _lib.check(rc, "pxf_transform_flat_moments")
    return _allreduce(out, group).tolist() if want_sums else None


def indAngle(rays, ind=None, normal=None, _graze=0, out=None):
    """Incidence angle of the rays with either the current or a specified surface normal
    (analyses.py:164-182).  Returns a device tensor (of the `ind`-selected rays if ind is given).

    ``out=`` (an addition): the column to fill.  Inside a ``pyxfocus_amd.fused()`` block (or a
    ``compile_segments`` function) a call WITH ``out=`` -- and no `ind`, no `normal` -- is RECORDED as a program
    operation: the angles of the rays as they stand at that point of the trace are written by the fused kernel
    itself, and `out` is valid once the block has run (in a segmented program pass the segment's slice of one
    bundle-long column).  WITHOUT ``out=`` the call flushes the pending program and launches, so the tensor it
    returns is valid at once -- the reference's ``refl = reflectivity(grazeAngle(rays))`` (sector.py:592-612)
    inside a fused block reads real angles, never memory a later launch will fill."""
    l, m, n, ux, uy, uz = rays[4:10]
    num, dev = native._check_cols((l, m, n, ux, uy, uz), "indAngle")
    deferred_ok = out is not None          # only a caller who named the column has agreed to wait for the block
    if out is None:
        out = torch.empty_like(l)
    rec = fused.active()
    if rec is not None and deferred_ok and ind is None and normal is None and num:
        cols = (l, m, n, ux, uy, uz, out)
        if rec.try_add("pxf_indangle", cols, (float(_graze),), None):
            return out
        rec.flush()
        if rec.try_add("pxf_indangle", cols, (float(_graze),), None):
            return out
    fused.flush()
    if normal is not None:
        ux, uy, uz = [torch.full_like(l, float(v)) for v in normal]
    with torch.cuda.device(dev):
        rc = _lib.lib().pxf_indangle(l.data_ptr(), m.data_ptr(), n.data_ptr(), ux.data_ptr(), uy.data_ptr(),
                                     uz.data_ptr(), out.data_ptr(), num, _graze, native._stream(dev))
    _lib.check(rc, "pxf_indangle")
    if ind is None:
        return out
    if isinstance(ind, tuple):
        ind = ind[0]
    sel = ind if isinstance(ind, torch.Tensor) else torch.as_tensor(np.asarray(ind))
    return out[sel.to(dev)]


def grazeAngle(rays, ind=None, out=None):
    """Graze angle of the rays with the current surface normal (analyses.py:184-187); `out=` as indAngle."""
    return indAngle(rays, ind=ind, _graze=1, out=out)
