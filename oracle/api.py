"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends for the two CPU checkers.

* ``load_port()``  -> oracle/libpxf_oracle[_omp].so, our plain-C restatement (pxf_oracle.c).
* ``load_ref()``   -> oracle/_ref/libpxfref_*.so, the reference's own Fortran compiled in place
                      by oracle/Makefile (Fortran ABI: every argument by reference, symbol
                      ``name_``, ``num`` an int32).

Both return an object with four namespaces that mimic the f2py modules the reference's Python
imports (``surfacesf``, ``woltsurf``, ``zernsurf``, ``transformationsf``; compiletrace.sh:1-6), with
the f2py call signatures: SoA float64 columns mutated in place, ``num``
implied by ``len(x)``.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(quiet=True):
    """Run oracle/Makefile (C restatement always; _ref only where /root/reference exists)."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _f64(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.ndim == 1):
        raise TypeError("ray columns must be 1-D C-contiguous float64 arrays (f2py intent(inout))")
    return a.ctypes.data_as(_dp)


def _cols(*arrs):
    n = len(arrs[0])
    for a in arrs:
        if len(a) != n:
            raise ValueError("ray columns differ in length")
    return [_f64(a) for a in arrs], n


def _g64(a):
    return a.ctypes.data_as(_dp)


def _grid2(*arrs):
    """Shape check for the Fortran-ordered 2-D arrays of reconstruct.f95 (f2py would raise/copy)."""
    shp = arrs[0].shape
    for a in arrs:
        if a.ndim != 2 or a.shape != shp or a.dtype != np.float64 or not a.flags.f_contiguous:
            raise TypeError("expected Fortran-contiguous float64 arrays of one (xdim, ydim) shape")
    return int(shp[0]), int(shp[1])


def _i32(a):
    a = np.ascontiguousarray(np.asarray(a), dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _f64in(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a, a.ctypes.data_as(_dp)


# --------------------------------------------------------------------------------------------
def load_port(omp=False):
    path = os.path.join(HERE, "libpxf_oracle_omp.so" if omp else "libpxf_oracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    d, i64, i32 = C.c_double, C.c_int64, C.c_int32

    def sig(name, *argtypes, restype=None):
        f = getattr(lib, name)
        f.argtypes = list(argtypes)
        f.restype = restype
        return f

    P = _dp
    f_flat = sig("orc_flat", *[P] * 9, i64)
    f_flatopd = sig("orc_flatopd", *[P] * 10, i64, d)
    f_conic = sig("orc_conic", *[P] * 9, i64, d, d)
    f_conicopd = sig("orc_conicopd", *[P] * 10, i64, d, d, d)
    f_wp = sig("orc_wolterprimary", *[P] * 9, i64, d, d, d)
    f_wpo = sig("orc_wolterprimaryopd", *[P] * 10, i64, d, d, d, d)
    f_ws = sig("orc_woltersecondary", *[P] * 9, i64, d, d, d)
    f_tz = sig("orc_tracezern", *[P] * 9, i64, P, _ip, _ip, i32, d, restype=C.c_int)
    f_tzo = sig("orc_tracezernopd", *[P] * 10, i64, P, _ip, _ip, i32, d, d, restype=C.c_int)
    f_zs = sig("orc_zernset", d, d, _ip, _ip, i32, P, P, P)
    f_tr = sig("orc_transform", *[P] * 9, i64, *[d] * 6)
    f_itr = sig("orc_itransform", *[P] * 9, i64, *[d] * 6)
    f_refl = sig("orc_reflect", *[P] * 6, i64)
    f_rg = sig("orc_radgrat", *[P] * 5, d, i64, d, d)
    f_rgw = sig("orc_radgratw", *[P] * 6, i64, d, d)
    f_par = sig("orc_wolter_params", d, d, d, P, P, P)
    f_spo = sig("orc_spocone", *[P] * 9, i64, d, d)
    f_grat = sig("orc_grat", *[P] * 5, i64, d, P, P)
    f_sph = sig("orc_tracesphere", *[P] * 9, i64, d)
    f_spho = sig("orc_tracesphereopd", *[P] * 10, i64, d, d)
    f_cyl = sig("orc_tracecyl", *[P] * 9, i64, d)
    f_cylo = sig("orc_tracecylopd", *[P] * 10, i64, d, d)
    f_cc = sig("orc_cylconic", *[P] * 9, i64, d, d)
    f_px = sig("orc_paraxial", *[P] * 4, i64, d, C.c_int)
    f_rf = sig("orc_refract", *[P] * 6, i64, d, d)
    f_wsin = sig("orc_woltersine", *[P] * 9, i64, d, d, d, d)
    f_pll = sig("orc_wolterprimll", *[P] * 9, i64, d, d, d, d, d, P, _ip, _ip, i32)
    f_sll = sig("orc_woltersecll", *[P] * 9, i64, d, d, d, d, d, d, P, _ip, _ip, i32)
    f_ell = sig("orc_ellipsoidwoltll", *[P] * 9, i64, d, d, d, d, d, d, d, P, _ip, _ip, i32)
    f_wsp = sig("orc_wsprimary", *[P] * 9, i64, d, d, d)
    f_wss = sig("orc_wssecondary", *[P] * 9, i64, d, d, d)
    f_leg = sig("orc_legendre", d, i32, P, P)
    f_wspb = sig("orc_wsprimaryback", *[P] * 9, i64, d, d, d, d)
    f_wssb = sig("orc_wssecondaryback", *[P] * 9, i64, d, d, d, d)
    f_tor = sig("orc_torus", *[P] * 9, i64, d, d)
    f_cp = sig("orc_conicplus", *[P] * 9, i64, d, d, P, i32)
    f_cpo = sig("orc_conicplusopd", *[P] * 10, i64, d, d, P, i32, d)
    f_lgs = sig("orc_legsurf", *[P] * 5, i64, d, d, d, P, _ip, _ip, i32)
    f_zph = sig("orc_zernphase", *[P] * 6, i64, P, _ip, _ip, i32, d, d, restype=C.c_int)
    f_zrot = sig("orc_tracezernrot", *[P] * 9, i64, P, _ip, _ip, i32, P, _ip, _ip, i32, d, d, restype=C.c_int)

    def flat(x, y, z, l, m, n, ux, uy, uz):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_flat(*p, num)

    def flatopd(x, y, z, l, m, n, ux, uy, uz, opd, nr):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz, opd)
        f_flatopd(*p, num, nr)

    def conic(x, y, z, l, m, n, ux, uy, uz, R, K):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_conic(*p, num, R, K)

    def conicopd(opd, x, y, z, l, m, n, ux, uy, uz, R, K, nr):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        f_conicopd(*p, num, R, K, nr)

    def wolterprimary(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_wp(*p, num, r0, z0, psi)

    def wolterprimaryopd(opd, x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, nr):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        f_wpo(*p, num, r0, z0, psi, nr)

    def woltersecondary(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_ws(*p, num, r0, z0, psi)

    def tracezern(x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        ck, cp = _f64in(coeff)
        rk, rp = _i32(rorder)
        ak, ap = _i32(aorder)
        if f_tz(*p, num, cp, rp, ap, len(ck), rad) != 0:
            raise ValueError("tracezern: (rorder, aorder) outside the reference's radial table")

    def tracezernopd(opd, x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad, nr):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        ck, cp = _f64in(coeff)
        rk, rp = _i32(rorder)
        ak, ap = _i32(aorder)
        if f_tzo(*p, num, cp, rp, ap, len(ck), rad, nr) != 0:
            raise ValueError("tracezernopd: (rorder, aorder) outside the reference's radial table")

    def zernset(rho, theta, rorder, aorder):
        rk, rp = _i32(rorder)
        ak, ap = _i32(aorder)
        out = [np.zeros(len(rk)) for _ in range(3)]
        f_zs(rho, theta, rp, ap, len(rk), *[_f64(o) for o in out])
        return out

    def transform(x, y, z, l, m, n, ux, uy, uz, tx, ty, tz, rx, ry, rz):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_tr(*p, num, tx, ty, tz, rx, ry, rz)

    def itransform(x, y, z, l, m, n, ux, uy, uz, tx, ty, tz, rx, ry, rz):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_itr(*p, num, tx, ty, tz, rx, ry, rz)

    def reflect(l, m, n, ux, uy, uz):
        p, num = _cols(l, m, n, ux, uy, uz)
        f_refl(*p, num)

    def radgrat(x, y, l, m, n, wave, dpermm, order):
        p, num = _cols(x, y, l, m, n)
        f_rg(*p, float(wave), num, dpermm, order)

    def radgratw(x, y, l, m, n, wave, dpermm, order):
        p, num = _cols(x, y, l, m, n, wave)
        f_rgw(*p, num, dpermm, order)

    def spocone(x, y, z, l, m, n, ux, uy, uz, R0, tg):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_spo(*p, num, R0, tg)

    def grat(x, y, l, m, n, d_, order, wave):
        p, num = _cols(x, y, l, m, n)
        ok, op_ = _f64in(np.broadcast_to(np.asarray(order, dtype=np.float64), (num,)))
        wk, wp_ = _f64in(np.broadcast_to(np.asarray(wave, dtype=np.float64), (num,)))
        f_grat(*p, num, d_, op_, wp_)

    def tracesphere(x, y, z, l, m, n, ux, uy, uz, rad):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_sph(*p, num, rad)

    def tracesphereopd(opd, x, y, z, l, m, n, ux, uy, uz, rad, nr):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        f_spho(*p, num, rad, nr)

    def tracecyl(x, y, z, l, m, n, ux, uy, uz, rad):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_cyl(*p, num, rad)

    def tracecylopd(opd, x, y, z, l, m, n, ux, uy, uz, rad, nr):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        f_cylo(*p, num, rad, nr)

    def cylconic(x, y, z, l, m, n, ux, uy, uz, rad, k):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_cc(*p, num, rad, k)

    def paraxial(x, y, z, l, m, n, ux, uy, uz, F):
        p, num = _cols(x, y, l, m)
        f_px(*p, num, F, 0)

    def paraxialy(x, y, z, l, m, n, ux, uy, uz, F):
        p, num = _cols(x, y, l, m)
        f_px(*p, num, F, 1)

    def refract(l, m, n, ux, uy, uz, n1, n2):
        p, num = _cols(l, m, n, ux, uy, uz)
        f_rf(*p, num, n1, n2)

    def woltersine(x, y, z, l, m, n, ux, uy, uz, r0, z0, amp, freq):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_wsin(*p, num, r0, z0, amp, freq)

    def _ll(coeff, axial, az):
        ck, cp = _f64in(coeff)
        ak, ap = _i32(axial)
        zk, zp = _i32(az)
        return (ck, ak, zk), (cp, ap, zp, len(ck))

    def wolterprimll(x, y, z, l, m, n, ux, uy, uz, r0, z0, zmax, zmin, dphi, coeff, axial, az):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        keep, t = _ll(coeff, axial, az)
        f_pll(*p, num, r0, z0, zmax, zmin, dphi, *t)

    def woltersecll(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, zmax, zmin, dphi, coeff, axial, az):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        keep, t = _ll(coeff, axial, az)
        f_sll(*p, num, r0, z0, psi, zmax, zmin, dphi, *t)

    def ellipsoidwoltll(x, y, z, l, m, n, ux, uy, uz, r0, z0, psi, S, zmax, zmin, dphi, coeff, axial, az):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        keep, t = _ll(coeff, axial, az)
        f_ell(*p, num, r0, z0, psi, S, zmax, zmin, dphi, *t)

    def wsprimary(x, y, z, l, m, n, ux, uy, uz, alpha, z0, psi):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_wsp(*p, num, alpha, z0, psi)

    def wssecondary(x, y, z, l, m, n, ux, uy, uz, alpha, z0, psi):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_wss(*p, num, alpha, z0, psi)

    def legendre(x, n):
        a, b = C.c_double(), C.c_double()
        f_leg(x, n, C.byref(a), C.byref(b))
        return a.value, b.value

    def wsprimaryback(x, y, z, l, m, n, ux, uy, uz, alpha, z0, psi, thick):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_wspb(*p, num, alpha, z0, psi, thick)

    def wssecondaryback(x, y, z, l, m, n, ux, uy, uz, alpha, z0, psi, thick):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_wssb(*p, num, alpha, z0, psi, thick)

    def torus(x, y, z, l, m, n, ux, uy, uz, rin, rout):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        f_tor(*p, num, rin, rout)

    def conicplus(x, y, z, l, m, n, ux, uy, uz, R, K, pp):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        pk, ppt = _f64in(pp)
        f_cp(*p, num, R, K, ppt, len(pk))

    def conicplusopd(opd, x, y, z, l, m, n, ux, uy, uz, R, K, pp, nr):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        pk, ppt = _f64in(pp)
        f_cpo(*p, num, R, K, ppt, len(pk), nr)

    def legsurf(x, y, z, l, m, n, ux, uy, uz, xwidth, ywidth, order, coeff, xo, yo):
        p, num = _cols(x, y, l, m, n)
        ck, cp = _f64in(coeff)
        xk, xp = _i32(xo)
        yk, yp = _i32(yo)
        f_lgs(*p, num, xwidth, ywidth, order, cp, xp, yp, len(ck))

    def zernphase(opd, x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad, wave):
        p, num = _cols(opd, x, y, l, m, n)
        ck, cp = _f64in(coeff)
        rk, rp = _i32(rorder)
        ak, ap = _i32(aorder)
        if f_zph(*p, num, cp, rp, ap, len(ck), rad, wave) != 0:
            raise ValueError("zernphase: (rorder, aorder) outside the reference's radial table")

    def tracezernrot(x, y, z, l, m, n, ux, uy, uz, coeff1, rorder1, aorder1, coeff2, rorder2, aorder2, rad, rot):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        c1, c1p = _f64in(coeff1)
        r1, r1p = _i32(rorder1)
        a1, a1p = _i32(aorder1)
        c2, c2p = _f64in(coeff2)
        r2, r2p = _i32(rorder2)
        a2, a2p = _i32(aorder2)
        if f_zrot(*p, num, c1p, r1p, a1p, len(c1), c2p, r2p, a2p, len(c2), rad, rot) != 0:
            raise ValueError("tracezernrot: (rorder, aorder) outside the reference's radial table")

    f_swb = sig("orc_southwellbin", *[P] * 4, i64, d, *[P] * 3, i32, i32)
    f_rec = sig("orc_reconstruct", P, P, i32, i32, d, d, P, P, i32, restype=i32)

    def southwellbin(x, y, l, m, binsize, xdim, ydim):
        p, num = _cols(x, y, l, m)
        out = [np.zeros((xdim, ydim), order="F") for _ in range(3)]
        f_swb(*p, num, binsize, *[_g64(a) for a in out], xdim, ydim)
        return tuple(out)

    def reconstruct(xang, yang, criteria, h, phase, maxiter):
        xdim, ydim = _grid2(xang, yang, phase)
        phasec = np.zeros((xdim, ydim), order="F")
        reconstruct.sweeps = f_rec(_g64(xang), _g64(yang), xdim, ydim, criteria, h, _g64(phase), _g64(phasec), maxiter)
        return phasec

    def wolter_params(r0, z0, psi=1.):
        o = [C.c_double() for _ in range(3)]
        f_par(r0, z0, psi, *[C.byref(v) for v in o])
        return tuple(v.value for v in o)

    return SimpleNamespace(
        kind="port", omp=omp,
        surfacesf=SimpleNamespace(flat=flat, flatopd=flatopd, conic=conic, conicopd=conicopd,
                                  tracesphere=tracesphere, tracesphereopd=tracesphereopd, tracecyl=tracecyl,
                                  tracecylopd=tracecylopd, cylconic=cylconic, paraxial=paraxial,
                                  paraxialy=paraxialy, torus=torus, conicplus=conicplus, conicplusopd=conicplusopd,
                                  legsurf=legsurf),
        woltsurf=SimpleNamespace(wolterprimary=wolterprimary, wolterprimaryopd=wolterprimaryopd,
                                 woltersecondary=woltersecondary, spocone=spocone, woltersine=woltersine,
                                 wolterprimll=wolterprimll, woltersecll=woltersecll,
                                 ellipsoidwoltll=ellipsoidwoltll, wsprimary=wsprimary, wssecondary=wssecondary,
                                 wsprimaryback=wsprimaryback, wssecondaryback=wssecondaryback, legendre=legendre),
        zernsurf=SimpleNamespace(tracezern=tracezern, tracezernopd=tracezernopd, zernset=zernset,
                                 zernphase=zernphase, tracezernrot=tracezernrot),
        transformationsf=SimpleNamespace(transform=transform, itransform=itransform, reflect=reflect,
                                         radgrat=radgrat, radgratw=radgratw, grat=grat, refract=refract),
        reconstruct=SimpleNamespace(southwellbin=southwellbin, reconstruct=reconstruct),
        wolter_params=wolter_params)


# --------------------------------------------------------------------------------------------
def ref_available(omp=False):
    sfx = "_omp" if omp else ""
    return all(os.path.exists(os.path.join(HERE, "_ref", "libpxfref_%s%s.so" % (m, sfx)))
               for m in ("surfacesf", "woltsurf", "zernsurf", "transformationsf", "reconstruct"))


def load_ref(omp=False):
    """The real reference (Fortran ABI).  Raises FileNotFoundError when oracle/_ref is absent."""
    if not ref_available(omp):
        raise FileNotFoundError("oracle/_ref not built (needs /root/reference + flang: make -C oracle ref)")
    sfx = "_omp" if omp else ""
    L = {m: C.CDLL(os.path.join(HERE, "_ref", "libpxfref_%s%s.so" % (m, sfx)))
         for m in ("surfacesf", "woltsurf", "zernsurf", "transformationsf", "reconstruct")}

    def fn(mod, name):
        f = getattr(L[mod], name + "_")
        f.restype = None
        return f

    def num32(num):
        if num > 2 ** 31 - 1:
            raise OverflowError("reference `num` is a default INTEGER (int32)")
        return C.byref(C.c_int32(num))

    def dref(v):
        return C.byref(C.c_double(float(v)))

    def mk9(mod, name, nscal):
        f = fn(mod, name)

        def call(x, y, z, l, m, n, ux, uy, uz, *scal):
            assert len(scal) == nscal
            p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
            f(*p, num32(num), *[dref(s) for s in scal])
        return call

    def mk10(mod, name, nscal):
        f = fn(mod, name)

        def call(opd, x, y, z, l, m, n, ux, uy, uz, *scal):
            assert len(scal) == nscal
            p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
            f(*p, num32(num), *[dref(s) for s in scal])
        return call

    f_flatopd = fn("surfacesf", "flatopd")

    def flatopd(x, y, z, l, m, n, ux, uy, uz, opd, nr):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz, opd)
        f_flatopd(*p, num32(num), dref(nr))

    f_tz = fn("zernsurf", "tracezern")
    f_tzo = fn("zernsurf", "tracezernopd")
    f_zs = fn("zernsurf", "zernset")

    def tracezern(x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        ck, cp = _f64in(coeff)
        rk, rp = _i32(rorder)
        ak, ap = _i32(aorder)
        f_tz(*p, num32(num), cp, rp, ap, C.byref(C.c_int32(len(ck))), dref(rad))

    def tracezernopd(opd, x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad, nr):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        ck, cp = _f64in(coeff)
        rk, rp = _i32(rorder)
        ak, ap = _i32(aorder)
        f_tzo(*p, num32(num), cp, rp, ap, C.byref(C.c_int32(len(ck))), dref(rad), dref(nr))

    def zernset(rho, theta, rorder, aorder):
        rk, rp = _i32(rorder)
        ak, ap = _i32(aorder)
        out = [np.zeros(len(rk)) for _ in range(3)]
        f_zs(dref(rho), dref(theta), rp, ap, C.byref(C.c_int32(len(rk))), *[_f64(o) for o in out])
        return out

    f_refl = fn("transformationsf", "reflect")

    def reflect(l, m, n, ux, uy, uz):
        p, num = _cols(l, m, n, ux, uy, uz)
        f_refl(*p, num32(num))

    f_rg = fn("transformationsf", "radgrat")
    f_rgw = fn("transformationsf", "radgratw")

    def radgrat(x, y, l, m, n, wave, dpermm, order):
        p, num = _cols(x, y, l, m, n)
        f_rg(*p, dref(wave), num32(num), dref(dpermm), dref(order))

    def radgratw(x, y, l, m, n, wave, dpermm, order):
        p, num = _cols(x, y, l, m, n, wave)
        f_rgw(*p, num32(num), dref(dpermm), dref(order))

    def mkll(name, nscal):
        f = fn("woltsurf", name)

        def call(x, y, z, l, m, n, ux, uy, uz, *args):
            scal, (coeff, axial, az) = args[:nscal], args[nscal:]
            p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
            ck, cp = _f64in(coeff)
            ak, ap = _i32(axial)
            zk, zp = _i32(az)
            f(*p, num32(num), *[dref(v) for v in scal], cp, ap, zp, C.byref(C.c_int32(len(ck))))
        return call

    f_leg = getattr(L["woltsurf"], "legendre_")
    f_leg.restype = C.c_double
    f_legp = getattr(L["woltsurf"], "legendrep_")
    f_legp.restype = C.c_double

    def legendre(x, n):
        return (f_leg(dref(x), C.byref(C.c_int32(n))), f_legp(dref(x), C.byref(C.c_int32(n))))

    f_cp = fn("surfacesf", "conicplus")
    f_cpo = fn("surfacesf", "conicplusopd")
    f_lgs = fn("surfacesf", "legsurf")
    f_zph = fn("zernsurf", "zernphase")
    f_zrot = fn("zernsurf", "tracezernrot")

    def conicplus(x, y, z, l, m, n, ux, uy, uz, R, K, pp):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        pk, ppt = _f64in(pp)
        f_cp(*p, num32(num), dref(R), dref(K), ppt, C.byref(C.c_int32(len(pk))))

    def conicplusopd(opd, x, y, z, l, m, n, ux, uy, uz, R, K, pp, nr):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        pk, ppt = _f64in(pp)
        f_cpo(*p, num32(num), dref(R), dref(K), ppt, C.byref(C.c_int32(len(pk))), dref(nr))

    def legsurf(x, y, z, l, m, n, ux, uy, uz, xwidth, ywidth, order, coeff, xo, yo):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        ck, cp = _f64in(coeff)
        xk, xp = _i32(xo)
        yk, yp = _i32(yo)    # legSurf(x,...,uz,xwidth,ywidth,order,coeff,xo,yo,Nc,num): num is LAST
        f_lgs(*p, dref(xwidth), dref(ywidth), dref(order), cp, xp, yp, C.byref(C.c_int32(len(ck))), num32(num))

    def zernphase(opd, x, y, z, l, m, n, ux, uy, uz, coeff, rorder, aorder, rad, wave):
        p, num = _cols(opd, x, y, z, l, m, n, ux, uy, uz)
        ck, cp = _f64in(coeff)
        rk, rp = _i32(rorder)
        ak, ap = _i32(aorder)
        f_zph(*p, num32(num), cp, rp, ap, C.byref(C.c_int32(len(ck))), dref(rad), dref(wave))

    def tracezernrot(x, y, z, l, m, n, ux, uy, uz, coeff1, rorder1, aorder1, coeff2, rorder2, aorder2, rad, rot):
        p, num = _cols(x, y, z, l, m, n, ux, uy, uz)
        c1, c1p = _f64in(coeff1)
        r1, r1p = _i32(rorder1)
        a1, a1p = _i32(aorder1)
        c2, c2p = _f64in(coeff2)
        r2, r2p = _i32(rorder2)
        a2, a2p = _i32(aorder2)
        f_zrot(*p, num32(num), c1p, r1p, a1p, C.byref(C.c_int32(len(c1))), c2p, r2p, a2p,
               C.byref(C.c_int32(len(c2))), dref(rad), dref(rot))

    f_gr = fn("transformationsf", "grat")

    def grat(x, y, l, m, n, d_, order, wave):
        p, num = _cols(x, y, l, m, n)
        ok, op_ = _f64in(np.array(np.broadcast_to(np.asarray(order, dtype=np.float64), (num,))))
        wk, wp_ = _f64in(np.array(np.broadcast_to(np.asarray(wave, dtype=np.float64), (num,))))
        f_gr(*p, num32(num), dref(d_), op_, wp_)        # grat(x,y,l,m,n,num,d,order,wave)

    f_rfr = fn("transformationsf", "refract")

    def refract(l, m, n, ux, uy, uz, n1, n2):
        p, num = _cols(l, m, n, ux, uy, uz)
        f_rfr(*p, num32(num), dref(n1), dref(n2))

    f_swb, f_rec = fn("reconstruct", "southwellbin"), fn("reconstruct", "reconstruct")

    def southwellbin(x, y, l, m, binsize, xdim, ydim):
        # subroutine southwellbin(x,y,l,m,num,binsize,xang,yang,phase,xdim,ydim)   reconstruct.f95:139
        # (out-of-grid rays index out of bounds in the reference: callers keep rays inside)
        p, num = _cols(x, y, l, m)
        out = [np.zeros((xdim, ydim), order="F") for _ in range(3)]
        f_swb(*p, num32(num), dref(binsize), *[_g64(a) for a in out], C.byref(C.c_int32(xdim)),
              C.byref(C.c_int32(ydim)))
        return tuple(out)

    def reconstruct(xang, yang, criteria, h, phase, maxiter):
        # subroutine reconstruct(xang,yang,xdim,ydim,criteria,h,phase,phasec,maxiter)  reconstruct.f95:1
        xdim, ydim = _grid2(xang, yang, phase)
        phasec = np.zeros((xdim, ydim), order="F")
        f_rec(_g64(xang), _g64(yang), C.byref(C.c_int32(xdim)), C.byref(C.c_int32(ydim)), dref(criteria), dref(h),
              _g64(phase), _g64(phasec), C.byref(C.c_int32(maxiter)))
        return phasec

    return SimpleNamespace(
        kind="reference", omp=omp,
        reconstruct=SimpleNamespace(southwellbin=southwellbin, reconstruct=reconstruct),
        surfacesf=SimpleNamespace(flat=mk9("surfacesf", "flat", 0), flatopd=flatopd,
                                  conic=mk9("surfacesf", "conic", 2),
                                  conicopd=mk10("surfacesf", "conicopd", 3),
                                  tracesphere=mk9("surfacesf", "tracesphere", 1),
                                  tracesphereopd=mk10("surfacesf", "tracesphereopd", 2),
                                  tracecyl=mk9("surfacesf", "tracecyl", 1),
                                  tracecylopd=mk10("surfacesf", "tracecylopd", 2),
                                  cylconic=mk9("surfacesf", "cylconic", 2),
                                  paraxial=mk9("surfacesf", "paraxial", 1),
                                  paraxialy=mk9("surfacesf", "paraxialy", 1),
                                  torus=mk9("surfacesf", "torus", 2), conicplus=conicplus,
                                  conicplusopd=conicplusopd, legsurf=legsurf),
        woltsurf=SimpleNamespace(wolterprimary=mk9("woltsurf", "wolterprimary", 3),
                                 wolterprimaryopd=mk10("woltsurf", "wolterprimaryopd", 4),
                                 woltersecondary=mk9("woltsurf", "woltersecondary", 3),
                                 spocone=mk9("woltsurf", "spocone", 2),
                                 woltersine=mk9("woltsurf", "woltersine", 4),
                                 wolterprimll=mkll("wolterprimll", 5), woltersecll=mkll("woltersecll", 6),
                                 ellipsoidwoltll=mkll("ellipsoidwoltll", 7),
                                 wsprimary=mk9("woltsurf", "wsprimary", 3),
                                 wssecondary=mk9("woltsurf", "wssecondary", 3),
                                 wsprimaryback=mk9("woltsurf", "wsprimaryback", 4),
                                 wssecondaryback=mk9("woltsurf", "wssecondaryback", 4), legendre=legendre),
        zernsurf=SimpleNamespace(tracezern=tracezern, tracezernopd=tracezernopd, zernset=zernset,
                                 zernphase=zernphase, tracezernrot=tracezernrot),
        transformationsf=SimpleNamespace(transform=mk9("transformationsf", "transform", 6),
                                         itransform=mk9("transformationsf", "itransform", 6),
                                         reflect=reflect, radgrat=radgrat, radgratw=radgratw, grat=grat,
                                         refract=refract))
