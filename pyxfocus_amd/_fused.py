"""Fused tracing: run a sequence of PyXFocus calls as ONE kernel launch.

    with pyxfocus_amd.fused():
        tran.transform(rays, 0, 0, -(z0 + 2*L), 0, 0, 0)
        surf.wolterprimary(rays, r0, z0);  tran.reflect(rays)
        surf.woltersecondary(rays, r0, z0); tran.reflect(rays)
        surf.flat(rays)
    # <- here the rays are up to date

Inside the block the per-ray calls (flat, conic, wolterprimary[opd], woltersecondary, transform,
itransform, reflect, radgrat -- with or without ``ind``; and, in programs of their own kind, a Zernike trace or
the W-S / Legendre-Legendre / woltersine mirrors between frame changes, reflections and flats) are RECORDED
instead of launched; at the
end of the block (or at ``flush()``, or as soon as a call arrives that cannot be fused: vignette,
applyT, pointTo, zernsurf, any analyses.* reduction, a call on a different bundle) the recorded
program runs through ``pxf_program_run``: each column is read once and written once, with the ray in
registers in between -- 120 B/ray instead of 576 for the two-bounce chain -- and operations whose
result is overwritten unread are dropped.  The arithmetic is the same device code the one-by-one
calls use, so the result is bit-identical to running without ``fused()``.

Reading a ray tensor directly (``rays[1].cpu()``) inside the block does NOT trigger a flush --
call ``flush()`` first.

``fused(fast=True)`` is an OPT-IN tolerance mode (``pxf_program_run_fast``): programs made only of
flat / transform / itransform / reflect / wolterprimary / woltersecondary / radgrat / spoCone / indAngle(out=)
calls without ``ind=`` run through a kernel that reaches the HBM roof by giving up bit identity (same
Newton steps on the exact quadratic along the ray, reciprocal-based divides, FMA).  What it does NOT do is meet
the 1e-10 mm RMS ray-position bar as a plain RMS over all rays: the reference rounds the propagation distance of
every ``flat`` to fp32 (surfacesf.f95:4-15), and for about 2.5 rays per million a last-bit difference upstream
makes that rounding fall the other way -- such a ray lands one fp32 quantum of its path aside (2.6e-5 mm on the
8.4 m leg of the BASELINE configs), which alone puts the all-ray RMS at ~1e-7 mm (3 such rays in 2e5 on
config 3).  With those fp32-flat flips set aside the agreement is 5.5e-11 (config 2) / 7.5e-11 mm (config 3)
RMS -- the size of the reference's own rounding jitter -- for focal lengths z0 <= 12 m; the jitter grows with
z0 squared, 1-6e-10 mm at z0 = 15-20 m, and a fast block that records a Wolter-I
surface with z0 > 12 m warns (``FastModeWarning``).  Programs holding anything else run through the exact kernel
as usual.  The default stays bit-identical, and the straight-line form of the bit-exact two-bounce program
runs within ~5 % of this mode's speed: the tolerance mode buys something only for longer programs
(config 3: 20 vs 28 ms per 1e9 rays).
"""
import ctypes as C
import threading
import warnings

import torch

from . import _lib

_tls = threading.local()

OPCODES = {"pxf_flat": 1, "pxf_flatopd": 2, "pxf_conic": 3, "pxf_conicopd": 4, "pxf_wolterprimary": 5,
           "pxf_wolterprimaryopd": 6, "pxf_woltersecondary": 7, "pxf_transform": 8, "pxf_itransform": 9,
           "pxf_reflect": 10, "pxf_radgrat": 11, "pxf_radgratw": 12, "pxf_spocone": 13, "pxf_tracesphere": 14,
           "pxf_tracesphereopd": 15, "pxf_tracecyl": 16, "pxf_tracecylopd": 17, "pxf_cylconic": 18,
           "pxf_paraxial": 19, "pxf_paraxialy": 20, "pxf_grat": 21, "pxf_refract": 22, "pxf_indangle": 25,
           "pxf_wsprimary": 26, "pxf_wssecondary": 27, "pxf_wsprimaryback": 28, "pxf_wssecondaryback": 29,
           "pxf_woltersine": 30}
# bundle slots (opd=0, x=1 ... uz=9) of the columns each entry point takes, in call order
SLOTS = {"pxf_flat": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_flatopd": (1, 2, 3, 4, 5, 6, 7, 8, 9, 0),
         "pxf_conic": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_conicopd": (0, 1, 2, 3, 4, 5, 6, 7, 8, 9),
         "pxf_wolterprimary": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_wolterprimaryopd": (0, 1, 2, 3, 4, 5, 6, 7, 8, 9),
         "pxf_woltersecondary": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_transform": (1, 2, 3, 4, 5, 6, 7, 8, 9),
         "pxf_itransform": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_reflect": (4, 5, 6, 7, 8, 9),
         "pxf_radgrat": (1, 2, 4, 5, 6), "pxf_radgratw": (1, 2, 4, 5, 6, -1),
         "pxf_spocone": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_tracesphere": (1, 2, 3, 4, 5, 6, 7, 8, 9),
         "pxf_tracesphereopd": (0, 1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_tracecyl": (1, 2, 3, 4, 5, 6, 7, 8, 9),
         "pxf_tracecylopd": (0, 1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_cylconic": (1, 2, 3, 4, 5, 6, 7, 8, 9),
         "pxf_paraxial": (1, 2, 4, 5), "pxf_paraxialy": (2, 5), "pxf_grat": (4, 5, 6, -1),
         "pxf_refract": (4, 5, 6, 7, 8, 9), "pxf_indangle": (4, 5, 6, 7, 8, 9, -1),
         "pxf_wsprimary": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_wssecondary": (1, 2, 3, 4, 5, 6, 7, 8, 9),
         "pxf_wsprimaryback": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_wssecondaryback": (1, 2, 3, 4, 5, 6, 7, 8, 9),
         "pxf_woltersine": (1, 2, 3, 4, 5, 6, 7, 8, 9)}
MAX_OPS = 32
MAX_OPS_ZERN = 16      # PXF_PROGRAM_MAX_OPS_ZERN: a program around a Zernike trace
ZERN_MAX_ORDER = 15    # PXF_PROGRAM_ZERN_MAX_ORDER
# what a program around a Zernike trace may hold besides the trace (pxf_program_zern.hip: zprog_supported)
ZERN_COMPANIONS = {1, 2, 8, 9, 10}     # flat, flatopd, transform, itransform, reflect
ZERN_CODES = {"pxf_tracezern": 23, "pxf_tracezernopd": 24}
# tolerance-parity mirrors inside a program (pxf_program_mirror.hip): W-S, woltersine (scalar arguments only) and
# the Legendre-Legendre kinds (host coefficient lists).  Such a program holds at most 12 unmasked operations out of
# the mirrors and flat, flatopd, transform, itransform, reflect; all L-L matrices share one 288-cell pool.
MIRROR_SCALAR_CODES = {26, 27, 28, 29, 30}
LL_CODES = {"pxf_wolterprimll": 31, "pxf_woltersecll": 32, "pxf_ellipsoidwoltll": 33}
MIRROR_CODES = MIRROR_SCALAR_CODES | set(LL_CODES.values())
MIRROR_COMPANIONS = {1, 2, 8, 9, 10}
MAX_OPS_MIRROR = 12    # PXF_PROGRAM_MAX_OPS_MIRROR
LL_POOL_CELLS, LL_POOL_ROWS, LL_MAX_AZ = 288, 16, 11
ZERN_SLOTS = {"pxf_tracezern": (1, 2, 3, 4, 5, 6, 7, 8, 9), "pxf_tracezernopd": (0, 1, 2, 3, 4, 5, 6, 7, 8, 9)}


class PxfZernArgs(C.Structure):
    _fields_ = [("coeff", C.POINTER(C.c_double)), ("rorder", C.POINTER(C.c_int32)), ("aorder", C.POINTER(C.c_int32)),
                ("arrsize", C.c_int32)]


class PxfLLArgs(C.Structure):
    _fields_ = [("coeff", C.POINTER(C.c_double)), ("axial", C.POINTER(C.c_int32)), ("az", C.POINTER(C.c_int32)),
                ("cnum", C.c_int32), ("S", C.c_double)]


class PxfOp(C.Structure):
    _fields_ = [("code", C.c_int32), ("reserved", C.c_int32), ("mask", C.c_void_p), ("aux", C.c_void_p),
                ("a", C.c_double * 6)]


FAST_CODES = {1, 5, 7, 8, 9, 10, 11, 13, 25}     # what pxf_program_run_fast accepts (pxf_program_fast.hip)
FAST_Z0_MAX = 12000.                              # mm: beyond it the tolerance mode's own jitter passes 1e-10 mm RMS


class FastModeWarning(UserWarning):
    """fused(fast=True) used where its documented agreement with the reference does not hold."""
AUX_OUT_CODES = {25}                              # operations whose aux is an OUTPUT column (indAngle)


class _Recorder:
    def __init__(self, fast=False):
        self.reset()
        self.launches = 0
        self.fused_ops = 0
        self.fast = bool(fast)
        self.fast_launches = 0

    def reset(self):
        self.ops = []
        self.cols = [None] * 10      # tensors by bundle slot (kept alive until the launch)
        self.ptrs = [0] * 10         # their device addresses (0 = slot not used yet)
        self.keep = []               # masks / aux columns / host tables kept alive until the launch
        self.has_zern = False        # the program holds its (one) Zernike operation
        self.has_mirror = False      # the program holds W-S / Legendre-Legendre / woltersine mirrors
        self.ll_shapes = []          # (highest axial order, highest azimuthal order) of its L-L operations
        self.num = None
        self.dev = None

    def _bind(self, cols, slots):
        """Claim the bundle slots for `cols`; returns (ok, aux) -- not ok when a different tensor
        already sits in a slot, or the bundle length / device differs from the pending program's."""
        num = cols[0].shape[0]
        if self.num is not None and (num != self.num or cols[0].device != self.dev):
            return False, None
        aux = None
        ptrs = self.ptrs
        new = []
        for t, s in zip(cols, slots):
            if s < 0:
                aux = t
                continue
            p = t.data_ptr()
            if ptrs[s] != p:
                if ptrs[s]:
                    return False, None   # a different tensor in a slot the program already uses
                new.append((s, t, p))
        for s, t, p in new:
            self.cols[s] = t
            ptrs[s] = p
        self.num, self.dev = num, cols[0].device
        return True, aux

    def try_add(self, name, cols, tail, mask):
        """Record one call; False if it cannot join the pending program (caller flushes and retries)."""
        if name not in OPCODES or len(self.ops) >= (MAX_OPS_ZERN if self.has_zern else MAX_OPS):
            return False
        if self.has_zern and (mask is not None or OPCODES[name] not in ZERN_COMPANIONS):
            return False                              # no ind= masks / heavy operations around a Zernike trace
        if OPCODES[name] in MIRROR_SCALAR_CODES and not self._mirror_may_join(mask):
            return False
        if self.has_mirror and (mask is not None or len(self.ops) >= MAX_OPS_MIRROR or
                                OPCODES[name] not in (MIRROR_COMPANIONS | MIRROR_SCALAR_CODES)):
            return False                              # ... nor around a W-S / L-L mirror
        ok, aux = self._bind(cols, SLOTS[name])
        if not ok:
            return False
        op = PxfOp()
        op.code = OPCODES[name]
        op.mask = mask.data_ptr() if mask is not None else None
        op.aux = aux.data_ptr() if aux is not None else None
        for i, v in enumerate(t for t in tail if t is not None):
            op.a[i] = float(v)
        self.ops.append(op)
        self.keep.extend(t for t in (mask, aux) if t is not None)
        self.has_mirror |= op.code in MIRROR_SCALAR_CODES
        return True

    def _mirror_may_join(self, mask):
        """A mirror operation can join the pending program: unmasked, no Zernike trace in it, every operation
        recorded so far is one a mirror program may hold, and there is room."""
        return (mask is None and not self.has_zern and len(self.ops) < MAX_OPS_MIRROR and
                not any(o.mask or o.code not in (MIRROR_COMPANIONS | MIRROR_CODES) for o in self.ops))

    def _ll_pool_fits(self, shapes):
        if any(kt > LL_MAX_AZ or ((ka + 2) & ~1) > LL_POOL_ROWS for ka, kt in shapes):
            return False
        kt1 = max((kt + 4) & ~3 for _, kt in shapes)
        return sum(((ka + 2) & ~1) * kt1 for ka, _ in shapes) <= LL_POOL_CELLS

    def try_add_ll(self, name, cols, scalars, tables, S, mask):
        """Record a Legendre-Legendre mirror.  `tables` = ((coeff, axial, az) host arrays, their ctypes pointers +
        length); all L-L matrices of a program share one zero-padded pool (module constants above)."""
        (ck, ak, zk), (cp, ap, zp, n) = tables
        if not self._mirror_may_join(mask) or n == 0:
            return False
        shape = (int(ak.max()), int(zk.max()))
        if not self._ll_pool_fits(self.ll_shapes + [shape]):
            return False
        ok, _ = self._bind(cols, (1, 2, 3, 4, 5, 6, 7, 8, 9))
        if not ok:
            return False
        la = PxfLLArgs(cp, ap, zp, n, float(S))
        op = PxfOp()
        op.code = LL_CODES[name]
        op.mask = None
        op.aux = C.addressof(la)
        for i, v in enumerate(scalars):
            op.a[i] = float(v)
        self.ops.append(op)
        self.keep.extend((la, ck, ak, zk))
        self.ll_shapes.append(shape)
        self.has_mirror = True
        return True

    def try_add_zern(self, name, cols, tables, rad, nr, mask):
        """Record a tracezern[opd] call.  `tables` = (coeff, rorder, aorder) host arrays and their ctypes
        pointers; one Zernike operation per program, and such a program holds at most 16 operations."""
        if self.has_zern or self.has_mirror or len(self.ops) >= MAX_OPS_ZERN or mask is not None:
            return False
        if any(o.mask or o.code not in ZERN_COMPANIONS for o in self.ops):
            return False
        if len(tables[1]) and int(tables[1].max()) > ZERN_MAX_ORDER:
            return False
        ok, _ = self._bind(cols, ZERN_SLOTS[name])
        if not ok:
            return False
        ck, rk, ak, cp, rp, ap = tables
        za = PxfZernArgs(cp, rp, ap, len(ck))
        op = PxfOp()
        op.code = ZERN_CODES[name]
        op.mask = mask.data_ptr() if mask is not None else None
        op.aux = C.addressof(za)
        op.a[0] = float(rad)
        op.a[1] = float(nr)
        self.ops.append(op)
        self.keep.extend(t for t in (mask, za, ck, rk, ak) if t is not None)
        self.has_zern = True
        return True

    def flush(self):
        if not self.ops:
            self.reset()
            return
        ops = (PxfOp * len(self.ops))(*self.ops)
        ptrs = (C.c_void_p * 10)(*[(p or None) for p in self.ptrs])
        dev = self.dev
        from .native import _stream          # (native imports this module)
        fast = self.fast and all(o.code in FAST_CODES and not o.mask and (not o.aux or o.code in AUX_OUT_CODES)
                                 for o in self.ops)
        run = _lib.lib().pxf_program_run_fast if fast else _lib.lib().pxf_program_run
        if fast:
            _warn_fast_z0(self.ops)
        if dev.index == torch.cuda.current_device():    # the common case: skip the device guard
            rc = run(ops, len(self.ops), ptrs, self.num, _stream(dev))
        else:
            with torch.cuda.device(dev):
                rc = run(ops, len(self.ops), ptrs, self.num, _stream(dev))
        n = len(self.ops)
        self.reset()
        _lib.check(rc, "pxf_program_run_fast" if fast else "pxf_program_run")
        self.launches += 1
        self.fast_launches += int(fast)
        self.fused_ops += n


def _warn_fast_z0(ops):
    """The tolerance mode's jitter is the rounding of terms of size z0^2 (tests/soak/fast_soak.py): past
    z0 = 12 m it exceeds 1e-10 mm RMS even with the fp32-flat flips set aside.  Say so once per offending call site."""
    for o in ops:
        if o.code in (5, 7) and abs(o.a[1]) > FAST_Z0_MAX:          # wolterprimary / woltersecondary: a = (r0, z0, psi)
            warnings.warn("fused(fast=True): Wolter-I surface with z0 = %g mm > %g mm -- the tolerance mode's "
                          "agreement with the reference degrades as z0^2 (1-6e-10 mm RMS at 15-20 m, fp32-flat flips "
                          "aside); the default fused() is bit-identical at any z0" % (o.a[1], FAST_Z0_MAX),
                          FastModeWarning, stacklevel=4)
            return


def active():
    return getattr(_tls, "rec", None)


def flush():
    """Run whatever is recorded (no-op outside a fused block)."""
    rec = active()
    if rec is not None:
        rec.flush()


class fused:
    """Context manager; re-entrant use nests into the outermost block.  fast=True: opt into the
    tolerance-mode kernel for the programs it supports (module docstring)."""

    def __init__(self, fast=False):
        self.fast = fast

    def __enter__(self):
        self.outer = active()
        if self.outer is None:
            _tls.rec = _Recorder(fast=self.fast)
        elif bool(self.fast) != self.outer.fast:
            # nesting joins the OUTER block, whose mode was fixed when it was entered: say so instead of silently
            # dropping (or gaining) the tolerance mode
            warnings.warn("fused(fast=%s) nested inside fused(fast=%s): the inner block joins the outer one and runs in "
                          "the OUTER block's mode" % (bool(self.fast), self.outer.fast), FastModeWarning, stacklevel=2)
        return active()

    def __exit__(self, exc_type, exc, tb):
        if self.outer is None:
            rec = active()
            _tls.rec = None
            if exc_type is None:
                rec.flush()
        return False


class SegmentedProgram:
    """A trace compiled for a bundle of equal contiguous SEGMENTS (mirror shells, modules ...): the
    same call sequence with per-segment parameters, executed by ONE launch (pxf_program_run_segments;
    parameter blocks staged per workgroup in LDS).  Build with `compile_segments`, then `.run()` as
    often as needed (e.g. after refilling the bundle from a segmented source)."""

    def __init__(self, rays, rays_per_seg, ops, table, keep, fast=False):
        self.fast = bool(fast) and all(o.code in FAST_CODES and not o.mask and (not o.aux or o.code in AUX_OUT_CODES)
                                       for o in ops)
        self.rays = list(rays)
        self.rps = int(rays_per_seg)
        self.ops = ops                  # ctypes array of PxfOp (codes, masks, aux)
        self.table = table              # device tensor [nseg, nlive, 9]
        self.keep = keep

    def run(self):
        flush()
        dev = self.rays[1].device
        ptrs = (C.c_void_p * 10)(*[t.data_ptr() for t in self.rays])
        run = _lib.lib().pxf_program_run_segments_fast if self.fast else _lib.lib().pxf_program_run_segments
        with torch.cuda.device(dev):
            rc = run(self.ops, len(self.ops), self.table.data_ptr(), self.table.shape[0], self.rps, ptrs,
                     self.rays[1].shape[0], torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "pxf_program_run_segments")


def compile_segments(rays, rays_per_seg, fn, params, fast=False):
    """Record `fn(segment_view, *params[s])` for every segment s WITHOUT launching anything, check that
    all segments issue the same operation sequence, and return a SegmentedProgram for the whole
    bundle.  `rays` is the full ten-column bundle, segment s = rays[s*rays_per_seg:(s+1)*rays_per_seg].
    `fn` may use any fusable call (see `fused`), including `ind=` masks only if they are whole-bundle
    masks (not supported per segment).  fast=True: the tolerance-mode kernel, where the sequence allows."""
    import numpy as np
    num = rays[1].shape[0]
    nseg = len(params)
    rps = int(rays_per_seg)
    if nseg * rps != num:
        raise ValueError("compile_segments: len(params) * rays_per_seg must equal the bundle length")
    h = _lib.lib()
    rows, codes0, ops0, aux0, keep_all = [], None, None, None, []
    if getattr(_tls, "rec", None) is not None:
        raise _lib.PxfError("compile_segments cannot be called inside a fused() block")
    for s_, par in enumerate(params):
        view = [c[s_ * rps:(s_ + 1) * rps] for c in rays]
        rec = _Recorder()
        _tls.rec = rec
        try:
            fn(view, *par)
        finally:
            _tls.rec = None
        if rec.launches or not rec.ops:
            raise _lib.PxfError("compile_segments: the segment function issued a call that cannot be fused")
        if rec.has_zern or rec.has_mirror:
            raise _lib.PxfError("compile_segments: a Zernike trace / W-S / Legendre-Legendre / woltersine mirror carries "
                                "its own tables and cannot be a per-segment operation")
        if any(o.mask or (o.aux and o.code not in AUX_OUT_CODES) for o in rec.ops):
            raise _lib.PxfError("compile_segments: ind= masks / per-ray input columns are not supported per segment")
        # output columns (indAngle(..., out=)): segment s must write the s-th slice of ONE bundle-long column
        auxs = [o.aux for o in rec.ops]
        if s_ == 0:
            aux0 = auxs
        elif any((a or 0) != ((b or 0) + 8 * s_ * rps if b else 0) for a, b in zip(auxs, aux0)):
            raise _lib.PxfError("compile_segments: out= of segment %d is not the slice [s*rays_per_seg, ...) of the "
                                "column segment 0 writes" % s_)
        keep_all.extend(rec.keep)
        arr = (PxfOp * len(rec.ops))(*rec.ops)
        out = (C.c_double * (9 * len(rec.ops)))()
        nl = C.c_int32()
        _lib.check(h.pxf_program_params(arr, len(rec.ops), out, C.byref(nl)), "pxf_program_params")
        codes = [o.code for o in rec.ops]
        if codes0 is None:
            codes0, ops0, nlive = codes, arr, nl.value
        elif codes != codes0 or nl.value != nlive:
            raise _lib.PxfError("compile_segments: segment %d issues a different operation sequence" % s_)
        rows.append(np.frombuffer(out, dtype=np.float64, count=9 * nlive).copy())
    table = torch.from_numpy(np.stack(rows).reshape(nseg, nlive, 9)).to(rays[1].device)
    return SegmentedProgram(rays, rps, ops0, table, keep_all, fast=fast)


def plan(ops):
    """(load_mask, store_mask, dropped) the planner derives for a list of PxfOp (tests/reports)."""
    arr = (PxfOp * len(ops))(*ops)
    lm, sm, dr = C.c_uint32(), C.c_uint32(), C.c_int32()
    _lib.check(_lib.lib().pxf_program_plan(arr, len(ops), C.byref(lm), C.byref(sm), C.byref(dr)),
               "pxf_program_plan")
    return lm.value, sm.value, dr.value
