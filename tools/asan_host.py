"""Driver for tools/asan_host.sh: random programs through the planner, random order tables through
the Zernike / Legendre-Legendre host folding (fake device pointers are never dereferenced on the host)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyxfocus_amd import _lib  # noqa: E402
from pyxfocus_amd._fused import PxfOp  # noqa: E402

h = _lib.lib()
rng = np.random.default_rng(0)
DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int32)
for trial in range(3000):
    n = int(rng.integers(0, 40))
    ops = (PxfOp * max(n, 1))()
    for i in range(n):
        ops[i].code = int(rng.integers(-2, 26))
        for k in range(len(ops[i].a)):
            ops[i].a[k] = float(rng.normal())
    lm, sm, dr = C.c_uint32(), C.c_uint32(), C.c_int32()
    h.pxf_program_plan(ops, n, C.byref(lm), C.byref(sm), C.byref(dr))
    out = (C.c_double * (9 * max(n, 1)))()
    nl = C.c_int32()
    h.pxf_program_params(ops, n, out, C.byref(nl))
fake = [C.c_void_p(0x1000 * (i + 1)) for i in range(10)]
codes = {}
for trial in range(300):
    na = int(rng.integers(1, 120))
    nmax = int(rng.integers(0, 18))
    ro = rng.integers(0, nmax + 1, na).astype(np.int32)
    ao = np.array([rng.choice(np.arange(-r, r + 1, 2)) for r in ro], dtype=np.int32)
    co = rng.normal(size=na)
    rc = h.pxf_tracezern(*fake[:9], 128, co.ctypes.data_as(DP), ro.ctypes.data_as(IP), ao.ctypes.data_as(IP), na, 50.,
                         None, None)
    codes[rc] = codes.get(rc, 0) + 1
    cn = int(rng.integers(1, 40))
    co = rng.normal(size=cn)
    ax, az = rng.integers(0, 18, cn).astype(np.int32), rng.integers(0, 18, cn).astype(np.int32)
    rc = h.pxf_wolterprimll(*fake[:9], 128, 220., 8400., 8500., 8400., 6.28, co.ctypes.data_as(DP), ax.ctypes.data_as(IP),
                            az.ctypes.data_as(IP), cn, None, None)
    codes[rc] = codes.get(rc, 0) + 1
print("asan host pass done; status codes seen:", codes)
