#!/usr/bin/env python3
"""Dev tool: interleaved A/B of library build variants in ONE process (cdna guide rule 24).

usage: ab_kernels.py name=path[:BPC] ...     e.g.  base=pyxfocus_amd/csrc/libpxf_hip.so:0 nt=build/libpxf_hip_nt.so:0
Each variant is a copy of a libpxf_hip build loaded under its own name (own statics), with its own
PXF_BPC (a grid cap: honoured by builds up to round 2 only -- the ray kernels have had no grid-stride loop since).  Per round every variant traces a fresh config-2 bundle; each launch is bracketed by HIP
events.  Prints median/min ms and GB/s per kernel per variant.
"""
import ctypes as C
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyxfocus_amd import _lib  # noqa: E402

R0, Z0, PSI, L = 220., 8400., 1., 100.
N = int(float(os.environ.get("AB_RAYS", "1e8")))
ROUNDS = int(os.environ.get("AB_ROUNDS", "6"))
BYTES = [("transform", 144), ("wolterprimary", 96), ("reflect", 72), ("woltersecondary", 96), ("reflect2", 72),
         ("flat", 96)]


def load(path, bpc, tmp, name):
    dst = os.path.join(tmp, "libpxf_%s.so" % name)
    shutil.copy(path, dst)
    os.environ["PXF_BPC"] = str(bpc)
    h = C.CDLL(dst)
    for fn, argtypes in _lib._SIGS.items():
        try:
            f = getattr(h, fn)
        except AttributeError:          # an older build variant without a newer entry point
            continue
        f.argtypes = argtypes
        f.restype = _lib._RESTYPE.get(fn, C.c_int)
    return h


def main():
    tmp = tempfile.mkdtemp()
    variants = []
    for spec in sys.argv[1:]:
        name, rest = spec.split("=")
        path, _, bpc = rest.partition(":")
        variants.append((name, path, int(bpc or 0)))
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    libs = {}
    bufs = torch.empty((10, N), dtype=torch.float64, device=dev)
    cols = [bufs[i] for i in range(10)]
    ptr = [c.data_ptr() for c in cols]
    arr = (C.c_void_p * 10)(*ptr)
    from pyxfocus_amd import conicsolve as con
    rin, rout = float(con.primrad(Z0, R0, Z0)), float(con.primrad(Z0 + L, R0, Z0))
    for name, path, bpc in variants:
        h = load(path, bpc, tmp, name)
        # first call fixes the variant's static grid cap while its PXF_BPC is in the environment
        assert h.pxf_source(2, arr, 1024, rin, rout, 0., -1., 1, 0, st) == 0
        libs[name] = h
    torch.cuda.synchronize()
    times = {name: {k: [] for k, _ in BYTES} for name in libs}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    x, y, z, l, m, n, ux, uy, uz = ptr[1:]
    for rnd in range(ROUNDS + 1):
        order = list(libs) if rnd % 2 == 0 else list(libs)[::-1]
        for name in order:
            h = libs[name]
            assert h.pxf_source(2, arr, N, rin, rout, 0., -1., 100 + rnd, 0, st) == 0
            ev[0].record()
            h.pxf_transform(x, y, z, l, m, n, ux, uy, uz, N, 0., 0., Z0 + 2 * L, 0., 0., 0., None, st)
            ev[1].record()
            h.pxf_wolterprimary(x, y, z, l, m, n, ux, uy, uz, N, R0, Z0, PSI, None, st)
            ev[2].record()
            h.pxf_reflect(l, m, n, ux, uy, uz, N, None, st)
            ev[3].record()
            h.pxf_woltersecondary(x, y, z, l, m, n, ux, uy, uz, N, R0, Z0, PSI, None, st)
            ev[4].record()
            h.pxf_reflect(l, m, n, ux, uy, uz, N, None, st)
            ev[5].record()
            h.pxf_flat(x, y, z, l, m, n, ux, uy, uz, N, None, st)
            ev[6].record()
            torch.cuda.synchronize()
            if rnd > 0:
                for j, (k, _) in enumerate(BYTES):
                    times[name][k].append(ev[j].elapsed_time(ev[j + 1]))
    print("N=%d rounds=%d" % (N, ROUNDS))
    for name in libs:
        tot_med = 0.
        parts = []
        for k, b in BYTES:
            t = np.array(times[name][k])
            med, mn = float(np.median(t)), float(t.min())
            tot_med += med
            parts.append("%s %.3f/%.3f (%.0f)" % (k[:8], med, mn, b * N / (med * 1e-3) / 1e9))
        print("%-10s chain %.3f ms  %.0f GB/s | %s" % (name, tot_med, 512 * N / (tot_med * 1e-3) / 1e9, " | ".join(parts)))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
