#!/usr/bin/env python3
"""CHILD-PROCESS helper of tests/test_gpu_parity.py::test_nontemporal_instantiations_in_child_process.

Must be started with PXF_NT_MIN_RAYS=0 in the environment: the library then launches the NT=true
instantiation of every ray kernel (non-temporal loads / stores -- what every bundle above 3.3M rays
runs) even for the small bundles the oracle can check.  Re-runs, on those code objects: every case of
oracle/cases.py plain and masked, the fused two-bounce program, the programs that hold mirrors, and the
Zernike sweep over all 15 radial orders plain and masked.  Prints "nt sweep ok" and exits 0, or raises."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import api, cases  # noqa: E402
from util import ALLCOLS, assert_cols, bits_equal, call_gpu, to_dev, to_host, zernike_order_check  # noqa: E402


def main():
    assert os.environ.get("PXF_NT_MIN_RAYS") == "0", "start me with PXF_NT_MIN_RAYS=0"
    port = api.load_port()
    rng = np.random.default_rng(17)
    ncase = 0
    for name, fn, cols, args in cases.cases(num=3001, seed=777):
        want = {k: v.copy() for k, v in cols.items()}
        cases.call(port, fn, want, args)
        t = to_dev(cols)
        call_gpu(fn, t, args)
        assert_cols(fn, to_host(t), want, ALLCOLS, context="nt:" + name)
        n = len(cols["x"])
        msk = rng.random(n) < .37
        want = {k: v.copy() for k, v in cols.items()}
        sub = {k: np.ascontiguousarray(v[msk]) for k, v in cols.items()}
        a2 = tuple(a[msk] if (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.shape == (n,)) else a
                   for a in args)
        cases.call(port, fn, sub, a2)
        for k in ALLCOLS:
            want[k][msk] = sub[k]
        t = to_dev(cols)
        call_gpu(fn, t, args, mask=torch.from_numpy(msk).cuda())
        got = to_host(t)
        assert_cols(fn, got, want, ALLCOLS, context="nt masked:" + name)
        for k in ALLCOLS:
            assert bits_equal(got[k][~msk], cols[k][~msk]).all(), (name, k)
        ncase += 1
    # the fused two-bounce program, NT instantiation
    import pyxfocus_amd as pxf
    from pyxfocus_amd import native
    c = cases.annulus_rays(rng, 5001, cases.primrad(cases.Z0), cases.primrad(cases.Z0 + cases.LMIR), 0.)
    want = {k: v.copy() for k, v in c.items()}
    cases.chain_config2(port, want)
    t = to_dev(c)
    with pxf.fused() as rec:
        cases.chain_config2(native, t)
    assert rec.launches == 1
    for k, v in to_host(t).items():
        assert bits_equal(v, want[k]).all(), k
    # the tolerance-mode program, NT instantiation: < 1e-10 mm RMS vs the oracle (config-2 chain)
    t = to_dev(c)
    with pxf.fused(fast=True) as rec:
        cases.chain_config2(native, t)
    assert rec.fast_launches == 1
    got = to_host(t)
    d2 = sum((got[k] - want[k]) ** 2 for k in "xyz")
    assert np.sqrt(d2[d2 < 1e-14].mean()) < 1e-10 and (d2 >= 1e-14).sum() <= 1, np.sqrt(d2.mean())
    # segmented programs (exact and tolerance mode), NT instantiations: the Arcus SPO stage vs the shell loop
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "examples"))
    import arcus_sector as A
    kw = dict(rows=(0, 5), modules_per_row=1, shell_stride=9)
    np.random.seed(8)
    a = A.defineSPOaperture(A.Api.amd(), 180, **kw)
    for fast in (False, True):
        np.random.seed(8)
        b = A.defineSPOaperture_segmented(180, source="numpy", fast=fast, **kw)
        for k, u, v in zip(ALLCOLS, a[0], b[0]):
            u, v = u.cpu().numpy(), v.cpu().numpy()
            if fast:
                assert np.abs(u - v).max() <= 1e-9 * (1 + np.abs(u).max()), k
            else:
                assert bits_equal(u, v).all(), k
    # programs that hold W-S / Legendre-Legendre / woltersine mirrors, NT instantiations: one launch == five launches
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    nmirror = 0
    for name, fn, cols, args in cases.cases(num=2001, seed=31):
        if not any(t_ in fn for t_ in ("ws", "ll", "woltersine")):
            continue

        def mchain(t):
            rays = [t[k] for k in ALLCOLS]
            tran.transform(rays, 1e-3, -2e-3, 0., 1e-6, -1e-6, 0.)
            call_gpu(fn, t, args)
            tran.reflect(rays)
            tran.itransform(rays, 1e-3, -2e-3, 0., 1e-6, -1e-6, 0.)
            surf.flat(rays)
        a_, b_ = to_dev(cols), to_dev(cols)
        mchain(a_)
        with pxf.fused() as rec:
            mchain(b_)
        assert rec.launches == 1 and rec.fused_ops == 5, (name, rec.launches)
        for k in ALLCOLS:
            assert bits_equal(a_[k].cpu().numpy(), b_[k].cpu().numpy()).all(), ("nt mirror program", name, k)
        nmirror += 1
    assert nmirror == 8
    for nmax in range(1, 16):
        zernike_order_check(port, nmax)
        zernike_order_check(port, nmax, masked=True)
    torch.cuda.synchronize()
    print("nt sweep ok: %d cases x (plain, masked), fused chain exact + fast, segmented SPO exact + fast, "
          "8 mirror programs, 15 Zernike orders x (plain, masked, fused)" % ncase)


if __name__ == "__main__":
    main()
