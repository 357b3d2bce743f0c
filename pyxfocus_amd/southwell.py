"""Southwell wavefront reconstruction from slope maps -- ``PyXFocus/southwell.py`` with the
reconstructor on the device (``native.reconstruct`` = reconstruct.f95 by anti-diagonals, bit-identical
sweeps).  Grids may be NumPy arrays or tensors; the result is a device tensor."""
import numpy as np
import torch

from . import native


def padArrays(imglist):
    """Pad each 2-D array with a border of 100s (the reconstructor's NaN sentinel); returns
    Fortran-ordered device grids (southwell.py:6-18)."""
    out = []
    for i in imglist:
        t = torch.as_tensor(i, dtype=torch.float64)
        dev = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
        g = native._fgrid(t.shape[0] + 2, t.shape[1] + 2, dev)
        g.fill_(100.)
        g[1:-1, 1:-1] = t.to(dev)
        out.append(g)
    return out


def southwell(gx, gy, criteria, h, maxiter=10000):
    """Reconstruct the phase whose x / y slopes are gx / gy (NaN = missing data); returns -phase
    with NaN where data were missing (southwell.py:20-45).  As in the reference, `criteria` and
    `h` are accepted but the reconstructor is called with 1e-10 and 1. (southwell.py:38), and gx,
    gy are modified in place where data are missing when they are NumPy arrays or tensors."""
    tx = gx if isinstance(gx, torch.Tensor) else torch.from_numpy(np.asarray(gx))
    ty = gy if isinstance(gy, torch.Tensor) else torch.from_numpy(np.asarray(gy))
    ind = torch.logical_or(torch.isnan(tx), torch.isnan(ty))
    tx[ind] = 100.
    ty[ind] = 100.
    phase = torch.zeros(tx.shape, dtype=torch.float64, device=tx.device)
    phase[ind] = 100.
    phase, px, py = padArrays([phase, tx, ty])
    native.reconstruct(px, py, 1e-10, 1., phase, maxiter)      # phase is updated in place (intent inout)
    out = phase[1:-1, 1:-1].clone()
    out[ind.to(out.device)] = float("nan")
    return -out
