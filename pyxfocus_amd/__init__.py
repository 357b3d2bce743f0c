"""pyxfocus_amd -- MI355X-native X-ray ray-trace core behind PyXFocus' call signatures.

    import pyxfocus_amd.sources as sources
    import pyxfocus_amd.surfaces as surf
    import pyxfocus_amd.transformations as tran
    import pyxfocus_amd.analyses as anal

mirror ``PyXFocus.sources / surfaces / transformations / analyses`` for the hot path
(SURVEY.md section 8): rays are the reference's ten-column SoA list, held as float64 tensors in
HBM; every per-ray operation is a hand-written gfx950 HIP kernel reached through the C ABI of
include/pxf.h (pyxfocus_amd/csrc/libpxf_hip.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (does not load the shared library until first use)
from ._fused import FastModeWarning, compile_segments, flush, fused  # noqa: E402,F401
from ._graph import capture  # noqa: E402,F401
