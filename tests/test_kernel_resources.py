"""Audit of the built code objects (CPU test: reads the AMDGPU metadata notes of libpxf_hip.so).

Round 1 found Zernike kernel instantiations that returned WRONG results whenever the register
allocator had to spill heavily (AGPRs, or scalar registers spilled while the vector file was full):
DESIGN.md 3.1.  This test keeps every kernel of the library out of that regime: no kernel may use
AGPRs, and vector-register spills are allowed only for the short list below (a handful of registers
each, all covered by parity tests over their whole parameter range)."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pyxfocus_amd", "csrc", "libpxf_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# demangled-name prefix -> most vector registers it may spill
VGPR_SPILL_ALLOWED = {
    "pxf::ZernOpSel<7,": 8,          # the 36-term kernel at three waves per SIMD
    "pxf::ZernOpSel<8,": 4,
    "pxf::ZernOpSel<6,": 8, "pxf::ZernOpSel<5,": 8,
    "pxf::program_zern_kernel<7,": 8,
}


def _kernels():
    tmp = tempfile.mkdtemp()
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(LIB, lib)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True)
        out = []
        for co in glob.glob(lib + ".*hipv4*"):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True,
                                   capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                f = {k: int(v) for k, v in re.findall(r"\.(agpr_count|sgpr_spill_count|vgpr_spill_count|vgpr_count|"
                                                      r"private_segment_fixed_size):\s+(\d+)", ".agpr_count:" + blk)}
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                out.append((name, f))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="needs the ROCm LLVM tools")
def test_no_kernel_in_the_heavy_spill_regime():
    ks = _kernels()
    assert len(ks) > 300                                   # every instantiation of the library
    names = subprocess.run(["c++filt"], input="\n".join(n for n, _ in ks), capture_output=True, text=True).stdout.split("\n")
    bad = []
    for (mangled, f), name in zip(ks, names):
        key = None
        for k in VGPR_SPILL_ALLOWED:
            if k in name:
                key = k
        if f.get("agpr_count", 0) != 0:
            bad.append(("AGPRs", name[:100], f))
        if f.get("vgpr_spill_count", 0) > (VGPR_SPILL_ALLOWED[key] if key else 0):
            bad.append(("VGPR spills", name[:100], f))
        if f.get("vgpr_count", 0) > 256:
            bad.append(("more than 256 registers", name[:100], f))
    assert not bad, bad[:10]
