"""The C ABI used from plain C (tests/c/abi_smoke.c): built with gcc (C11) against include/pxf.h and
libpxf_hip.so, run as its own process -- no Python, no torch in the loop."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
EXE = os.path.join(ROOT, "build", "abi_smoke")
LIBDIR = os.path.join(ROOT, "pyxfocus_amd", "csrc")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    # plain gcc, C11: the header must be valid C; the HIP runtime is used through its C API only
    subprocess.run(["gcc", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-I" + os.path.join(ROOT, "include"), SRC, "-L" + LIBDIR, "-lpxf_hip", "-L/opt/rocm/lib",
                    "-lamdhip64", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE], check=True)


def test_c_program_builds_against_header():
    """CPU: the header is valid C (not just C++) and the program links against the library."""
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_program_runs_on_gpu():
    build()                      # always: a stale binary from an earlier source must not be what runs
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke ok" in out.stdout and "fused_mismatch_columns=0" in out.stdout
    assert "median_ok=1" in out.stdout and "sweeps=" in out.stdout
