import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no binaries (they are git-ignored): build the HIP library (hipcc
    cross-compiles without a GPU) and the oracle once, exactly as __graft_entry__.build() does."""
    need = [os.path.join(ROOT, "pyxfocus_amd", "csrc", "libpxf_hip.so"),
            os.path.join(ROOT, "oracle", "libpxf_oracle.so"), os.path.join(ROOT, "oracle", "libpxf_oracle_omp.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def port():
    """The C restatement (oracle/pxf_oracle.c) -- the checker."""
    from oracle import api
    return api.load_port()


@pytest.fixture(scope="session")
def ref():
    """The reference's own Fortran compiled in place (oracle/_ref); absent -> skip."""
    from oracle import api
    if not api.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference + flang)")
    return api.load_ref()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {k: np.load(os.path.join(d, k + ".npz")) for k in
            ("native_cases", "chain_config2", "config1", "wrappers", "sources", "wavefront", "design", "analyses", "config5", "alignment", "ws_heights", "retrace")}
