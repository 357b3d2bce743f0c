import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
            ("native_cases", "chain_config2", "config1", "wrappers", "sources")}
