"""The hand-run randomised checkers of tests/soak stay runnable: a few iterations of each inside the suite (the long
runs -- hundreds of random prescriptions, thousands of fuzz seeds -- are launched by hand, tests/soak/README.md)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,args,ok", [
    ("param_soak.py", ["6", "101"], "param soak: 6 iterations"),
    ("misc_soak.py", ["3", "102"], "misc soak: 3 iterations, 0 mismatches"),
    ("grating_soak.py", ["6", "103"], "grating soak: 0 mismatches"),
    ("source_soak.py", ["6", "104"], "source soak: 0 mismatches"),
    ("fast_soak.py", ["6", "105"], "tolerance-mode soak: worst RMS"),
    ("fan_soak.py", ["4", "106"], "fan soak: 4 iterations, 0 mismatches"),
])
def test_soak_scripts_run_clean(script, args, ok):
    r = subprocess.run([sys.executable, os.path.join(HERE, "soak", script)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ok in r.stdout, r.stdout[-2000:]
    assert "MISMATCH" not in r.stdout, r.stdout[-2000:]          # (bit-exact entries; W-S residuals print as TOL)
