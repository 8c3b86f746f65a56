"""Bounded run of the randomised GPU-vs-oracle parity fuzz (tools/fuzz_parity.py): random T, M, P and codebooks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_parity_60_cases():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", "20261003"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "60 cases, 0 mismatches" in r.stdout
