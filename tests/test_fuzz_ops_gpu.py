"""The random-shape sweep of the widened-row operators (tests/fuzz_ops.py: fused SSIM, fused Adam, cube-map lookups
against the numpy oracle, grid_sample with border padding) as a driver-run test: the unit tests use fixed shapes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [0, 1])
def test_widened_row_operators_on_random_shapes(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_ops.py"), str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "failures: 0" in r.stdout, r.stdout[-2000:]
