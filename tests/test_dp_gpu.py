"""The data-parallel reduction on REAL RCCL (one rank: all a single GPU allows; two ranks need two devices): see
tests/dp_gpu_worker.py.  The N-rank arithmetic is covered on gloo by tests/test_dp.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_reduction_of_rasterizer_gradients_on_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(34000 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), port], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DP_GPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
