import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(autouse=True)
def _restore_native_modes():
    """The library's mode switches (blend implementation, reference binning, spin wait, debug, markers, profiling) are
    process-wide: whatever a test sets is put back to the defaults afterwards, also when the test fails."""
    yield
    import torch
    if torch.cuda.is_available():
        import gs2m_native
        gs2m_native.reset_modes()
