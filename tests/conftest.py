import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# single-threaded BLAS like the reference's own conftest (tests/conftest.py:16-18)
for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(v, "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_DEVICES = None


def pytest_runtest_setup(item):
    """`gpu` tests need a device.  Where the HIP library loads but sees none (the build container) they are skipped, so a
    plain `pytest tests` works there; a library that is missing or fails to load is an ERROR, never a skip — on the GPU box
    the product path must fail loudly."""
    global _DEVICES
    if item.get_closest_marker("gpu") is None:
        return
    if _DEVICES is None:
        from pyqmc_amd import _ffi

        _DEVICES = int(_ffi.lib().pqa_device_count())
    if _DEVICES == 0:
        import pytest

        pytest.skip("no ROCm device visible (gpu tests run on the MI355X box: pytest -m gpu)")
