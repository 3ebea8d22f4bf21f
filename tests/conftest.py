import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc/g++."""
    from tests import _oracle
    return _oracle.load()


@pytest.fixture(scope="session")
def gpu_context():
    """One GPUContext for the whole GPU session; fails loudly when the HIP library or device is missing."""
    import vkradixsort_amd as vrs
    from vkradixsort_amd import capi
    ctx = vrs.GPUContext(int(os.environ.get("VRS_DEVICE", "0")))
    ctx.init()
    # the tests that share this context count kernels sort by sort: every pool sort samples (the kept layouts of
    # VRS_TUNE_MSD_POOL_REUSE_LAYOUT have tests of their own, on contexts of their own: tests/test_gpu_pool.py)
    ctx.setTuning(capi.VRS_TUNE_MSD_POOL_REUSE_LAYOUT, 0)
    yield ctx
    ctx.shutdown()
