import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nbody-eurohpc_amd"))   # -> import murbhip
sys.path.insert(0, os.path.join(ROOT, "oracle"))              # -> import oracle (checker only)

# liboracle/libmurbref are linked with -ffast-math like the reference executable, which turns on
# flush-to-zero for the process; numpy notices and warns once.
warnings.filterwarnings("ignore", message=".*smallest subnormal.*")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import murbhip
        return murbhip.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """The HIP product library with a usable device — fails loudly (no skip) when missing,
    so that a GPU run can never pass on a silent fallback."""
    import murbhip
    murbhip.lib()
    assert murbhip.device_count() > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return murbhip


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.lib()
    return oracle


GOLDEN = os.path.join(ROOT, "tests", "golden")


def free_port():
    """A TCP port that is free right now on 127.0.0.1 (torch.distributed.run rendezvous of the tests)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
