import os
import sys

import pytest

try:  # torch bundles its own libamdhip64.so.7: load it first so the process ends up with ONE HIP runtime
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:      # tests/assets.py, tests/io_common.py, the second sources
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# Every device buffer the library allocates under test starts as garbage (0xCD) instead of whatever the allocator hands out — fresh device memory reads as zero,
# recycled memory does not, and a value read before it is written must show here rather than in a long-lived process (round 4: an all-NaN leaf box left an entry of
# the builder's work list unwritten; it only ever faulted on recycled memory).
os.environ.setdefault("MSNE_DEBUG_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): builds oracle/liborc.so on first use."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def gpu_api():
    """The product: HIP library through its C ABI.  Fails (not skips) if the library is missing."""
    from moonshine_amd import api
    api.load_library()
    return api
