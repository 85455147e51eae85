"""pytest configuration: the `gpu` marker and shared helpers.

`-m "not gpu"` (CPU container): oracle vs golden vectors / vs the compiled reference, host
logic, the C-ABI library loads and exports every declared symbol.
`-m gpu` (MI355X box): the parity tests proper -- the HIP path, called through the C ABI,
against the CPU oracle on the same inputs.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import pyoracle
    return pyoracle.oracle()


@pytest.fixture(scope="session")
def ref_lib():
    """the real reference (development container only); tests that need it skip elsewhere"""
    import pyoracle
    lib = pyoracle.reference()
    if lib is None:
        pytest.skip("oracle/_ref/libbbo_ref.so not built (no /root/reference here)")
    return lib


@pytest.fixture(scope="session")
def hip():
    """the product library; GPU tests FAIL (not skip) when it or the device is missing"""
    import bboptpy_amd
    from bboptpy_amd import _ffi
    lib = _ffi.lib()
    assert lib.bbo_device_count() > 0, "no HIP device visible on a -m gpu run"
    return bboptpy_amd
