import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): oracle/cloudy_oracle.c via ctypes."""
    from oracle import cloudy_oracle

    cloudy_oracle.lib()
    return cloudy_oracle


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def cloudy():
    """The product package (cloudy.jl_amd/), loaded through __graft_entry__.load_package()."""
    import __graft_entry__ as ge

    return ge.load_package()


@pytest.fixture(scope="session")
def gpu_cloudy(cloudy):
    if not cloudy.device_count():
        pytest.fail("gpu-marked test but no HIP device is visible")
    return cloudy
