import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's own DataLoader pin-memory thread calls Tensor.pin_memory(device) / is_pinned(device), which this
    # torch deprecates: thousands of identical warnings per run of the disk-dataset loop test
    config.addinivalue_line("filterwarnings", "ignore:The argument 'device' of Tensor.:DeprecationWarning")


@pytest.fixture(scope="session")
def raster_oracle():
    from oracle.gsr_oracle import RasterOracle
    return RasterOracle()


@pytest.fixture(scope="session")
def raster_oracle_f64():
    from oracle.gsr_oracle import RasterOracle
    return RasterOracle(f64=True)
