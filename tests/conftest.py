import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(params=["fp32", "split"])
def precision(request):
    """Run a GPU test under both arithmetics of the 3x3 MFMA convolutions (ops.set_precision): exact fp32 products,
    and the split-bf16 products (hi*hi + hi*lo + lo*hi, fp32 accumulate) that are the library default."""
    from effi_mvs_plus_amd import ops
    before = ops.get_precision()
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision(before)
