import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


from quantum_simulator_amd.golden import (ALL_CASES, GOLDEN_INDEX, SMALL_CASES, GoldenCase, case_arch,  # noqa: E402,F401
                                          case_circuit, load_case)


@pytest.fixture(params=ALL_CASES)
def golden_case(request):
    return load_case(request.param)


@pytest.fixture(params=SMALL_CASES)
def small_case(request):
    return load_case(request.param)


