import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


SESSION_T0 = __import__("time").time()  # tests/test_zz_full_grid_parity.py sizes its CPU-side budgets against the suite's elapsed time
