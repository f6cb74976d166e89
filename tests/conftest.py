import os
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Estimators writes estimates.<n>.h5 into the working directory by default, as the reference does; the suite keeps the
# in-memory rows only unless a test names a file (the default is exercised in tests/test_dropin_cpu.py under tmp_path)
os.environ.setdefault("AFQ_ESTIMATES_FILE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return dict(numpy.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
