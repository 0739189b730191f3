import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """tests/golden/gpexp_golden.npz: vectors produced by the reference itself (make_golden.py)."""

    def __init__(self):
        d = os.path.join(ROOT, "tests", "golden")
        self.arr = np.load(os.path.join(d, "gpexp_golden.npz"))
        with open(os.path.join(d, "gpexp_golden.json")) as f:
            self.index = json.load(f)

    def __call__(self, case, name):
        return self.arr["%s/%s" % (case, name)]

    def has(self, case, name):
        return "%s/%s" % (case, name) in self.arr.files

    def cases(self, typ):
        return sorted(c for c, v in self.index.items() if v["type"] == typ)

    def noise(self, case):
        nz = self.index[case]["noise"]
        return self(case, "noise") if nz == "array" else float(nz)


@pytest.fixture(scope="session")
def golden():
    return Golden()
