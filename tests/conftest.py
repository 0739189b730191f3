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

    # make_golden.py, make_golden_r2.py, make_golden_r4.py, make_golden_r6_ref.py (all import the reference)
    FILES = ["gpexp_golden", "gpexp_golden_r2", "gpexp_golden_r4", "gpexp_golden_r6_ref"]

    def __init__(self):
        d = os.path.join(ROOT, "tests", "golden")
        self.arrs = []
        self.index = {}
        for stem in self.FILES:
            self.arrs.append(np.load(os.path.join(d, stem + ".npz")))
            with open(os.path.join(d, stem + ".json")) as f:
                self.index.update(json.load(f))

    def __call__(self, case, name):
        key = "%s/%s" % (case, name)
        for a in self.arrs:
            if key in a.files:
                return a[key]
        raise KeyError(key)

    def has(self, case, name):
        return any("%s/%s" % (case, name) in a.files for a in self.arrs)

    def cases(self, typ):
        return sorted(c for c, v in self.index.items() if v["type"] == typ)

    def noise(self, case):
        nz = self.index[case]["noise"]
        return self(case, "noise") if nz == "array" else float(nz)


@pytest.fixture(scope="session")
def golden():
    return Golden()
