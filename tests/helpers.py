"""Small shared pieces of the test-suite (data, not product code)."""
import numpy as np


class NoiseFunc(object):
    """Heteroscedastic noise model with the interface the reference expects of space.noiseFunc (demo2.py:45-58:
    callable + .deriv); the same definition tests/golden/make_golden_r2.py fed to the reference."""

    def __init__(self, d):
        self.dimension = d
        self.w = 0.5 + 0.25 * np.arange(d)

    def __call__(self, points):
        return 0.01 + 0.05 * np.sum(self.w[None, :] * points ** 2.0, axis=1)

    def deriv(self, points):
        return 0.1 * self.w[None, :] * points


def rel(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
