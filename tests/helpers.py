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


def elementwise(a, b, floor=1e-3):
    """max_i |a_i - b_i| / max(|b_i|, floor * max|b|): every entry is held to a RELATIVE bound of its own size, down to
    entries `floor` times the largest (below that the bound is relative to floor * max|b|: the reference's own pinv/SVD
    results carry an absolute error of that order in the variances, gp.py:142-144)."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    scale = np.maximum(np.abs(b), floor * max(np.max(np.abs(b)), 1e-300))
    return float(np.max(np.abs(a - b) / scale))


def c4_lite_inputs(ix):
    """Inputs of the `c4_lite` fixture, regenerated from its seed exactly as tests/golden/make_golden_r4.py drew them."""
    rng = np.random.default_rng(ix["seed"])
    N, M, d = ix["N"], ix["M"], ix["kernel"]["d"]
    X = rng.uniform(-1, 1, (N, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(ix["noise"]) * rng.standard_normal(N)
    Z = rng.uniform(-1, 1, (M, d))
    return X, y, Z
