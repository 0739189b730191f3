"""Seeded random sweep over the whole device path (-m gpu): kernel kind x dimension 1..32 x ragged sizes x offsets x scalar /
per-point noise, each configuration against the oracle at north_star's tolerances (K 1e-13; coeff / mean / variance /
log-marginal / IVAR 1e-10 on well-conditioned problems).  The named tests elsewhere pin specific edges; this one walks the
combinations nobody thought of (the d -> MFMA K-step classes, one- and two-tile rectangular fills with an odd number of
column tiles, centring with every kind, N below / at / above the 128 padding and the 2048 out-of-place solve)."""
import numpy as np
import pytest

from oracle import gpexp_oracle as orc
from helpers import rel

pytestmark = pytest.mark.gpu

KINDS = ["se", "matern32", "matern52", "mehler"]


@pytest.fixture(scope="module")
def dev():
    from gpexp_amd import device
    return device


@pytest.fixture(scope="module")
def ctx(dev):
    return dev.context()


def _config(i):
    rng = np.random.default_rng(9000 + i)
    kind = KINDS[i % 4]
    d = int(rng.choice([1, 2, 3, 5, 6, 7, 10, 11, 18, 19, 32]))
    n = int(rng.choice([1, 2, 17, 63, 64, 65, 127, 128, 129, 200, 257, 383, 513, 700]))
    m = int(rng.choice([1, 5, 63, 64, 65, 129, 191, 320]))
    if i % 9 == 0:
        n = int(rng.choice([2047, 2048, 2100]))     # around the switch to the out-of-place block-inverse solve
        d = min(d, 6)
    offset = float(rng.choice([0.0, 0.0, 3.0, 250.0])) if kind != "mehler" else 0.0
    if kind == "se":
        s = dict(kind="se", d=d, cl=list(rng.uniform(0.6, 1.4, d) * np.sqrt(d)), signalSize=float(rng.uniform(0.5, 2.0)))
    elif kind == "mehler":
        s = dict(kind="mehler", d=d, t=list(rng.uniform(0.05, 0.5, d) / np.sqrt(d)))
    else:
        s = dict(kind=kind, d=d, rho=float(rng.uniform(0.8, 1.6) * np.sqrt(d)), signalSize=float(rng.uniform(0.5, 2.0)))
    X = rng.uniform(-1, 1, (n, d)) + offset
    Z = rng.uniform(-1, 1, (m, d)) + offset
    y = rng.standard_normal(n)
    noise = float(rng.choice([0.05, 0.2])) if i % 3 else rng.uniform(0.05, 0.3, n)
    return s, X, y, Z, noise


@pytest.mark.parametrize("i", range(48))
def test_random_configuration_against_oracle(dev, ctx, i):
    from test_gpu_parity import spec_of
    s, X, y, Z, noise = _config(i)
    spec = spec_of(dev, s)
    Xd, Zd = dev.points(ctx, X), dev.points(ctx, Z)
    Kd = dev.kfill(ctx, spec, Xd, nugget=noise)
    K = orc.cov_matrix(s, X, noise, row_loop=False)
    assert rel(Kd.to_host(), K) <= 1e-13
    assert rel(dev.kfill(ctx, spec, Xd, Z=Zd).to_host(), orc.cross_matrix(s, Z, X).T) <= 1e-13
    model = orc.fit(s, X, y, noise)
    L = dev.potrf(ctx, Kd)
    cond = np.linalg.cond(K)
    tol = 1e-10 * max(1.0, cond / 1e3)   # north_star's 1e-10 is stated for cond ~ 1e3
    alpha = dev.potrs(ctx, L, y)
    assert rel(alpha, model["coeff"]) <= tol
    ll = -0.5 * float(y @ alpha) - 0.5 * dev.logdet(ctx, L) - len(y) / 2.0 * np.log(2 * np.pi)
    assert ll == pytest.approx(orc.loglike(s, X, y, noise), rel=tol, abs=tol)
    mean, var = dev.posterior(ctx, spec, L, Xd, alpha, Zd)
    mo, vo = orc.posterior(s, model, Z, compvar=1)
    assert rel(mean, mo) <= tol
    assert np.max(np.abs(np.abs(var) - vo)) <= tol * max(1.0, np.max(np.abs(vo)))
    if np.isscalar(noise):
        assert abs(dev.ivar(ctx, spec, L, Xd, Zd)) == pytest.approx(orc.ivar(s, X, Z, noise), rel=tol, abs=tol)
