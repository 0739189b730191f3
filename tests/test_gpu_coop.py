"""Round 5: the ONE-LAUNCH factorisation of a diagonal block of order <= 1024 (potrf_coop_kernel, chol.hip:
a chain workgroup factoring the 128-leaves, helper workgroups doing strips and rank-128 updates, hand-over through flags in
global memory).  It measured no faster than the launch chain and is off by default (GPX_POTRF_COOP=1 selects it; the switch is read
once per process, so the cases run in a child process): same factor as LAPACK to 1e-13, leaf inverses included (a solve), the
non-positive-definite report, the rank-deficient skip policy, and a 4096-order matrix whose diagonal blocks take it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from gpexp_amd import device as dev
from gpexp_amd._lib import NotPositiveDefinite
ctx = dev.context()
rng = np.random.default_rng(5)
for n in (256, 384, 512, 768, 1024, 1000, 4096):
    A = rng.standard_normal((n, n + 7)); A = A @ A.T / n + np.eye(n)
    y = rng.standard_normal(n)
    L = dev.potrf(ctx, dev.DeviceMatrix.from_host(ctx, A, pad=True))
    Lh = np.tril(L.to_host()[:n, :n]); Lr = np.linalg.cholesky(A)
    e1 = np.abs(Lh - Lr).max() / np.abs(Lr).max()
    e2 = np.abs(dev.potrs(ctx, L, y) - np.linalg.solve(A, y)).max() / np.abs(np.linalg.solve(A, y)).max()
    assert e1 < 1e-13 and e2 < 1e-10, (n, e1, e2)
    print("ok n=%%d  factor %%.1e  solve %%.1e" %% (n, e1, e2), flush=True)
# not positive definite: the first bad pivot's index comes back through the cooperative kernel's leaves
A = rng.standard_normal((640, 700)); A = A @ A.T / 640 + np.eye(640); A[300, 300] = -1.0
try:
    dev.potrf(ctx, dev.DeviceMatrix.from_host(ctx, A, pad=True))
    raise SystemExit("a negative pivot went unnoticed")
except NotPositiveDefinite as e:
    assert e.pivot == 301, e.pivot
    print("ok non-PD pivot", e.pivot, flush=True)
# a duplicated point dropped by the skip policy (GP._factor's retry)
X = rng.uniform(-1, 1, (700, 3)); X[511] = X[17]
sp = dev.KernelSpec(dev.K_SE, 3, [0.5, 0.6, 0.7, 1.0])
prev = dev.potrf_policy(ctx, 1e-13, True)
K = dev.potrf(ctx, dev.kfill(ctx, sp, dev.points(ctx, X), nugget=0.0))
assert dev.potrf_dropped(ctx) >= 1
dev.potrf_policy(ctx, *prev)
print("ok dropped", dev.potrf_dropped(ctx), flush=True)
print("COOP_OK", flush=True)
'''


def test_one_launch_diagonal_block_factorisation():
    env = dict(os.environ, GPX_POTRF_COOP="1", GPX_COOP_HELPERS="12")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "COOP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok n=") == 7
