"""Does device memory stay flat over an optimiser-style loop of the class API?  hipMemGetInfo before / after N iterations of
train + evaluate + IVAR cost + gradient + greedy step (sizes vary, so the exact-size pool cannot hide growth by reuse)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpExp.kernels import KernelSquaredExponential
from gpExp.gp import GP
from gpExp.approximation import Space
from gpExp.experimentalDesign import costFunctionGP_IVAR, performGreedyVarExperimentalDesign
from gpexp_amd import device as dev
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 2**20
ctx = dev.context()
rng = np.random.default_rng(0)
d = 2
space = Space(d, lambda size: rng.uniform(-1, 1, size), lambda p: 0.25 * np.ones(len(p)))
def one(i):
    n = 200 + (i % 7) * 37
    X = rng.uniform(-1, 1, (n, d)); y = np.sin(X.sum(1))
    g = GP(KernelSquaredExponential([0.4, 0.5], 1.0, d), 1e-2)
    g.train(X, y)
    m, v = g.evaluate(rng.uniform(-1, 1, (500 + (i % 5) * 100, d)), compvar=1)
    cf = costFunctionGP_IVAR(g, n, space, mcPoints=rng.uniform(-1, 1, (1000, d)))
    c = cf.evaluate(X); gr = cf.derivative(X)
    ll = g.computeLogLike(X, y)
    return float(m[0] + v[0] + c + gr[0] + ll)
for i in range(20): one(i)
ctx.sync(); f0 = free_mb()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300): one(i)
ctx.sync(); f1 = free_mb()
ctx.trim(); f2 = free_mb()
print("free MiB: after warm-up %.1f, after the loop %.1f (delta %.1f), after trim %.1f (delta vs warm-up %.1f)" % (f0, f1, f1 - f0, f2, f2 - f0))
