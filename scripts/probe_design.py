"""C3 / C5-sized runs of the design kernels (parity is covered by the fixtures; here: scale, timing, properties)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()

def t(f):
    ctx.sync(); t0 = time.perf_counter(); r = f(); ctx.sync(); return r, 1e3 * (time.perf_counter() - t0)

# ---- C3: N=16384, d=8 ARD-SE, 65536 candidates, nMC=4096 ----
N, d, M, nmc = 16384, 8, 65536, 4096
rng = np.random.default_rng(16384)
Xh = rng.uniform(-1, 1, (N, d)); y = np.sin(2*np.pi*Xh.sum(1)/d) + np.sqrt(0.1)*rng.standard_normal(N)
Ch = rng.uniform(-1, 1, (M, d)); Zh = rng.uniform(-1, 1, (nmc, d))
sp = dev.KernelSpec(dev.K_SE, d, list(0.4 + 0.05*np.arange(d)) + [1.0])
X, C, Z = dev.points(ctx, Xh), dev.points(ctx, Ch), dev.points(ctx, Zh)
K, tk = t(lambda: dev.kfill(ctx, sp, X, nugget=0.1)); _, tp = t(lambda: dev.potrf(ctx, K))
print("C3 fit: kfill %.2f ms potrf %.1f ms" % (tk, tp), flush=True)
(best, costs), tg = t(lambda: dev.greedy_ivar_step(ctx, sp, K, X, C, Z, 0.1))
iv0 = abs(dev.ivar(ctx, sp, K, X, Z))
print("C3 greedy-IVAR step over %d candidates, nMC=%d: %.1f ms  best=%d cost=%.9g (IVAR before %.9g) all costs <= before: %s" %
      (M, nmc, tg, best, costs[best], iv0, bool(np.all(costs <= iv0 * (1 + 1e-12)))), flush=True)
# cross-check the winner by an actual refit with that candidate added
X2 = dev.points(ctx, np.vstack((Xh, Ch[best:best+1])))
K2 = dev.potrf(ctx, dev.kfill(ctx, sp, X2, nugget=0.1))
iv1 = abs(dev.ivar(ctx, sp, K2, X2, Z))
print("   refit check: IVAR(X u c_best) = %.12g vs rank-one cost %.12g  rel diff %.2e" % (iv1, costs[best], abs(iv1-costs[best])/iv1), flush=True)
idx, tv = t(lambda: dev.greedy_var(ctx, sp, C, 16))
print("C3 greedy variance 16 of %d: %.2f ms idx=%s distinct=%s" % (M, tv, list(idx), len(set(idx)) == 16), flush=True)
idx64, tv64 = t(lambda: dev.greedy_var(ctx, sp, C, 256))
print("   256 steps: %.1f ms distinct=%s prefix-consistent=%s" % (tv64, len(set(idx64)) == 256, list(idx64[:16]) == list(idx)), flush=True)
del K, K2
# ---- C5-ish: MI greedy 8 of 8192; lml gradient at N=8192, d=10 ----
d5 = 10
sp5 = dev.KernelSpec(dev.K_SE, d5, list(0.5 + 0.03*np.arange(d5)) + [1.0])
Cm = dev.points(ctx, rng.uniform(-1, 1, (8192, d5)))
(mi_idx, ratios), tm = t(lambda: dev.mi_greedy(ctx, sp5, Cm, 0.1, 8, 0))
print("C5 MI greedy 8 of 8192: %.1f ms idx=%s ratios decreasing-ish=%s" % (tm, list(mi_idx), [float("%.4g" % r) for r in ratios]), flush=True)
N5 = 8192
X5h = rng.uniform(-1, 1, (N5, d5)); y5 = np.sin(2*np.pi*X5h.sum(1)/d5) + np.sqrt(0.1)*rng.standard_normal(N5)
X5 = dev.points(ctx, X5h)
def ll(hyp, noise):
    s = dev.KernelSpec(dev.K_SE, d5, hyp)
    L = dev.potrf(ctx, dev.kfill(ctx, s, X5, nugget=noise)); a = dev.potrs(ctx, L, y5)
    return -0.5*y5@a - 0.5*dev.logdet(ctx, L) - N5/2*np.log(2*np.pi), L, a, s
hyp = list(0.5 + 0.03*np.arange(d5)) + [1.0]
v, L, a, s = ll(hyp, 0.1)
g, tgm = t(lambda: dev.lml_grad(ctx, s, L, X5, a))
print("C5 lml_grad N=%d d=%d: %.1f ms (incl. potri)" % (N5, d5, tgm), flush=True)
for k in (0, 5, 10):
    h = 1e-5; hp = list(hyp); hm = list(hyp); hp[k] += h; hm[k] -= h
    fd = (ll(hp, 0.1)[0] - ll(hm, 0.1)[0]) / (2*h)
    print("   d/dhyp[%d]: analytic %.8g  central diff %.8g  rel %.1e" % (k, g[k], fd, abs(g[k]-fd)/abs(fd)), flush=True)
fdn = (ll(hyp, 0.1+1e-6)[0] - ll(hyp, 0.1-1e-6)[0]) / 2e-6
print("   d/dnoise: analytic %.8g  central diff %.8g" % (g[-1], fdn), flush=True)
