"""Wall times of BASELINE configs C2 and C3 through this build (MI355X box), second run of each (pools warm):
C2  N=4096, d=3 iso-SE, noise 0.05: GP.train + evaluate(4096 test points, compvar=1) + computeLogLike through the gpExp class API
    (the reference itself: 2.07 s fill + 9.75 s pinv + variance loop ~ 25 s in the build container)
C3  N=16384, d=8 ARD-SE, noise 0.1: fit, ONE greedy-IVAR step over 65 536 candidates with 4096 MC points, 16 greedy-variance picks"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
from gpExp.kernels import KernelSquaredExponential
from gpExp.gp import GP
ctx = dev.context()
out = {}
def timed(fn):
    ctx.sync(); t0 = time.perf_counter(); r = fn(); ctx.sync(); return time.perf_counter() - t0, r
# C2
N, d, M = 4096, 3, 4096
rng = np.random.default_rng(4096)
X = rng.uniform(-1, 1, (N, d)); y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(0.05) * rng.standard_normal(N); Z = rng.uniform(-1, 1, (M, d))
for rep in range(2):
    g = GP(KernelSquaredExponential([0.2], 1.0, d), 0.05)
    t_train, _ = timed(lambda: g.train(X, y))
    t_eval, _ = timed(lambda: g.evaluate(Z, compvar=1))
    t_ll, ll = timed(lambda: g.computeLogLike(X, y))
out["C2"] = {"train_s": t_train, "evaluate_4096_compvar1_s": t_eval, "computeLogLike_s": t_ll, "total_s": t_train + t_eval + t_ll, "loglike": float(ll)}
# C3
N, d, M, nmc = 16384, 8, 65536, 4096
rng = np.random.default_rng(16384)
Xh = rng.uniform(-1, 1, (N, d)); Ch, Zh = rng.uniform(-1, 1, (M, d)), rng.uniform(-1, 1, (nmc, d))
sp = dev.KernelSpec(dev.K_SE, d, list(0.4 + 0.05 * np.arange(d)) + [1.0])
Xp, C, Zp = dev.points(ctx, Xh), dev.points(ctx, Ch), dev.points(ctx, Zh)
for rep in range(2):
    t_fit, K = timed(lambda: dev.potrf(ctx, dev.kfill(ctx, sp, Xp, nugget=0.1)))
    t_step, (best, costs) = timed(lambda: dev.greedy_ivar_step(ctx, sp, K, Xp, C, Zp, 0.1))
    t_gv, picks = timed(lambda: dev.greedy_var(ctx, sp, C, 16))
    if rep == 0:
        del K
out["C3"] = {"fit_s": t_fit, "greedy_ivar_step_65536_candidates_s": t_step, "greedy_variance_16_picks_s": t_gv, "total_s": t_fit + t_step + t_gv,
             "best": int(best), "cost": float(costs[best])}
print(json.dumps(out))
