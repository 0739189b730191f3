import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
rng = np.random.default_rng(8192)
d5 = 10
sp5 = dev.KernelSpec(dev.K_SE, d5, list(0.5 + 0.03 * np.arange(d5)) + [1.0])
Mm = 8192
Cm = dev.points(ctx, rng.uniform(-1, 1, (Mm, d5)))
def best(f, reps=4):
    ts = []
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); out = f(); ctx.sync(); ts.append(time.perf_counter() - t0)
    return out, min(ts)
_, t = best(lambda: dev.kfill(ctx, sp5, Cm, nugget=0.1)); print("kfill %.2f ms" % (1e3 * t))
K = dev.kfill(ctx, sp5, Cm, nugget=0.1)
def fac():
    dev.kfill_into(ctx, sp5, Cm, K, nugget=0.1); return dev.potrf(ctx, K)
_, t = best(fac); print("kfill+potrf %.2f ms" % (1e3 * t))
_, t = best(lambda: dev.potri(ctx, K)); print("potri %.2f ms" % (1e3 * t))
ctx.profile(True); ctx.profile_reset()
_, t = best(lambda: dev.mi_greedy(ctx, sp5, Cm, 0.1, 8, 0), reps=1); 
p = ctx.profile_get(); ctx.profile(False)
print("mi_greedy %.2f ms" % (1e3 * t), {k: (v["launches"], round(v["ms"], 2)) for k, v in p.items() if v["launches"]})
_, t = best(lambda: dev.mi_greedy(ctx, sp5, Cm, 0.1, 8, 0)); print("mi_greedy best %.2f ms" % (1e3 * t))
