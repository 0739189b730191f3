import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
n, b, d = 16384, 512, 8
rng = np.random.default_rng(n)
X = rng.uniform(-1, 1, (n, d))
sp = dev.KernelSpec(dev.K_SE, d, list(0.4 + 0.05 * np.arange(d)) + [1.0])
Xd = dev.points(ctx, X)
L = dev.potrf(ctx, dev.kfill(ctx, sp, Xd, nugget=0.1))
X2 = X.copy(); X2[n - b:] = rng.uniform(-1, 1, (b, d)); X2d = dev.points(ctx, X2)
keep = n - b
ts = []
for it in range(6):
    ctx.sync(); t0 = time.perf_counter()
    Lr = dev.refit_rows(ctx, sp, X2d, 0.1, L, keep); ctx.sync(); ts.append(time.perf_counter() - t0)
    ld = dev.logdet(ctx, Lr); Lr.free()
print("refit N=%d b=%d: best %.2f ms logdet %.12g env %s" % (n, b, 1e3 * min(ts[1:]), ld, {k: v for k, v in os.environ.items() if k.startswith("GPX_")}), flush=True)
