"""Where does the event span of the symmetric fill grow inside a fit?  Per iteration: event span of the kfill class, with and
without a sync right behind the fill; kernel durations come from the rocprofv3 trace of the same run."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
N = 32768
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, 8)))
y = dev.padded_vector(ctx, rng.standard_normal(N)); a = dev.padded_vector(ctx, np.zeros(N))
sp = dev.KernelSpec(dev.K_MATERN52, 8, [0.5, 1.0])
K = dev.DeviceMatrix.zeros(ctx, N, N)
def fit(sync_after_fill):
    dev.kfill_into(ctx, sp, X, K, nugget=0.1)
    if sync_after_fill: ctx.sync()
    dev.potrf(ctx, K); dev.potrs_dev(ctx, K, y, a); dev.logdet(ctx, K)
fit(False); ctx.sync()
for mode in (False, True, False, True):
    for it in range(3):
        ctx.profile(True); ctx.profile_reset()
        fit(mode)
        p = ctx.profile_get(); ctx.profile(False)
        print("sync behind the fill: %-5s  kfill event span %.3f ms  (leaf %.2f, trsv %.2f)" % (mode, p["kfill"]["ms"], p["leaf"]["ms"], p["trsv"]["ms"]), flush=True)
