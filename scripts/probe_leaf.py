"""Time the 128x128 leaf kernel alone: potrf of a 128x128 SPD matrix = one leaf launch (profile class 'leaf')."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
rng = np.random.default_rng(0)
A = rng.standard_normal((128, 128)); A = A @ A.T + 128 * np.eye(128)
for it in range(3):
    K = dev.DeviceMatrix.from_host(ctx, A)
    ctx.profile(True); ctx.profile_reset()
    for r in range(20):
        try:
            dev.potrf(ctx, K)
        except Exception:
            pass
    p = ctx.profile_get()["leaf"]; ctx.profile(False)
print("leaf dbg=%s: %.2f us per launch" % (os.environ.get("GPX_LEAF_DBG", "0"), 1e3 * p["ms"] / p["launches"]))
