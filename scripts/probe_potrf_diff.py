"""Factor the same matrix with the variant in the environment and dump a checksum of L (A/B across GPX_POTRF_* settings)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(N)
X = dev.points(ctx, rng.uniform(-1, 1, (N, 8)))
sp = dev.KernelSpec(2, 8, [0.5, 1.0])
K = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
L = K.to_host(tri=1)
ctx.profile(True); ctx.profile_reset()
dev.kfill_into(ctx, sp, X, K, nugget=0.1); dev.potrf(ctx, K)
p = ctx.profile_get(); ctx.profile(False)
print("N=%d sum=%.17g sumsq=%.17g L[N-1,7]=%.17g gemm launches=%d leaf=%d" % (N, L.sum(), (L * L).sum(), L[N - 1, 7], p["gemm"]["launches"], p["leaf"]["launches"]))
