"""L2 experiment for the NT rectangular GEMM: the same launch with (a) separate A and B, (b) B = the very
buffer of A, (c) the NN form of the same product.  Run under
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d <dir> -- python3 scripts/probe_l2_nt.py
and read the dispatches in order (scripts/pmc_by_dispatch.py <dir>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
m = n = int(os.environ.get("PROBE_MN", "16384"))
k = int(os.environ.get("PROBE_K", "4096"))
rng = np.random.default_rng(3)
blk = rng.standard_normal((1024, k))
A = dev.DeviceMatrix.from_host(ctx, np.tile(blk, (m // 1024, 1)))
B = dev.DeviceMatrix.from_host(ctx, np.tile(blk[::-1].copy(), (n // 1024, 1)))
Bt = dev.DeviceMatrix.from_host(ctx, np.tile(blk[::-1].T.copy(), (1, n // 1024)))     # k x n
C = dev.DeviceMatrix.zeros(ctx, m, n)
for rep in range(2):
    dev.dbg_gemm(ctx, A, B, C, 1, 1, 0)      # NT, separate operands
    dev.dbg_gemm(ctx, A, A, C, 1, 1, 0)      # NT, B is A itself
    dev.dbg_gemm(ctx, A, Bt, C, 0, 1, 0)     # NN
print("done")
