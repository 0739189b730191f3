"""Back-to-back launches of one GEMM shape (no host sync in between): is only the first launch of a burst slow?
run under rocprofv3 --kernel-trace; usage: probe_b2b.py m,n,k,bt,acc,lower reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPX_DBG_NOSYNC"] = "1"
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
m, n, k, bt, acc, low = map(int, sys.argv[1].split(","))
reps = int(sys.argv[2])
rng = np.random.default_rng(1)
A = dev.DeviceMatrix.from_host(ctx, np.tile(rng.standard_normal((1024, k)), (m // 1024, 1)))
B = dev.DeviceMatrix.from_host(ctx, np.tile(rng.standard_normal((1024, k)), (n // 1024, 1)))
C = dev.DeviceMatrix.zeros(ctx, m, n)
ctx.sync()
for burst in range(2):
    for r in range(reps):
        dev.dbg_gemm(ctx, A, B, C, bt, acc, low)
    ctx.sync()
    import time; time.sleep(0.2)
