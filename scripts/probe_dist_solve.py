"""Time of the distributed substitution (dist2_potrs + logdet: alpha and the log-likelihood from the block-cyclic factor) beside
the factorisation, RCCL communicator at world 1 (collectives are no-ops: this is the launch / latency floor of the 128-step
chain, not its communication).  Usage: GPX_COMM=rccl python scripts/probe_dist_solve.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
from gpexp_amd import device as dev, dist
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = dev.Context(0); dev._ctx = ctx
comm = dist.init_from_env(ctx)
rng = np.random.default_rng(N)
Xh = rng.uniform(-1, 1, (N, 8)); yh = np.sin(2 * np.pi * Xh.sum(1) / 8) + 0.3 * rng.standard_normal(N); Zh = rng.uniform(-1, 1, (1024, 8))
spec = dev.KernelSpec(dev.K_MATERN52, 8, [0.5, 1.0])
run = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=int(os.environ.get("GPX_DIST_NB", "512")))
for it in range(3):
    ctx.sync(); t0 = time.perf_counter(); run.fit(); ctx.sync(); t1 = time.perf_counter(); ll, _ = run.solve(); ctx.sync(); t2 = time.perf_counter()
    print("N=%d fit %.2f ms  solve (alpha + logdet) %.2f ms  host issue: factor %.2f solve %.2f  ll=%.12g" %
          (N, 1e3 * (t1 - t0), 1e3 * (t2 - t1), run.host_ms.get("factor", 0), run.host_ms.get("solve", 0), ll), flush=True)
comm.close(); ctx.close()
