"""Debug driver: the 2-D distributed fit with every device primitive followed by a device sync and a NaN scan of the local
matrix and the packed buffers (use with GPX_ALLOC_GUARD=2: library scratch is then NaN-filled, so any read of memory nobody
wrote shows up where it first lands).  torchrun --nproc-per-node W scripts/debug_dist_nan.py N NB, GPX_COMM=host GPX_FORCE_DEVICE=0"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev, dist
N, NB = int(sys.argv[1]), int(sys.argv[2])
ctx = dev.Context(int(os.environ.get("GPX_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))); dev._ctx = ctx
comm = dist.init_from_env(ctx)
rng = np.random.default_rng(N)
d = 4
Xh = rng.uniform(-1, 1, (N, d)); yh = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(N); Zh = rng.uniform(-1, 1, (777, d))
spec = dev.KernelSpec(dev.K_MATERN52, d, [0.5, 1.0])
seen = {}
def wrap(name):
    f = getattr(dist.DeviceOps2D, name)
    def g(self, *a, **kw):
        r = f(self, *a, **kw)
        ctx.sync()
        for i, m in enumerate(a):
            if isinstance(m, dev.DeviceMatrix):
                h = m.to_host()
                if np.isnan(h).any() and (name, i) not in seen:
                    seen[(name, i)] = 1
                    idx = np.argwhere(np.isnan(h))
                    print("rank %d: NaN after %s in argument %d shape %s: %d entries, first at %s, args %s" %
                          (comm.rank, name, i, h.shape, len(idx), idx[0], [x for x in a if not isinstance(x, dev.DeviceMatrix)]), flush=True)
        return r
    setattr(dist.DeviceOps2D, name, g)
if os.environ.get("DEBUG_WRAP", "1") == "1":
    for nm in ("kfill_local", "diag_factor", "panel_trsm", "update", "unpack_rows", "unpack_diag"):
        wrap(nm)
runner = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=NB, streamed=(sys.argv[3] == "1") if len(sys.argv) > 3 else None)
try:
    for rep in range(int(os.environ.get("DEBUG_REPS", "1"))):
        print("rank", comm.rank, "result", runner.step(), flush=True)
except Exception as e:
    print("rank", comm.rank, "raised", repr(e)[:200], flush=True)
comm.barrier(); comm.close(); ctx.close()
