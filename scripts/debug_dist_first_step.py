"""Debug driver: does the FIRST step of a fresh 2-D runner differ from the second (same inputs)?  Prints, per rank, which
block columns of the replicated factor differ.  torchrun --nproc-per-node W ... N NB"""
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
runner = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=NB)
r1 = runner.step(); L1 = runner.L.to_host(tri=1); A1 = runner.A.to_host()
r2 = runner.step(); L2 = runner.L.to_host(tri=1); A2 = runner.A.to_host()
if r1 != r2:
    nblk = (L1.shape[0] + NB - 1) // NB
    badL = [(i, j) for i in range(nblk) for j in range(i + 1) if np.max(np.abs(L1[i*NB:(i+1)*NB, j*NB:(j+1)*NB] - L2[i*NB:(i+1)*NB, j*NB:(j+1)*NB])) > 1e-12]
    la = A1.shape[0] // NB, A1.shape[1] // NB
    badA = [(i, j) for i in range(la[0]) for j in range(la[1]) if np.max(np.abs(A1[i*NB:(i+1)*NB, j*NB:(j+1)*NB] - A2[i*NB:(i+1)*NB, j*NB:(j+1)*NB])) > 1e-12]
    print("MISMATCH rank %d grid %dx%d pr,pc=%d,%d: r1=%s r2=%s\n   replicated-factor blocks (I,J) that differ: %s\n   local blocks that differ: %s" %
          (comm.rank, runner.geo.Pr, runner.geo.Pc, runner.geo.pr, runner.geo.pc, r1, r2, badL[:12], badA[:12]), flush=True)
elif comm.rank == 0:
    print("same", r1, flush=True)
comm.barrier(); comm.close(); ctx.close()
