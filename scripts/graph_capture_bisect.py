"""Bisects hipGraph capture of the recorded 2-D panel loop (gpexp_amd/dist.py, DistFitIvar2D.use_graph): captures only the first
`ncut` rows of the factorisation program (plus the join rows) of a 1 x 1 replay and launches the graph.  On ROCm 7.2
hipStreamEndCapture segfaults from the first buffer-reuse wait of the panel stream on (n = 4096, nb = 256: row 339), while every
shorter prefix captures and runs -- run it per cut in a fresh process:  python scripts/graph_capture_bisect.py NCUT [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev, dist
from gpexp_amd.dist import OP, Program, MAIN, PANEL, COMM, BACK, EVAL, BULK
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from replay_comm import ReplayComm
ncut = int(sys.argv[1])
ctx = dev.Context(0); dev._ctx = ctx
rng = np.random.default_rng(1)
n, d = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 4
Xh = rng.uniform(-1, 1, (n, d)); yh = rng.standard_normal(n); Zh = rng.uniform(-1, 1, (256, d))
spec = dev.KernelSpec(dev.K_MATERN52, d, [0.5, 1.0])
X = dev.points(ctx, Xh)
Lref = dev.potrf(ctx, dev.kfill(ctx, spec, X, nugget=0.1))
comm = ReplayComm(ctx, 1, 0, Lref)
run = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=256, grid=(1, 1), agg=4, streamed=False, fit_only=True)
run.step(); ctx.sync()
full = run.programs["factor"]
p = Program()
p.rows = [list(r) for r in full.rows[:ncut]]
p.extra = list(full.extra)
p.keep = full.keep
side = (PANEL, COMM, BACK, EVAL, BULK)
for i, s in enumerate(side):
    p.emit(OP["STREAM"], (), (s,)); p.emit(OP["RECORD"], (), (3 + i,))
p.emit(OP["STREAM"], (), (MAIN,))
for i in range(len(side)):
    p.emit(OP["WAIT"], (), (3 + i,))
names = {v: k for k, v in OP.items()}
print("last rows:", [(names[r[0]], r[4]) for r in full.rows[max(0, ncut - 4):ncut]], flush=True)
p.capture(ctx)
p.launch(ctx); ctx.sync()
print("cut %d OK nodes=%d" % (ncut, p.graph_nodes), flush=True)
