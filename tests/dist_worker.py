"""Worker for the multi-rank tests; launched as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P \
        tests/dist_worker.py --mode {cpu,gpu} --npts N --blk NB
cpu: the real panel loop (gpexp_amd.dist.dist_potrf) over gloo with a NumPy stand-in for the device primitives
     (test double, built on the oracle's kernel functions) -- checks ownership / panel / broadcast logic.
gpu: the real HIP primitives; ranks share GPU 0 and exchange panels through the host-staged gloo communicator
     (RCCL refuses two ranks on one device); rank 0 compares against the single-GPU path.
Prints "DIST_OK ..." on success from rank 0.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gpexp_amd import dist  # noqa: E402


class NumpyMat:
    def __init__(self, a):
        self.a = a


class NumpyOps:
    """NumPy double of gpexp_amd.dist.DeviceOps (same call signatures, same panel-buffer layout)."""

    def __init__(self, spec_dict):
        from oracle import gpexp_oracle as orc
        self.orc = orc
        self.spec = spec_dict
        self.aux = {}

    def alloc_matrix(self, n):
        np_ = dist.padded(n)
        return NumpyMat(np.full((np_, np_), np.nan))  # NaN: reading a block nobody wrote must show up

    def alloc_panel(self, n, nb):
        return NumpyMat(np.zeros(dist.panel_elems(n, nb)))

    def kfill_owned(self, spec, X, K, nugget, nb, rank, world):
        n = X.shape[0]
        np_ = K.a.shape[0]
        full = np.eye(np_)
        full[:n, :n] = self.orc.cov_matrix(self.spec, X, float(nugget), row_loop=False)
        for j in dist.owned_blocks(n, nb, rank, world):
            c0, c1 = j * nb, min((j + 1) * nb, np_)
            K.a[c0:, c0:c1] = full[c0:, c0:c1]

    def begin(self):
        self._info = 0

    def info(self):
        return self._info

    def stream(self, which):  # the double executes synchronously: streams / events are no-ops
        pass

    def record(self, ev):
        pass

    def wait(self, ev):
        pass

    def panel_factor(self, K, k, nb, P):
        np_ = K.a.shape[0]
        r0 = k * nb
        w = min(nb, np_ - r0)
        rows = np_ - r0
        pan = K.a[r0:, r0:r0 + w].copy()
        assert not np.isnan(np.tril(pan[:w])).any(), "owner factors a panel it never assembled/updated"
        try:
            L11 = np.linalg.cholesky(np.tril(pan[:w]) + np.tril(pan[:w], -1).T)
        except np.linalg.LinAlgError:
            if self._info == 0:
                self._info = r0 + 1
            L11 = np.eye(w)
        pan[:w] = L11
        if rows > w:
            pan[w:] = np.linalg.solve(L11, pan[w:].T).T
        buf = np.zeros((rows, nb))
        buf[:, :w] = pan
        P.a[:rows * nb] = buf.ravel()
        for q in range(w // 128):
            blk = L11[q * 128:(q + 1) * 128, q * 128:(q + 1) * 128]
            P.a[rows * nb + q * 128 * 128: rows * nb + (q + 1) * 128 * 128] = np.linalg.inv(blk).ravel()

    def _panel(self, K, k, nb, P):
        np_ = K.a.shape[0]
        r0 = k * nb
        w = min(nb, np_ - r0)
        rows = np_ - r0
        return np_, r0, w, rows, P.a[:rows * nb].reshape(rows, nb)[:, :w]

    def panel_store(self, K, k, nb, P):
        np_, r0, w, rows, pan = self._panel(K, k, nb, P)
        K.a[r0:, r0:r0 + w] = pan
        self.aux[k] = P.a[rows * nb: rows * nb + (w // 128) * 128 * 128].copy()

    def panel_update(self, K, k, nb, P, j0, j1, rank, world):
        np_, r0, w, rows, pan = self._panel(K, k, nb, P)
        nblk = (np_ + nb - 1) // nb
        for j in range(max(j0, k + 1), min(j1, nblk)):
            if j % world != rank:
                continue
            c0, c1 = j * nb, min((j + 1) * nb, np_)
            K.a[c0:, c0:c1] -= pan[c0 - r0:] @ pan[c0 - r0:c1 - r0].T

    def finish(self, K):
        pass

    # streamed evaluation (same semantics as gpx_dist_ivar_step): one right-looking solve step per stored panel
    def alloc_cross(self, n, m):
        return NumpyMat(np.zeros((dist.padded(n), m)))

    def cross_fill(self, spec, X, Z, B):
        B.a[:] = 0.0
        B.a[:X.shape[0], :] = self.orc.cross_matrix(self.spec, Z, X).T

    def ivar_step(self, K, k, nb, B):
        np_ = K.a.shape[0]
        r0 = k * nb
        w = min(nb, np_ - r0)
        Lkk = np.tril(K.a[r0:r0 + w, r0:r0 + w])
        B.a[r0:r0 + w] = np.linalg.solve(Lkk, B.a[r0:r0 + w])
        if r0 + w < np_:
            B.a[r0 + w:] -= K.a[r0 + w:, r0:r0 + w] @ B.a[r0:r0 + w]

    def variances(self, spec, Z, B, n):
        return self.orc.kernel_diag(self.spec, Z) - np.sum(B.a[:n] ** 2, axis=0)


class NumpyComm:
    def __init__(self):
        self.group = dist._TorchGroup()
        self.rank, self.world = self.group.rank, self.group.world

    def bcast_panel(self, P, count, root):
        buf = np.ascontiguousarray(P.a[:count])
        self.group.bcast_array(buf, root)
        P.a[:count] = buf

    def allgather(self, vec):
        return self.group.allgather(np.atleast_1d(vec))

    def barrier(self):
        self.group.barrier()

    def max_float(self, v):
        return self.group.max_float(v)


def run_cpu(args):
    rng = np.random.default_rng(args.n)
    d = 3
    X = rng.uniform(-1, 1, (args.n, d))
    spec = dict(kind="se", cl=[0.3], signalSize=1.0, d=d)
    comm = NumpyComm()
    ops = NumpyOps(spec)
    K = ops.alloc_matrix(args.n)
    P = [ops.alloc_panel(args.n, args.nb), ops.alloc_panel(args.n, args.nb)]
    ops.kfill_owned(None, X, K, 0.05, args.nb, comm.rank, comm.world)
    info = dist.dist_potrf(ops, comm, K, args.n, args.nb, P)
    assert info == 0
    from oracle import gpexp_oracle as orc
    Lref = np.linalg.cholesky(orc.cov_matrix(spec, X, 0.05, row_loop=False))
    L = np.tril(K.a[:args.n, :args.n])
    err = np.max(np.abs(L - Lref)) / np.max(np.abs(Lref))
    assert err < 1e-12, err
    # every rank must hold the complete factor (that is what lets evaluation shard without moving L)
    allerr = comm.allgather(np.array([err]))[:, 0]
    # streamed evaluation hook: one solve step per stored panel, in panel order, ends with L^-1 K(X, Z_local)
    Z = rng.uniform(-1, 1, (41 + comm.world, d))
    lo, hi = dist.eval_slice(len(Z), comm.rank, comm.world)
    K2 = ops.alloc_matrix(args.n)
    ops.kfill_owned(None, X, K2, 0.05, args.nb, comm.rank, comm.world)
    B = ops.alloc_cross(args.n, hi - lo)
    ops.cross_fill(None, X, Z[lo:hi], B)
    seen = []

    def hook(k):
        seen.append(k)
        ops.ivar_step(K2, k, args.nb, B)

    assert dist.dist_potrf(ops, comm, K2, args.n, args.nb, P, on_stored=hook) == 0
    assert seen == list(range(dist.num_blocks(args.n, args.nb)))
    Wref = np.linalg.solve(Lref, orc.cross_matrix(spec, Z[lo:hi], X).T)
    assert np.max(np.abs(B.a[:args.n] - Wref)) <= 1e-11 * max(np.max(np.abs(Wref)), 1e-300)
    var = ops.variances(None, Z[lo:hi], B, args.n)
    allvar = np.concatenate([v[:c] for v, c in zip(comm.allgather(np.pad(var, (0, 64 - len(var)))), [e - b for b, e in (dist.eval_slice(len(Z), r, comm.world) for r in range(comm.world))])])
    m_ref = orc.fit(spec, X, np.zeros(args.n), 0.05)
    assert np.max(np.abs(allvar - orc.posterior(spec, m_ref, Z, compvar=1)[1])) <= 1e-10
    # evaluation slices tile the index range exactly
    m = 1000 + comm.world
    sl = [dist.eval_slice(m, r, comm.world) for r in range(comm.world)]
    assert sl[0][0] == 0 and sl[-1][1] == m and all(sl[i][1] == sl[i + 1][0] for i in range(comm.world - 1))
    # non-PD detection is agreed on by all ranks
    Kb = ops.alloc_matrix(args.n)
    ops.kfill_owned(None, X, Kb, 0.05, args.nb, comm.rank, comm.world)
    if dist.owner(0, comm.world) == comm.rank:
        Kb.a[5, 5] = -1.0
    info = dist.dist_potrf(ops, comm, Kb, args.n, args.nb, P)
    assert info == 1, info
    t = comm.max_float(float(comm.rank))
    assert t == comm.world - 1
    if comm.rank == 0:
        print("DIST_OK cpu world=%d n=%d nb=%d maxerr=%.2e" % (comm.world, args.n, args.nb, float(allerr.max())), flush=True)


def run_gpu(args):
    from gpexp_amd import device as dev
    ctx = dev.Context(int(os.environ.get("GPX_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dev._ctx = ctx
    comm = dist.init_from_env(ctx)
    rng = np.random.default_rng(args.n)
    d = 4
    N, M = args.n, args.m
    Xh = rng.uniform(-1, 1, (N, d))
    yh = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(N)
    Zh = rng.uniform(-1, 1, (M, d))
    spec = dev.KernelSpec(dev.K_MATERN52, d, [0.5, 1.0])
    runner = dist.DistFitIvar(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=args.nb)
    ll, iv = runner.step()
    ll2, iv2 = runner.step()  # second step re-assembles in place
    assert ll == ll2 and iv == iv2
    # the other evaluation schedule (streamed <-> after the factorisation) must agree
    other = dist.DistFitIvar(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=args.nb, streamed=not runner.streamed)
    ll3, iv3 = other.step()
    assert abs(iv3 - iv) <= 1e-11 * abs(iv), (runner.streamed, iv, iv3)
    if comm.rank == 0:
        assert abs(ll3 - ll) <= 1e-12 * abs(ll), (ll, ll3)
    del other
    Ld = runner.K.to_host(tri=1)
    # single-GPU path on the same inputs (every rank checks its own copy of L)
    X = dev.points(ctx, Xh)
    K1 = dev.kfill(ctx, spec, X, nugget=0.1)
    dev.potrf(ctx, K1)
    L1 = K1.to_host(tri=1)
    errL = float(np.max(np.abs(Ld - L1)) / np.max(np.abs(L1)))
    alpha = dev.potrs(ctx, K1, yh)
    ll1 = -0.5 * float(yh @ alpha) - 0.5 * dev.logdet(ctx, K1) - N / 2.0 * np.log(2 * np.pi)
    iv1 = abs(dev.ivar(ctx, spec, K1, X, dev.points(ctx, Zh)))
    assert errL < 1e-12, errL
    if comm.rank == 0:  # the log-likelihood is produced on rank 0 only (its sweeps hide under rank 0's evaluation slice)
        assert abs(ll - ll1) <= 1e-11 * abs(ll1), (ll, ll1)
    assert abs(iv - iv1) <= 1e-11 * abs(iv1), (iv, iv1)
    # sharded design search: candidates split over the ranks, first-minimum merge == single-GPU selection
    Ch = rng.uniform(-1, 1, (517, d))
    Zd = dev.points(ctx, Zh)
    gidx, gcost = dist.dist_greedy_ivar_step(ctx, comm, spec, runner.K, runner.X, Ch, Zd, 0.1)
    sbest, scosts = dev.greedy_ivar_step(ctx, spec, K1, X, dev.points(ctx, Ch), Zd, 0.1)
    assert gidx == sbest and abs(gcost - scosts[sbest]) <= 1e-12 * abs(gcost), (gidx, sbest, gcost, scosts[sbest])
    # exact tie between two candidates living on different ranks: the lower global index must win
    Ct = Ch.copy()
    Ct[-1] = Ct[sbest]
    if sbest != len(Ct) - 1:
        tidx, _ = dist.dist_greedy_ivar_step(ctx, comm, spec, runner.K, runner.X, Ct, Zd, 0.1)
        assert tidx == sbest, (tidx, sbest)
    errs = comm.allgather(np.array([errL]))[:, 0]
    comm.barrier()
    if comm.rank == 0:
        print("DIST_OK gpu world=%d n=%d nb=%d comm=%s errL=%.2e ll=%.12g ivar=%.12g" %
              (comm.world, N, args.nb, type(comm).__name__, float(errs.max()), ll, iv), flush=True)
    comm.close()
    ctx.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="cpu")
    ap.add_argument("--npts", dest="n", type=int, default=700)
    ap.add_argument("--mpts", dest="m", type=int, default=333)
    ap.add_argument("--blk", dest="nb", type=int, default=256)
    a = ap.parse_args()
    (run_cpu if a.mode == "cpu" else run_gpu)(a)
