"""Multi-GPU GP fit and evaluation: one process per GPU, RCCL over xGMI.

This replaces the reference's only parallel backend -- `parallelizeMcForLoop`, a fork + `mp.Queue` helper that
slices the evaluation points row-wise over CPU processes (parallel_utilities.py:26-80, used at gp.py:258) -- by
two shardings:

  fit   the covariance matrix is distributed by block columns (width `nb`, owner = block index mod world).
        Assembly is communication-free; the right-looking Cholesky broadcasts one factored panel per step
        (`gpx_comm_bcast` -> ncclBroadcast) and every rank updates the block columns it owns.  Every rank also
        KEEPS each received panel, so at the end each GPU holds the complete factor L.
  eval  posterior / IVAR evaluation points are split in contiguous slices [r*M/W, (r+1)*M/W), exactly the
        chunking the reference's helper intends (parallel_utilities.py:46-60); each rank solves against its own
        copy of L and the only exchange is an all-gather of one partial sum per rank (summed in rank order, so
        the result does not depend on arrival order).

The panel loop lives here, in Python, on top of three C-ABI primitives (gpx_dist_panel_factor /
gpx_comm_bcast / gpx_dist_panel_store + gpx_dist_panel_update); per step that is a handful of ctypes calls against ~10^10 flops of GPU
work.  The communicator is an object with `bcast_panel`, `allgather`, `barrier`, `max_float`:
  RcclComm      device-to-device over RCCL (the product path; no torch in the process, see FileRendezvous);
  HostStagedComm  device -> host -> torch.distributed(gloo) -> device, for the world_size>1 tests that share one
                GPU (RCCL refuses two ranks on one device) -- test infrastructure, never used by bench.py.
The index arithmetic (ownership, slices, panel sizes) is plain Python and is unit-tested on CPU with gloo.
"""
import ctypes as C
import os

import numpy as np

from . import device as _dev
from ._lib import check, dptr, as_f64, c_i64

TILE = 128


# ---- pure index logic (CPU-testable) ---------------------------------------------------------------------
def padded(n):
    return (max(n, 1) + TILE - 1) // TILE * TILE


def num_blocks(n, nb):
    return (padded(n) + nb - 1) // nb


def owner(j, world):
    """Rank that owns block column j."""
    return j % world


def owned_blocks(n, nb, rank, world):
    return [j for j in range(num_blocks(n, nb)) if owner(j, world) == rank]


def eval_slice(m, rank, world):
    """Contiguous slice of the M evaluation points handled by `rank` (balanced to within one point)."""
    base, rem = divmod(m, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def panel_elems(n, nb):
    np_ = padded(n)
    return np_ * nb + (nb // TILE) * TILE * TILE


def ordered_sum(parts):
    """Sum partial results in rank order (deterministic irrespective of arrival order)."""
    s = 0.0
    for p in parts:
        s += float(p)
    return s


def merge_argmin(values, indices):
    """First-minimum rule across ranks: lowest value, ties -> lowest global index (np.argmin semantics)."""
    best = None
    for v, i in zip(values, indices):
        if best is None or v < best[0] or (v == best[0] and i < best[1]):
            best = (float(v), int(i))
    return best


# ---- communicators -----------------------------------------------------------------------------------------
class _TorchGroup:
    """Thin wrapper over torch.distributed (gloo, CPU tensors) for rendezvous-level exchanges."""

    def __init__(self):
        import torch
        import torch.distributed as td
        self.torch = torch
        self.td = td
        if not td.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            td.init_process_group(backend="gloo")
        self.rank = td.get_rank()
        self.world = td.get_world_size()

    def bcast_bytes(self, b, root=0):
        t = self.torch.zeros(len(b), dtype=self.torch.uint8)
        if self.rank == root:
            t = self.torch.tensor(list(b), dtype=self.torch.uint8)
        self.td.broadcast(t, src=root)
        return bytes(t.tolist())

    def bcast_array(self, a, root):
        t = self.torch.from_numpy(a)
        self.td.broadcast(t, src=root)
        return a

    def make_grid(self, Pr, Pc):
        """Process-row / process-column groups (every rank creates every group, in the same order)."""
        self.rows = [self.td.new_group([p * Pc + q for q in range(Pc)]) for p in range(Pr)]
        self.cols = [self.td.new_group([p * Pc + q for p in range(Pr)]) for q in range(Pc)]
        self.Pr, self.Pc = Pr, Pc

    def _group(self, grp):
        """(process group, world ranks of its members) for WORLD / ROW / COL of this rank."""
        pr, pc = self.rank // self.Pc, self.rank % self.Pc
        if grp == 1:
            return self.rows[pr], [pr * self.Pc + q for q in range(self.Pc)]
        if grp == 2:
            return self.cols[pc], [p * self.Pc + pc for p in range(self.Pr)]
        return None, list(range(self.world))

    def bcast_array_grp(self, a, root, grp):
        g, members = self._group(grp)
        self.td.broadcast(self.torch.from_numpy(a), src=members[root], group=g)
        return a

    def reduce_array_grp(self, a, root, grp):
        g, members = self._group(grp)
        t = self.torch.from_numpy(a.copy())
        self.td.reduce(t, dst=members[root], op=self.td.ReduceOp.SUM, group=g)
        if self.rank == members[root]:
            a[:] = t.numpy()
        return a

    def allreduce_array(self, a):
        self.td.all_reduce(self.torch.from_numpy(a), op=self.td.ReduceOp.SUM)
        return a

    def allgather(self, vec):
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        outs = [self.torch.zeros(vec.size, dtype=self.torch.float64) for _ in range(self.world)]
        self.td.all_gather(outs, self.torch.from_numpy(vec.copy()))
        return np.stack([o.numpy() for o in outs])

    def barrier(self):
        self.td.barrier()

    def max_float(self, v):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t[0])


class FileRendezvous:
    """Single-node exchange of the 128-byte ncclUniqueId without any framework in the process: rank 0 writes it
    atomically to a file keyed by the launcher's (MASTER_PORT, run id, launcher pid); the others poll for it.
    (Importing torch next to the system RCCL loads a second, uninitialised HSA runtime into the process and
    ncclCommInitRank then fails with "no ROCm-capable device", so the RCCL path stays torch-free.)"""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        key = "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                            os.getppid())
        self.path = os.path.join(os.environ.get("GPX_RDV_DIR", "/tmp"), "gpx_rdv_" + key)

    def exchange(self, payload):
        import time
        if self.rank == 0:
            tmp = self.path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(payload)
            os.replace(tmp, self.path)
            return payload
        t0 = time.time()
        while not os.path.exists(self.path):
            if time.time() - t0 > 300:
                raise RuntimeError("rendezvous file %s did not appear" % self.path)
            time.sleep(0.01)
        with open(self.path, "rb") as f:
            return f.read()

    def cleanup(self):
        if self.rank == 0:
            try:
                os.remove(self.path)
            except OSError:
                pass


class RcclComm:
    """Device-side collectives over RCCL; barriers and the timing max are RCCL all-gathers of one double."""

    def __init__(self, ctx, rendezvous=None):
        self.ctx = ctx
        rdv = rendezvous or FileRendezvous()
        self.rank, self.world = rdv.rank, rdv.world
        uid = C.create_string_buffer(128)
        if self.rank == 0:
            check(ctx.lib.gpx_comm_unique_id(uid))
        raw = rdv.exchange(uid.raw)
        check(ctx.lib.gpx_comm_init(ctx.h, self.rank, self.world, C.create_string_buffer(raw, 128)))
        self.barrier()
        rdv.cleanup()

    def bcast_panel(self, P, count, root):
        check(self.ctx.lib.gpx_comm_bcast(self.ctx.h, P.h, int(count), int(root)))

    # ---- 2-D path: process grid, sub-communicators (ncclCommSplit), all-link panel broadcast ----
    def set_grid(self, Pr, Pc):
        if getattr(self, "grid", None) != (Pr, Pc):
            check(self.ctx.lib.gpx_comm_grid(self.ctx.h, int(Pr), int(Pc)))
            self.grid = (Pr, Pc)

    def bcast_grp(self, buf, offset, count, root, grp):
        check(self.ctx.lib.gpx_comm_bcast_grp(self.ctx.h, buf.h, int(offset), int(count), int(root), int(grp)))

    def reduce_grp(self, buf, offset, count, root, grp):
        check(self.ctx.lib.gpx_comm_reduce_grp(self.ctx.h, buf.h, int(offset), int(count), int(root), int(grp)))

    def allreduce(self, buf, offset, count):
        check(self.ctx.lib.gpx_comm_allreduce(self.ctx.h, buf.h, int(offset), int(count)))

    def allreduce_host(self, vec):
        vec = as_f64(np.atleast_1d(vec)).copy()
        check(self.ctx.lib.gpx_comm_allreduce_host(self.ctx.h, dptr(vec), vec.size))
        return vec

    def panel_bcast(self, buf, pieces):
        n = len(pieces)
        offs = (c_i64 * n)(*[int(p[0]) for p in pieces])
        cnts = (c_i64 * n)(*[int(p[1]) for p in pieces])
        roots = (C.c_int * n)(*[int(p[2]) for p in pieces])
        check(self.ctx.lib.gpx_comm_panel_bcast(self.ctx.h, buf.h, offs, cnts, roots, n))

    def allgather(self, vec):
        vec = as_f64(np.atleast_1d(vec))
        out = np.empty((self.world, vec.size))
        check(self.ctx.lib.gpx_comm_allgather_host(self.ctx.h, dptr(vec), vec.size, dptr(out)))
        return out

    def barrier(self):
        self.ctx.sync()
        self.allgather(np.array([0.0]))

    def max_float(self, v):
        return float(np.max(self.allgather(np.array([float(v)]))))

    def close(self):
        self.ctx.lib.gpx_comm_destroy(self.ctx.h)


class HostStagedComm:
    """Test-only communicator: panels bounce through host memory and gloo (several ranks may share one GPU)."""

    def __init__(self, ctx, group=None):
        self.ctx = ctx
        self.group = group or _TorchGroup()
        self.rank, self.world = self.group.rank, self.group.world

    def bcast_panel(self, P, count, root):
        buf = np.empty(int(count))
        if self.rank == root:
            check(self.ctx.lib.gpx_mat_read(self.ctx.h, P.h, 0, int(count), dptr(buf)))
        self.group.bcast_array(buf, root)
        if self.rank != root:
            check(self.ctx.lib.gpx_mat_write(self.ctx.h, P.h, 0, int(count), dptr(buf)))

    # ---- 2-D path (same interface as RcclComm) ----
    def set_grid(self, Pr, Pc):
        if getattr(self, "grid", None) != (Pr, Pc):
            self.group.make_grid(Pr, Pc)
            self.grid = (Pr, Pc)

    def _read(self, buf, offset, count):
        a = np.empty(int(count))
        check(self.ctx.lib.gpx_mat_read(self.ctx.h, buf.h, int(offset), int(count), dptr(a)))
        return a

    def _write(self, buf, offset, a):
        check(self.ctx.lib.gpx_mat_write(self.ctx.h, buf.h, int(offset), a.size, dptr(a)))

    def bcast_grp(self, buf, offset, count, root, grp):
        if count == 0:
            return
        self._write(buf, offset, self.group.bcast_array_grp(self._read(buf, offset, count), root, grp))

    def reduce_grp(self, buf, offset, count, root, grp):
        if count == 0:
            return
        self._write(buf, offset, self.group.reduce_array_grp(self._read(buf, offset, count), root, grp))

    def allreduce(self, buf, offset, count):
        self._write(buf, offset, self.group.allreduce_array(self._read(buf, offset, count)))

    def allreduce_host(self, vec):
        return self.group.allreduce_array(np.array(np.atleast_1d(vec), dtype=np.float64))

    def panel_bcast(self, buf, pieces):
        for off, cnt, root in pieces:
            a = self._read(buf, off, cnt)
            self.group.bcast_array(a, root)
            if self.rank != root:
                self._write(buf, off, a)

    def allgather(self, vec):
        return self.group.allgather(np.atleast_1d(vec))

    def barrier(self):
        self.ctx.sync()
        self.group.barrier()

    def max_float(self, v):
        return self.group.max_float(v)

    def close(self):
        pass


def init_from_env(ctx):
    """Communicator for the process group the launcher created (RANK / WORLD_SIZE / MASTER_* in the environment)."""
    kind = os.environ.get("GPX_COMM", "rccl")
    return HostStagedComm(ctx) if kind == "host" else RcclComm(ctx)


# ---- distributed operations -----------------------------------------------------------------------------------
MAIN, PANEL, COMM, BACK = 0, 1, 2, 3  # stream indices of the context (gpx_stream_select); BACK = CU-masked background


class DeviceOps:
    """The device primitives the panel loop needs (the tests substitute a NumPy double for CPU runs)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def alloc_matrix(self, n):
        return _dev.DeviceMatrix.zeros(self.ctx, n, n)

    def points(self, x):
        return _dev.points(self.ctx, x)

    def posterior_var(self, spec, L, X, Z):
        return _dev.posterior(self.ctx, spec, L, X, None, Z, want_mean=False)[1]

    def alloc_panel(self, n, nb):
        return _dev.DeviceMatrix.zeros(self.ctx, panel_elems(n, nb), 1, pad=False)

    def kfill_owned(self, spec, X, K, nugget, nb, rank, world):
        n = X.shape[0]
        nug, nlen = _dev._nugget_args(nugget, n)
        check(self.ctx.lib.gpx_dist_kfill(self.ctx.h, *spec.args(), X.h, dptr(nug), nlen, K.h, int(nb), int(rank),
                                          int(world)))

    def begin(self):
        check(self.ctx.lib.gpx_dist_begin(self.ctx.h))

    def info(self):
        v = C.c_int(0)
        check(self.ctx.lib.gpx_dist_info(self.ctx.h, C.byref(v)))
        return v.value

    def panel_factor(self, K, k, nb, P):
        check(self.ctx.lib.gpx_dist_panel_factor(self.ctx.h, K.h, int(k), int(nb), P.h))

    def panel_store(self, K, k, nb, P):
        check(self.ctx.lib.gpx_dist_panel_store(self.ctx.h, K.h, int(k), int(nb), P.h))

    def panel_update(self, K, k, nb, P, j0, j1, rank, world):
        check(self.ctx.lib.gpx_dist_panel_update(self.ctx.h, K.h, int(k), int(nb), P.h, int(j0), int(j1), int(rank),
                                                 int(world)))

    def finish(self, K):
        check(self.ctx.lib.gpx_dist_finish(self.ctx.h, K.h))

    # streamed evaluation (one right-looking solve step per stored panel)
    def alloc_cross(self, n, m):
        return _dev.DeviceMatrix.zeros(self.ctx, n, m)

    def cross_fill(self, spec, X, Z, B):
        _dev.kfill_into(self.ctx, spec, X, B, Z=Z)

    def ivar_step(self, K, k, nb, B):
        check(self.ctx.lib.gpx_dist_ivar_step(self.ctx.h, K.h, int(k), int(nb), B.h))

    def variances(self, spec, Z, B, n):
        """k(z,z) - column sums of squares of the solved cross matrix (signed, as evaluateVariance)."""
        ss = np.empty(B.shape[1])
        check(self.ctx.lib.gpx_col_sumsq(self.ctx.h, B.h, int(n), dptr(ss)))
        return _dev.kdiag(self.ctx, spec, Z) - ss

    # stream / event plumbing of the look-ahead pipeline
    def stream(self, which):
        check(self.ctx.lib.gpx_stream_select(self.ctx.h, int(which)))

    def record(self, ev):
        check(self.ctx.lib.gpx_event_record(self.ctx.h, int(ev)))

    def wait(self, ev):
        check(self.ctx.lib.gpx_event_wait(self.ctx.h, int(ev)))


# event ids of the pipeline (per step k, 5 kinds)
def _ev(kind, k):
    return 5 * (k + 1) + kind


EV_COLREADY, EV_FACT, EV_BCAST, EV_APPLIED, EV_STORED = 0, 1, 2, 3, 4


def dist_potrf(ops, comm, K, n, nb, panels, on_stored=None):
    """Right-looking block-column Cholesky of the distributed matrix K (in place) with one step of LOOK-AHEAD; every
    rank ends with all of L.  `panels` = two packed panel buffers used alternately.

    Three streams per rank.  For step k:
      PANEL  (owner of k)  waits until block column k has received update k-1, packs + factors it      -> EV_FACT[k]
      COMM   (all)         ncclBroadcast of the packed panel from its owner                              -> EV_BCAST[k]
      MAIN   (all)         stores the panel, updates the owned column k+1 FIRST (-> EV_COLREADY[k+1], which releases
                           the owner's PANEL stream for step k+1) and then the remaining owned columns   -> EV_APPLIED[k]
    so the factorisation and broadcast of panel k+1 run underneath the bulk of update k.  A buffer is reused at step
    k+2 only after EV_APPLIED[k].  Everything is enqueued asynchronously in step order, hence every rank issues its
    collectives in the same order.  Returns 0 or the 1-based index of the first non-positive pivot (agreed by all).

    `on_stored(k)` (optional) is called right after panel k has been stored (event EV_STORED[k] recorded on MAIN): the
    hook of the streamed evaluation, which enqueues its step k on the BACK stream behind that event.
    """
    nblk = num_blocks(n, nb)
    np_ = padded(n)
    ops.stream(MAIN)
    ops.begin()
    ops.record(_ev(EV_COLREADY, 0))  # column 0 is ready once the assembly (queued on MAIN) is done
    for k in range(nblk):
        root = owner(k, comm.world)
        P = panels[k & 1]
        rows = np_ - k * nb
        count = rows * nb + (nb // TILE) * TILE * TILE
        if comm.rank == root:
            ops.stream(PANEL)
            ops.wait(_ev(EV_COLREADY, k))
            if k >= 2:
                ops.wait(_ev(EV_APPLIED, k - 2))
            ops.panel_factor(K, k, nb, P)
            ops.record(_ev(EV_FACT, k))
        ops.stream(COMM)
        if comm.rank == root:
            ops.wait(_ev(EV_FACT, k))
        if k >= 2:
            ops.wait(_ev(EV_APPLIED, k - 2))
        comm.bcast_panel(P, count, root)
        ops.record(_ev(EV_BCAST, k))
        ops.stream(MAIN)
        ops.wait(_ev(EV_BCAST, k))
        ops.panel_store(K, k, nb, P)
        if on_stored is not None:
            ops.record(_ev(EV_STORED, k))
            on_stored(k)
            ops.stream(MAIN)
        if k + 1 < nblk:
            ops.panel_update(K, k, nb, P, k + 1, k + 2, comm.rank, comm.world)
            ops.record(_ev(EV_COLREADY, k + 1))
            ops.panel_update(K, k, nb, P, k + 2, nblk, comm.rank, comm.world)
        ops.record(_ev(EV_APPLIED, k))
    ops.stream(MAIN)
    info = ops.info()  # synchronises every stream
    ops.finish(K)
    allinfo = comm.allgather(np.array([float(info)]))[:, 0]
    bad = [int(v) for v in allinfo if v > 0]
    return min(bad) if bad else 0


class DistFitIvar:
    """bench.py's multi-GPU step: distributed fit (kfill + potrf), alpha/logdet/log-likelihood, sharded IVAR."""

    def __init__(self, ctx, comm, spec, Xh, yh, Zh, noise, nb=512, ops=None, streamed=None):
        self.ctx, self.comm, self.spec = ctx, comm, spec
        self.ops = ops or DeviceOps(ctx)
        # Streamed evaluation (default from 4 ranks; GPX_DIST_STREAM_IVAR=0/1 overrides): the rank's slice of the IVAR
        # solve advances by one right-looking step per arrived panel on the background stream, underneath the
        # broadcast-bound panel chain, instead of starting after the factorisation.  It pays where the panel chain leaves
        # the GPU idle: with 2 ranks each GPU still carries half of the trailing update and half of the solve (GPU-bound),
        # and the streamed form's K=512 updates on 224 CUs would cost more than the idle time they fill.
        env = os.environ.get("GPX_DIST_STREAM_IVAR")
        self.streamed = (comm.world >= 4) if streamed is None else bool(streamed)
        if env is not None:
            self.streamed = env == "1"
        self.n, self.noise, self.nb = Xh.shape[0], float(noise), int(nb)
        self.yh = np.ascontiguousarray(yh, dtype=np.float64)
        self.m = Zh.shape[0]
        self.X = _dev.points(ctx, Xh)
        lo, hi = eval_slice(self.m, comm.rank, comm.world)
        self.Zloc = _dev.points(ctx, Zh[lo:hi]) if hi > lo else None
        self.K = self.ops.alloc_matrix(self.n)
        self.P = [self.ops.alloc_panel(self.n, self.nb), self.ops.alloc_panel(self.n, self.nb)]
        self.y_dev = _dev.padded_vector(ctx, self.yh)
        self.alpha_dev = _dev.padded_vector(ctx, np.zeros(self.n))
        self.B = self.ops.alloc_cross(self.n, hi - lo) if (self.streamed and hi > lo) else None

    def step(self):
        ctx, comm = self.ctx, self.comm
        self.ops.kfill_owned(self.spec, self.X, self.K, self.noise, self.nb, comm.rank, comm.world)
        hook = None
        if self.B is not None:
            ops = self.ops
            ops.stream(BACK)
            ops.cross_fill(self.spec, self.X, self.Zloc, self.B)   # independent of the factorisation
            ops.stream(MAIN)

            def hook(k):
                ops.stream(BACK)
                ops.wait(_ev(EV_STORED, k))
                ops.ivar_step(self.K, k, self.nb, self.B)

        info = dist_potrf(self.ops, comm, self.K, self.n, self.nb, self.P, on_stored=hook)
        if info:
            from ._lib import NotPositiveDefinite
            raise NotPositiveDefinite(info)
        # every rank holds L: alpha / logdet need no exchange.  The sweeps are latency-bound micro-launches: rank 0 runs
        # them on the side stream underneath its slice of the evaluation GEMMs; the other ranks skip them.
        if comm.rank == 0:
            ctx.stream(PANEL)
            _dev.potrs_dev(ctx, self.K, self.y_dev, self.alpha_dev)
            ctx.stream(MAIN)
        part = 0.0
        if self.B is not None:     # the solve finished with the last panel (dist_potrf synchronised every stream)
            part = float(np.sum(self.ops.variances(self.spec, self.Zloc, self.B, self.n)))
        elif self.Zloc is not None:
            _, var = _dev.posterior(ctx, self.spec, self.K, self.X, None, self.Zloc, want_mean=False)
            part = float(np.sum(var))
        ll = 0.0
        if comm.rank == 0:
            logdet = _dev.logdet(ctx, self.K)
            ctx.sync()
            alpha = self.alpha_dev.to_host()[:self.n, 0]
            ll = -0.5 * float(self.yh @ alpha) - 0.5 * logdet - self.n / 2.0 * np.log(2 * np.pi)
        iv = abs(ordered_sum(comm.allgather(np.array([part]))[:, 0]) / self.m)
        return ll, iv


def dist_greedy_ivar_step(ctx, comm, spec, L, X, cand_host, Z, noise):
    """Greedy-IVAR step with the CANDIDATES sharded over the ranks: every rank already holds the
    complete factor, scores its contiguous slice of candidates with gpx_greedy_ivar_step, and the ranks exchange one
    (cost, global index) pair each; the winner follows np.argmin's rule -- lowest cost, ties to the lowest global index
    -- so the selection is identical to the single-GPU one.  Returns (global best index, its cost)."""
    m = cand_host.shape[0]
    lo, hi = eval_slice(m, comm.rank, comm.world)
    if hi > lo:
        best, costs = _dev.greedy_ivar_step(ctx, spec, L, X, _dev.points(ctx, cand_host[lo:hi]), Z, noise)
        mine = np.array([costs[best], float(lo + best)])
    else:
        mine = np.array([np.inf, float(m)])
    pairs = comm.allgather(mine)
    cost, idx = merge_argmin(pairs[:, 0], pairs[:, 1].astype(np.int64))
    return idx, cost


def dist_greedy_var(ctx, comm, spec, cand_host, nsel, keep=()):
    """Greedy maximum-variance design is O(M*n) per step and sequential in the steps: every rank runs the identical
    deterministic selection on the full candidate set (no exchange) -- "replicas" for this sub-path, by design."""
    return _dev.greedy_var(ctx, spec, _dev.points(ctx, cand_host), nsel, keep=keep)


# =====================================================================================================================
# 2-D block-cyclic fit (north_star) -- the default multi-GPU path
# =====================================================================================================================
# Process grid Pr x Pc (8 -> 2 x 4, 4 -> 2 x 2, 2 -> 1 x 2), rank (pr, pc) = (rank // Pc, rank % Pc).  Global block (I, J)
# of the padded matrix lives on rank (I % Pr, J % Pc) at local block (I // Pr, J // Pc): every rank allocates only its
# ~N^2 / (Pr Pc) share of the working matrix.  Right-looking step k:
#   owner (k % Pr, k % Pc)   factors the diagonal block            gpx_dist2_diag_factor            PANEL stream
#   process column k % Pc    receives it                            ncclBroadcast on the column sub-communicator
#                            solves its rows of the panel           gpx_dist2_panel_trsm             PANEL stream
#   all ranks                receive every piece of the panel       gpx_comm_panel_bcast: scatter + all-gather over grouped
#                                                                   ncclSend/ncclRecv, every xGMI link carries 2/(W-1) of it
#                            update the local block columns         gpx_dist2_update (column k+1 first: look-ahead)
#                            keep the panel in the replicated L     gpx_dist2_unpack_*               BACK stream
# The replicated copy of L (finished panels only) is what lets the evaluation phase shard the evaluation points with no
# exchange (and stream, see DistFitIvar); alpha / logdet come from the block-cyclic factor by distributed forward / back
# substitution (reduce along process rows, broadcast down process columns) and ncclAllReduce.
WORLD, ROW, COL = 0, 1, 2   # communicator groups (gpx_comm_grid)


def choose_grid(world):
    """Pr x Pc with Pr <= Pc, Pr the largest divisor of `world` not above sqrt(world): 8 -> 2x4, 4 -> 2x2, 2 -> 1x2."""
    pr = 1
    for c in range(1, int(world ** 0.5) + 1):
        if world % c == 0:
            pr = c
    return pr, world // pr


class Grid2D:
    """Pure index logic of the 2-D block-cyclic layout (CPU-testable)."""

    def __init__(self, n, nb, Pr, Pc, rank):
        assert nb % TILE == 0 and 0 <= rank < Pr * Pc
        self.n, self.nb, self.Pr, self.Pc, self.rank = int(n), int(nb), int(Pr), int(Pc), int(rank)
        self.pr, self.pc = rank // Pc, rank % Pc
        self.np = padded(n)
        self.nblk = num_blocks(n, nb)
        self.dsz = nb * nb + (nb // TILE) * TILE * TILE
        self.piece_stride = self.dsz + max(self.local_rows(p) for p in range(Pr)) * nb

    def height(self, I):
        return min(self.nb, self.np - I * self.nb)

    def owner_rank(self, I, J):
        return (I % self.Pr) * self.Pc + (J % self.Pc)

    def blocks_before(self, p, P, I):
        """Number of blocks I' < I with I' % P == p."""
        return 0 if I <= p else (I - 1 - p) // P + 1

    def local_rows(self, p):
        return sum(self.height(I) for I in range(p, self.nblk, self.Pr))

    def local_cols(self, q):
        return sum(self.height(J) for J in range(q, self.nblk, self.Pc))

    def li0(self, p, k):
        """First local block row of process row p whose global block index exceeds k."""
        return self.blocks_before(p, self.Pr, k + 1)

    def row_off(self, p, li):
        """Local row offset of local block row li on process row p (only the matrix's last block can be short, so the
        offset one past the last local block is the local row count, not li * nb)."""
        return min(li * self.nb, self.local_rows(p))

    def piece_rows(self, p, k):
        """Rows of panel k held by process row p: the blocks I > k with I % Pr == p."""
        return max(self.local_rows(p) - self.li0(p, k) * self.nb, 0)

    def piece_off(self, p):
        return p * self.piece_stride

    def buf_elems(self):
        return self.Pr * self.piece_stride

    def pieces(self, k):
        """(offset, count, world root) of every region the panel broadcast of step k delivers."""
        kr, kc = k % self.Pr, k % self.Pc
        out = []
        for p in range(self.Pr):
            m = self.piece_rows(p, k)
            if p == kr:
                out.append((self.piece_off(p), self.dsz + m * self.nb, p * self.Pc + kc))
            elif m > 0:
                out.append((self.piece_off(p) + self.dsz, m * self.nb, p * self.Pc + kc))
        return out

    def my_cols_after(self, k):
        """Global block columns J > k owned by this rank's process column."""
        return [J for J in range(self.pc, self.nblk, self.Pc) if J > k]

    def update_args(self, k, J, below_diag=False):
        """(lr0, m, lc0, n, aoff, boff) of the local trailing update of block column J > k by panel k (this rank);
        below_diag: only the block rows I > J (the diagonal block of column J is updated by the early path)."""
        nb, pr = self.nb, self.pr
        liJ = self.blocks_before(pr, self.Pr, J + (1 if below_diag else 0))   # first local block row with I >= J (> J)
        m = max(self.local_rows(pr) - liJ * nb, 0)
        if m == 0:
            liJ = self.local_rows(pr) // nb                    # nothing below: stay inside the local matrix
        aoff = self.piece_off(pr) + self.dsz + (liJ - self.li0(pr, k)) * nb * nb
        pj = J % self.Pr
        boff = self.piece_off(pj) + self.dsz + (J // self.Pr - self.li0(pj, k)) * nb * nb
        return self.row_off(pr, liJ), m, (J // self.Pc) * nb, self.height(J), aoff, boff


class DeviceOps2D(DeviceOps):
    """Device primitives of the 2-D panel loop (gpx_dist2_*); the tests substitute a NumPy double."""

    def alloc_local(self, geo):
        return _dev.DeviceMatrix.zeros(self.ctx, max(geo.local_rows(geo.pr), 1), max(geo.local_cols(geo.pc), 1))

    def alloc_buf(self, geo):
        return _dev.DeviceMatrix.zeros(self.ctx, geo.buf_elems(), 1, pad=False)

    def alloc_vec(self, n):
        return _dev.DeviceMatrix.zeros(self.ctx, max(int(n), 1), 1, pad=False)

    def kfill_local(self, spec, X, A, nugget, geo):
        nug, nlen = _dev._nugget_args(nugget, X.shape[0])
        # tests: hold the assembly of ONE rank back ("ms@rank"), so that a missing dependency on it shows every time
        spec_delay = os.environ.get("GPX_TEST_DELAY_FILL", "")
        delay = int(spec_delay.split("@")[0]) if spec_delay and int(spec_delay.split("@")[1]) == geo.rank else 0
        if delay:
            check(self.ctx.lib.gpx_dbg_spin(self.ctx.h, delay))
        check(self.ctx.lib.gpx_dist2_kfill(self.ctx.h, *spec.args(), X.h, dptr(nug), nlen, A.h, geo.nb, geo.Pr, geo.Pc,
                                           geo.pr, geo.pc))

    def diag_factor(self, A, lr, lc, w, G, doff, nb, base, n_valid):
        check(self.ctx.lib.gpx_dist2_diag_factor(self.ctx.h, A.h, lr, lc, w, G.h, doff, nb, base, n_valid))

    def panel_trsm(self, A, lr0, m, lc, w, G, doff, roff, nb):
        check(self.ctx.lib.gpx_dist2_panel_trsm(self.ctx.h, A.h, lr0, m, lc, w, G.h, doff, roff, nb))

    def update(self, A, lr0, m, lc0, n, G, aoff, boff, w, nb):
        check(self.ctx.lib.gpx_dist2_update(self.ctx.h, A.h, lr0, m, lc0, n, G.h, aoff, boff, w, nb))

    def unpack_rows(self, G, roff, m, w, nb, L, first_block, stride, col0):
        check(self.ctx.lib.gpx_dist2_unpack_rows(self.ctx.h, G.h, roff, m, w, nb, L.h, first_block, stride, col0))

    def unpack_diag(self, G, doff, w, nb, L, r0):
        check(self.ctx.lib.gpx_dist2_unpack_diag(self.ctx.h, G.h, doff, w, nb, L.h, r0))

    def trsv_diag(self, A, lr, lc, w, v, voff, transposed):
        check(self.ctx.lib.gpx_dist2_trsv_diag(self.ctx.h, A.h, lr, lc, w, v.h, voff, int(transposed)))

    def gemv(self, A, lr0, m, lc, w, x, xoff, acc, aoff, transposed):
        check(self.ctx.lib.gpx_dist2_gemv(self.ctx.h, A.h, lr0, m, lc, w, x.h, xoff, acc.h, aoff, int(transposed)))

    def logdet_acc(self, A, lr, lc, w, n_valid, acc):
        check(self.ctx.lib.gpx_dist2_logdet_acc(self.ctx.h, A.h, lr, lc, w, n_valid, acc.h))

    def vec_op(self, dst, doff, src, soff, n, mode):
        check(self.ctx.lib.gpx_vec_op(self.ctx.h, dst.h, doff, src.h if src is not None else None, soff, n, mode))

    def vec_to_host(self, v, n):
        return v.to_host()[:n, 0]

    def vec_from_host(self, v, a):
        a = as_f64(np.ravel(a))
        check(self.ctx.lib.gpx_mat_write(self.ctx.h, v.h, 0, a.size, dptr(a)))


# event kinds of the 2-D pipeline (per step k)
(E_COLREADY, E_DFACT, E_DBC, E_PIECE, E_ARRIVED, E_STORED, E_UPD, E_DIAGREADY, E_EARLYSOLVED, E_EARLY, E_COL2,
 E_PANELDONE) = range(12)


def _ev2(kind, k):
    return 12 * (k + 1) + kind


def dist2_potrf(ops, comm, geo, A, G, L=None, on_stored=None):
    """2-D block-cyclic right-looking Cholesky of the distributed matrix A (in place: A ends as the block-cyclic factor)
    with look-ahead and a CRITICAL-PATH-FIRST diagonal chain.  G = two packed panel buffers (geo.buf_elems() doubles) used
    alternately; L (optional) = full-size matrix that receives every finished panel (replicated factor for the evaluation
    phase).

    What serialises a distributed factorisation is the chain diag(k) -> panel(k) -> update of column k+1 -> diag(k+1).  Only
    ONE nb x nb block of panel k enters diag(k+1): L[k+1, k].  So, per step k,
      diagonal chain   owner of (k,k) factors it; the holder of block row k+1 solves THAT block first and sends it along its
                       process row (one small ncclBroadcast on the row sub-communicator); the owner of (k+1,k+1) applies it
                       to the diagonal block (which got the contributions of the panels before k from the two-ahead column
                       update below) and can factor at once: potrf(nb) + two small hops per step;
      panel chain      meanwhile the rest of panel k is solved, every piece goes to every rank over all links
                       (gpx_comm_panel_bcast), column k+1 is updated below its diagonal block (releases panel k+1's solve);
      bulk             column k+2 next (its diagonal block is the one the chain needs in two steps), then the other local
                       block columns.
    Streams per rank: PANEL (diagonal factor, solves, the early diagonal update), COMM (collectives, enqueued in the same
    order on every rank of every communicator), MAIN (trailing updates), BACK (copies into L, then the streamed-evaluation
    hook `on_stored(k)`).  A panel buffer is rewritten at step k+2 only after everything of step k that reads it is done.
    Returns 0 or the 1-based index of the first non-positive pivot (agreed by all)."""
    nb, Pr, Pc, pr, pc = geo.nb, geo.Pr, geo.Pc, geo.pr, geo.pc
    nblk = geo.nblk
    ops.stream(MAIN)
    ops.begin()
    ops.record(_ev2(E_DIAGREADY, 0))  # the assembly was queued on MAIN
    ops.record(_ev2(E_COLREADY, 0))
    for k in range(nblk):
        kr, kc = k % Pr, k % Pc
        g = G[k & 1]
        w = geo.height(k)
        lr, lc = (k // Pr) * nb, (k // Pc) * nb
        holder = pc == kc
        owner = holder and pr == kr
        nxt = k + 1 < nblk
        r1, c1 = (k + 1) % Pr, (k + 1) % Pc        # process row of block row k+1 / column of the next diagonal owner
        h1 = geo.height(k + 1) if nxt else 0
        early_off = geo.piece_off(r1) + geo.dsz      # L[k+1, k] is the first block of piece r1

        def wait_free():
            if k >= 2:
                ops.wait(_ev2(E_UPD, k - 2))
                ops.wait(_ev2(E_STORED, k - 2))
                ops.wait(_ev2(E_PANELDONE, k - 2))

        # ---- diagonal chain -------------------------------------------------------------------------------------
        if owner:
            ops.stream(PANEL)
            ops.wait(_ev2(E_DIAGREADY, k))
            wait_free()
            ops.diag_factor(A, lr, lc, w, g, geo.piece_off(kr), nb, k * nb, geo.n)
            ops.record(_ev2(E_DFACT, k))
        if holder:
            ops.stream(COMM)
            if owner:
                ops.wait(_ev2(E_DFACT, k))
            else:
                wait_free()
            comm.bcast_grp(g, geo.piece_off(kr), geo.dsz, kr, COL)      # L_kk (+ leaf inverses) down the process column
            ops.record(_ev2(E_DBC, k))
            ops.stream(PANEL)
            ops.wait(_ev2(E_DBC, k))
            ops.wait(_ev2(E_COLREADY, k))
            lr0, m = geo.row_off(pr, geo.li0(pr, k)), geo.piece_rows(pr, k)
            roff = geo.piece_off(pr) + geo.dsz
            if nxt and pr == r1:                                         # block row k+1 first: the next diagonal needs it
                ops.panel_trsm(A, lr0, h1, lc, w, g, geo.piece_off(kr), roff, nb)
                ops.record(_ev2(E_EARLYSOLVED, k))
                ops.panel_trsm(A, lr0 + h1, m - h1, lc, w, g, geo.piece_off(kr), roff + h1 * nb, nb)
            else:
                ops.panel_trsm(A, lr0, m, lc, w, g, geo.piece_off(kr), roff, nb)
            ops.record(_ev2(E_PIECE, k))
        if nxt and pr == r1:
            ops.stream(COMM)
            if holder:
                ops.wait(_ev2(E_EARLYSOLVED, k))
            else:
                wait_free()
            comm.bcast_grp(g, early_off, h1 * nb, kc, ROW)               # L[k+1, k] along the process row of block row k+1
            ops.record(_ev2(E_EARLY, k))
            if pc == c1:                                                 # owner of the next diagonal block
                ops.stream(PANEL)
                ops.wait(_ev2(E_EARLY, k))
                if k >= 1:
                    ops.wait(_ev2(E_COL2, k + 1))                        # contributions of the panels before k
                else:
                    # k = 0: nothing else orders this stream behind the ASSEMBLY of A (queued on MAIN); without it the
                    # update could land on the block before the fill wrote it and be overwritten -- seen on the first
                    # step of a fresh process only, when the fill kernel's first launch is slow (1 run in 12)
                    ops.wait(_ev2(E_COLREADY, 0))
                ops.update(A, ((k + 1) // Pr) * nb, h1, ((k + 1) // Pc) * nb, h1, g, early_off, early_off, w, nb)
                ops.record(_ev2(E_DIAGREADY, k + 1))
        ops.stream(PANEL)
        ops.record(_ev2(E_PANELDONE, k))
        # ---- panel chain ----------------------------------------------------------------------------------------
        ops.stream(COMM)
        if holder:
            ops.wait(_ev2(E_PIECE, k))
        else:
            wait_free()
        comm.panel_bcast(g, geo.pieces(k))                               # every piece to every rank, all links
        ops.record(_ev2(E_ARRIVED, k))
        ops.stream(BACK)
        ops.wait(_ev2(E_ARRIVED, k))
        if L is not None:
            ops.unpack_diag(g, geo.piece_off(kr), w, nb, L, k * nb)
            for p in range(Pr):
                m = geo.piece_rows(p, k)
                if m > 0:
                    ops.unpack_rows(g, geo.piece_off(p) + geo.dsz, m, w, nb, L, p + geo.li0(p, k) * Pr, Pr, k * nb)
        ops.record(_ev2(E_STORED, k))
        if on_stored is not None:
            on_stored(k)
        ops.stream(MAIN)
        ops.wait(_ev2(E_ARRIVED, k))
        cols = geo.my_cols_after(k)
        if cols and cols[0] == k + 1:                                    # look-ahead: column k+1 BELOW its diagonal block
            lr0, m, lc0, n, aoff, boff = geo.update_args(k, k + 1, below_diag=True)
            ops.update(A, lr0, m, lc0, n, g, aoff, boff, w, nb)
            cols = cols[1:]
        if nxt and pc == c1:
            ops.record(_ev2(E_COLREADY, k + 1))
        if cols and cols[0] == k + 2:                                    # two ahead: the diagonal chain needs it next step
            lr0, m, lc0, n, aoff, boff = geo.update_args(k, k + 2)
            ops.update(A, lr0, m, lc0, n, g, aoff, boff, w, nb)
            cols = cols[1:]
        if k + 2 < nblk and (k + 2) % Pc == pc:
            ops.record(_ev2(E_COL2, k + 2))
        for J in cols:
            lr0, m, lc0, n, aoff, boff = geo.update_args(k, J)
            ops.update(A, lr0, m, lc0, n, g, aoff, boff, w, nb)
        ops.record(_ev2(E_UPD, k))
    ops.stream(MAIN)
    info = ops.info()  # synchronises every stream
    if L is not None:
        ops.finish(L)
    allinfo = comm.allgather(np.array([float(info)]))[:, 0]
    bad = [int(v) for v in allinfo if v > 0]
    return min(bad) if bad else 0


def dist2_potrs(ops, comm, geo, A, yv, acc_r, acc_c, out):
    """alpha = K^-1 y on the block-cyclic factor A: block forward substitution (partial sums reduced along the process
    row of the diagonal owner, the solved block broadcast down its process column), then the transposed sweep with the
    roles of rows and columns exchanged; the blocks of alpha (one per diagonal owner) are assembled on every rank by one
    ncclAllReduce.  yv: device vector (padded N) holding y on every rank -- overwritten; acc_r / acc_c: scratch vectors
    of local_rows / local_cols doubles; out: device vector (padded N) that receives alpha everywhere."""
    nb, Pr, Pc, pr, pc = geo.nb, geo.Pr, geo.Pc, geo.pr, geo.pc
    ops.stream(MAIN)
    ops.vec_op(acc_r, 0, None, 0, max(geo.local_rows(pr), 1), 2)
    ops.vec_op(acc_c, 0, None, 0, max(geo.local_cols(pc), 1), 2)
    ops.vec_op(out, 0, None, 0, geo.np, 2)
    for k in range(geo.nblk):                                   # L w = y
        kr, kc = k % Pr, k % Pc
        w = geo.height(k)
        lr, lc = (k // Pr) * nb, (k // Pc) * nb
        if pr == kr:
            comm.reduce_grp(acc_r, lr, w, kc, ROW)              # -(sum_{J<k} L_kJ w_J), one partial per process column
        if pr == kr and pc == kc:
            ops.vec_op(yv, k * nb, acc_r, lr, w, 1)
            ops.trsv_diag(A, lr, lc, w, yv, k * nb, 0)
        if pc == kc:
            comm.bcast_grp(yv, k * nb, w, kr, COL)
            r0 = geo.row_off(pr, geo.li0(pr, k))
            ops.gemv(A, r0, geo.piece_rows(pr, k), lc, w, yv, k * nb, acc_r, r0, 0)
    for k in reversed(range(geo.nblk)):                         # L^T alpha = w
        kr, kc = k % Pr, k % Pc
        w = geo.height(k)
        lr, lc = (k // Pr) * nb, (k // Pc) * nb
        if pc == kc:
            comm.reduce_grp(acc_c, lc, w, kr, COL)              # -(sum_{I>k} L_Ik^T alpha_I), one partial per process row
        if pr == kr and pc == kc:
            ops.vec_op(yv, k * nb, acc_c, lc, w, 1)
            ops.trsv_diag(A, lr, lc, w, yv, k * nb, 1)
            ops.vec_op(out, k * nb, yv, k * nb, w, 0)
        if pr == kr:
            comm.bcast_grp(yv, k * nb, w, kc, ROW)
            ncb = geo.blocks_before(pc, Pc, k)                  # local block columns J < k
            ops.gemv(A, lr, w, 0, ncb * nb, yv, k * nb, acc_c, 0, 1)
    comm.allreduce(out, 0, geo.np)
    return out


def dist2_logdet(ops, comm, geo, A, scal):
    """log det K = 2 sum log L_ii: every diagonal owner adds its blocks, one ncclAllReduce of a scalar."""
    ops.stream(MAIN)
    ops.vec_op(scal, 0, None, 0, 1, 2)
    for k in range(geo.nblk):
        if k % geo.Pr == geo.pr and k % geo.Pc == geo.pc:
            ops.logdet_acc(A, (k // geo.Pr) * geo.nb, (k // geo.Pc) * geo.nb, geo.height(k), geo.height(k), scal)
    return float(comm.allreduce_host(np.array([ops.vec_to_host(scal, 1)[0]]))[0])


class DistFitIvar2D:
    """bench.py's multi-GPU step on the 2-D block-cyclic layout: distributed fit (local assembly + dist2_potrf), alpha
    by distributed substitution, logdet / y^T alpha through all-reduce, IVAR with the evaluation points sharded over the
    ranks against the replicated factor (streamed underneath the factorisation from 4 ranks)."""

    def __init__(self, ctx, comm, spec, Xh, yh, Zh, noise, nb=512, ops=None, streamed=None, grid=None):
        self.ctx, self.comm, self.spec = ctx, comm, spec
        self.ops = ops or DeviceOps2D(ctx)
        Pr, Pc = grid or choose_grid(comm.world)
        comm.set_grid(Pr, Pc)
        self.n, self.noise = Xh.shape[0], float(noise)
        self.geo = Grid2D(self.n, nb, Pr, Pc, comm.rank)
        env = os.environ.get("GPX_DIST_STREAM_IVAR")
        self.streamed = (comm.world >= 4) if streamed is None else bool(streamed)
        if env is not None:
            self.streamed = env == "1"
        self.yh = np.ascontiguousarray(yh, dtype=np.float64)
        self.m = Zh.shape[0]
        self.X = self.ops.points(Xh)
        lo, hi = eval_slice(self.m, comm.rank, comm.world)
        self.Zloc = self.ops.points(Zh[lo:hi]) if hi > lo else None
        self.A = self.ops.alloc_local(self.geo)
        self.G = [self.ops.alloc_buf(self.geo), self.ops.alloc_buf(self.geo)]
        self.L = self.ops.alloc_matrix(self.n)
        self.yv = self.ops.alloc_vec(self.geo.np)
        self.alpha = self.ops.alloc_vec(self.geo.np)
        self.acc_r = self.ops.alloc_vec(self.geo.local_rows(self.geo.pr))
        self.acc_c = self.ops.alloc_vec(self.geo.local_cols(self.geo.pc))
        self.scal = self.ops.alloc_vec(8)
        self.B = self.ops.alloc_cross(self.n, hi - lo) if (self.streamed and hi > lo) else None

    def step(self):
        ops, comm, geo = self.ops, self.comm, self.geo
        ops.stream(MAIN)
        ops.kfill_local(self.spec, self.X, self.A, self.noise, geo)
        hook = None
        if self.B is not None:
            ops.stream(BACK)
            ops.cross_fill(self.spec, self.X, self.Zloc, self.B)   # independent of the factorisation
            ops.stream(MAIN)

            def hook(k):
                ops.stream(BACK)                                    # behind the copy of panel k into L (same stream)
                ops.ivar_step(self.L, k, geo.nb, self.B)

        info = dist2_potrf(ops, comm, geo, self.A, self.G, L=self.L, on_stored=hook)
        if info:
            from ._lib import NotPositiveDefinite
            raise NotPositiveDefinite(info)
        ypad = np.zeros(geo.np)
        ypad[:self.n] = self.yh
        ops.vec_from_host(self.yv, ypad)
        dist2_potrs(ops, comm, geo, self.A, self.yv, self.acc_r, self.acc_c, self.alpha)
        logdet = dist2_logdet(ops, comm, geo, self.A, self.scal)
        part = 0.0
        if self.B is not None:     # the solve finished with the last panel (dist2_potrf synchronised every stream)
            part = float(np.sum(ops.variances(self.spec, self.Zloc, self.B, self.n)))
        elif self.Zloc is not None:
            part = float(np.sum(ops.posterior_var(self.spec, self.L, self.X, self.Zloc)))
        alpha = ops.vec_to_host(self.alpha, self.n)
        ll = -0.5 * float(self.yh @ alpha) - 0.5 * logdet - self.n / 2.0 * np.log(2 * np.pi)
        iv = abs(ordered_sum(comm.allgather(np.array([part]))[:, 0]) / self.m)
        return ll, iv
