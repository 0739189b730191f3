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
