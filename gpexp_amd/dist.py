"""Multi-GPU GP fit and evaluation: one process per GPU, RCCL over xGMI.

This replaces the reference's only parallel backend -- `parallelizeMcForLoop`, a fork + `mp.Queue` helper that
slices the evaluation points row-wise over CPU processes (parallel_utilities.py:26-80, used at gp.py:258) -- by
two shardings:

  fit   2-D BLOCK-CYCLIC (the default, `DistFitIvar2D` / `dist2_potrf`): process grid Pr x Pc (8 -> 2x4, 4 -> 2x2,
        2 -> 1x2), global block (I, J) of size nb on rank (I % Pr, J % Pc); assembly is communication-free; the
        right-looking Cholesky factors the diagonal block on its owner, broadcasts it down the process column
        (ncclCommSplit sub-communicator), the column's ranks solve their rows of the panel, every piece of the panel
        reaches every rank over all xGMI links (grouped ncclSend / ncclRecv scatter + all-gather), and every rank
        updates its local blocks -- the next two block columns at once (look-ahead: they gate the diagonal chain),
        everything else in GROUPS of `agg` panels (one launch over the whole local trailing matrix with K = agg * nb).
        Every rank also keeps the finished panels for the evaluation phase: a WINDOW of two groups of them when the
        evaluation is streamed underneath the factorisation (every panel is consumed when it arrives: no N x N replica, the
        factor stays distributed), a replicated copy of L otherwise (evaluation after the fit at 1-2 ranks, the C5 gradient).
        (`DistFitIvar` / `dist_potrf`: round 1's 1-D block-column layout, GPX_DIST_LAYOUT=1d.)
  eval  posterior / IVAR evaluation points are split in contiguous slices [r*M/W, (r+1)*M/W), exactly the
        chunking the reference's helper intends (parallel_utilities.py:46-60); each rank solves against its own
        copy of L and the only exchange is an all-gather of one partial sum per rank (summed in rank order, so
        the result does not depend on arrival order).

The panel loops are Python on top of C-ABI primitives (gpx_dist2_*), so that the very same loop also drives the NumPy
device double of the CPU tests (tests/dist_worker.py).  On the device the loop is not interpreted per step: it runs ONCE
against a recorder (`Program`), and the recorded rows are replayed natively by gpx_program_run on every step -- same call
sequence, no interpreter and no ctypes marshalling in the issue path.  Communicators (`bcast_grp`, `reduce_grp`,
`allreduce`, `panel_bcast`, `allgather`, `barrier`, `max_float`):
  RcclComm    device-to-device over RCCL (the product path; no torch in the process, see FileRendezvous);
  (scripts/replay_comm.py: ReplayComm, a MEASUREMENT double -- one process plays rank (pr, pc) of a larger grid, every
              receive becomes a device copy of the same bytes out of a complete factor resident on the GPU; it measures a
              rank's GPU time and host issue time without the other GPUs and is not part of the product package);
  tests/dist_testcomm.py holds the host-staged gloo communicator of the shared-GPU tests (RCCL refuses two ranks on
              one device); GPX_COMM=host selects it, nothing in this package imports torch.
The index arithmetic (ownership, slices, piece offsets, update ranges) is plain Python and unit-tested on CPU with gloo.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import device as _dev
from ._lib import check, dptr, as_f64, c_i64, GpxError, hdot

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


# ---- recorded programs (include/gpx.h: gpx_program_run) -----------------------------------------------------------
OP = dict(STREAM=1, RECORD=2, WAIT=3, BEGIN=4, DIAG_FACTOR=5, PANEL_TRSM=6, UPDATE=7, UPDATE_MULTI=8, UNPACK_ROWS=9,
          UNPACK_DIAG=10, PACK_ROWS=11, PACK_DIAG=12, BCAST_GRP=13, REDUCE_GRP=14, ALLREDUCE=15, PANEL_BCAST=16, IVAR_STEP=17,
          TRSV_DIAG=18, GEMV=19, LOGDET_ACC=20, VEC_OP=21, SPIN=22, COPY=23, IVAR_GROUP=24, FWD_GROUP=25, PANEL_INV=26, BCAST_GRP2=27, PANEL_COPYBACK=28,
          DIAG_STAGE=29, DIAG_UPDATE=30, DIAG_FACTOR_STAGED=31, DIAG_STORE=32, PANEL_PACK=33, DIAG_PACK=34)


class Program:
    """Rows of 16 int64 [opcode, handle0..2, a0..a11] + a pool of variable-length arguments; replayed by gpx_program_run."""

    def __init__(self):
        self.rows, self.extra, self.keep = [], [], []
        self._frozen = None
        self.host_ms = 0.0   # host time of the last run (issue only: every op is asynchronous)

    def emit(self, op, handles=(), args=(), extra=None):
        row = [int(op), 0, 0, 0] + [0] * 12
        for i, h in enumerate(handles):
            if h is not None:
                row[1 + i] = int(h.h.value)
                self.keep.append(h)          # the matrix must outlive the program
        args = [int(a) for a in args]
        if extra is not None:
            args = args + [len(self.extra)]
            self.extra.extend(int(e) for e in extra)
        assert len(args) <= 12, (op, args)
        row[4:4 + len(args)] = args
        self.rows.append(row)
        self._frozen = None

    def __len__(self):
        return len(self.rows)

    def _arrays(self):
        if self._frozen is None:
            self._frozen = (np.ascontiguousarray(np.array(self.rows, dtype=np.int64).reshape(-1, 16)),
                            np.ascontiguousarray(np.array(self.extra + [0], dtype=np.int64)))
        return self._frozen

    def run(self, ctx):
        ops, extra = self._arrays()
        ms = C.c_double(0.0)
        check(ctx.lib.gpx_program_run(ctx.h, ops.ctypes.data_as(C.POINTER(c_i64)), ops.shape[0],
                                      extra.ctypes.data_as(C.POINTER(c_i64)), extra.size, C.byref(ms)))
        self.host_ms = ms.value
        return ms.value

    # ---- the program as a hipGraph: one launch per step (gpx_program_capture; only after one ordinary run) ----
    graph = None
    graph_nodes = 0

    def capture(self, ctx):
        ops, extra = self._arrays()
        h = C.c_void_p()
        check(ctx.lib.gpx_program_capture(ctx.h, ops.ctypes.data_as(C.POINTER(c_i64)), ops.shape[0],
                                          extra.ctypes.data_as(C.POINTER(c_i64)), extra.size, C.byref(h)))
        self.graph, self._ctx = h, ctx

    def launch(self, ctx):
        ms = C.c_double(0.0)
        n = c_i64(0)
        check(ctx.lib.gpx_graph_launch(ctx.h, self.graph, C.byref(ms), C.byref(n)))
        self.host_ms, self.graph_nodes = ms.value, n.value
        return ms.value

    def __del__(self):
        try:
            if self.graph is not None and self._ctx.h:
                self._ctx.lib.gpx_graph_free(self._ctx.h, self.graph)
            self.graph = None
        except Exception:
            pass


class Emitter:
    """Mixin of the device-primitive and communicator objects: every call becomes a program row, appended to `self.prog`
    while a program is being recorded, executed at once (a one-row program: the same native dispatch) otherwise."""
    prog = None

    def _emit(self, op, handles=(), args=(), extra=None):
        if self.prog is not None:
            self.prog.emit(op, handles, args, extra)
            return
        one = Program()
        one.emit(op, handles, args, extra)
        one.run(self.ctx)


# ---- communicators -----------------------------------------------------------------------------------------
class FileRendezvous:
    """Single-node exchange of the 128-byte ncclUniqueId without any framework in the process: rank 0 writes it
    atomically to a file keyed by the launcher's (MASTER_PORT, run id, launcher pid); the others poll for it.
    (Importing torch next to the system RCCL loads a second, uninitialised HSA runtime into the process and
    ncclCommInitRank then fails with "no ROCm-capable device", so the RCCL path stays torch-free.)"""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        key = os.environ.get("GPX_RDV_KEY") or "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"),
                                                             os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
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


class RcclComm(Emitter):
    """Device-side collectives over RCCL; barriers and the timing max are RCCL all-gathers of one double.  The stream-ordered
    collectives (bcast_grp, reduce_grp, allreduce, panel_bcast) are recordable (Emitter)."""
    recordable = True

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
        self._emit(OP["BCAST_GRP"], (buf,), (offset, count, root, grp))

    def bcast_grp2(self, sbuf, soff, rbuf, roff, count, root, grp):
        """out-of-place broadcast: the root sends sbuf[soff ..], every member receives into rbuf[roff ..]"""
        self._emit(OP["BCAST_GRP2"], (sbuf, rbuf), (soff, roff, count, root, grp))

    def reduce_grp(self, buf, offset, count, root, grp):
        self._emit(OP["REDUCE_GRP"], (buf,), (offset, count, root, grp))

    def allreduce(self, buf, offset, count):
        self._emit(OP["ALLREDUCE"], (buf,), (offset, count))

    def panel_bcast(self, buf, pieces):
        self._emit(OP["PANEL_BCAST"], (buf,), (len(pieces),),
                   extra=[p[0] for p in pieces] + [p[1] for p in pieces] + [p[2] for p in pieces])

    def allreduce_host(self, vec):
        vec = as_f64(np.atleast_1d(vec)).copy()
        check(self.ctx.lib.gpx_comm_allreduce_host(self.ctx.h, dptr(vec), vec.size))
        return vec

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


def init_from_env(ctx):
    """Communicator for the process group the launcher created (RANK / WORLD_SIZE / MASTER_* in the environment).
    GPX_COMM=host selects the TEST communicator (tests/dist_testcomm.py: device -> host -> gloo -> device, several ranks
    sharing one GPU); it is test infrastructure and is imported from the tests directory, never shipped in this package."""
    kind = os.environ.get("GPX_COMM", "rccl")
    if kind == "host":
        import sys
        tests = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")
        if tests not in sys.path:
            sys.path.insert(0, tests)
        from dist_testcomm import HostStagedComm   # noqa: E402  (test-only)
        return HostStagedComm(ctx)
    return RcclComm(ctx)


# ---- distributed operations -----------------------------------------------------------------------------------
MAIN, PANEL, COMM, BACK, EVAL, BULK = 0, 1, 2, 3, 4, 5  # stream indices of the context (gpx_stream_select); BACK and BULK are
                                                         # CU-masked (4 CUs per XCD stay free for the diagonal chain)


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

    def cross_mean(self, spec, X, Z, alpha):
        """K(Z, X) alpha on the host: the posterior mean of a slice of evaluation points without the factor (its own small fill:
        the solve overwrites K(X, Z) in place)."""
        return _dev.matvec(self.ctx, _dev.kfill(self.ctx, spec, Z, Z=X), as_f64(alpha))

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
            ll = -0.5 * hdot(self.yh, alpha) - 0.5 * logdet - self.n / 2.0 * np.log(2 * np.pi)
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


def dist_greedy_ivar(ctx, comm, spec, L, X, cand_host, mc_host, noise, nsel, want_all=False, be=None):
    """nsel picks of discrete greedy IVAR (composition of experimentalDesign.py:79-117) with the CANDIDATES sharded
    and the state RESIDENT (gpx_givar_*): every rank holds the replicated factor, W_C and cov(Z, C | design) for its contiguous
    slice of the candidates and the whole of Z.  Per pick: local first minimum -> one (cost, global index) pair per rank, merged
    by np.argmin's rule; the winner's owner packs its pivot (delta, the point, cov(Z, c_s) / sqrt(delta), its column of W_C, its
    coordinates along the earlier picks: 2 + d + nMC + N + nsel doubles) and broadcasts it; every rank conditions its slice on
    it (one pass over its W_C, one over its G).  No refit, no N^2 solve after the set-up.  Returns (indices, costs[, all costs])."""
    be = be or _dev
    m = cand_host.shape[0]
    nsel = int(nsel)
    lo, hi = eval_slice(m, comm.rank, comm.world)
    Z = be.points(ctx, mc_host)
    # (the slice carries the bounding box of ALL candidates: same centring, same arithmetic per candidate as on one rank)
    slice_pts = be.points_slice(ctx, cand_host, lo, hi) if hasattr(be, "points_slice") else be.points(ctx, cand_host[lo:hi])
    st = be.GivarState(ctx, spec, L, X, slice_pts, Z, noise, nsel) if hi > lo else None
    npad = padded(X.shape[0])
    pack = 2 + spec.d + padded(mc_host.shape[0]) + npad + nsel
    assert st is None or st.pivot_elems == pack
    buf = be.alloc_vector(ctx, pack)
    idx, costs = np.empty(nsel, dtype=np.int64), np.empty(nsel)
    allc = np.empty((nsel, m)) if want_all else None
    per = (m + comm.world - 1) // comm.world
    for t in range(nsel):
        if st is not None:
            c, i, loc = st.score(want_all)
            mine = np.array([c, float(lo + i)]) if np.isfinite(c) else np.array([np.inf, float(m)])   # (nothing finite: as an empty slice)
        else:
            loc, mine = np.zeros(0), np.array([np.inf, float(m)])
        pairs = comm.allgather(mine)
        cost, s = merge_argmin(pairs[:, 0], pairs[:, 1].astype(np.int64))
        if not np.isfinite(cost):            # the same on every rank (the merge is)
            raise RuntimeError("greedy IVAR: pick %d of %d: no candidate has a finite cost (every remaining one is already in "
                               "the design with zero noise, or the state is not finite)" % (t + 1, nsel))
        idx[t], costs[t] = s, cost
        if want_all:
            pad = np.zeros(per)
            pad[:hi - lo] = loc
            g = comm.allgather(pad)
            for r in range(comm.world):
                a, b = eval_slice(m, r, comm.world)
                allc[t, a:b] = g[r, :b - a]
        if t + 1 == nsel:
            break
        owner_rank = mi_owner(m, s, comm.world)
        if comm.rank == owner_rank:
            st.pack(s - lo, buf)
        comm.bcast_grp(buf, 0, pack, owner_rank, WORLD)
        if st is not None:
            st.apply(buf)
    return (idx, costs, allc) if want_all else (idx, costs)


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
G_SKEW = 0                  # packed panel rows have stride nb + G_SKEW doubles (= gpx_dist2_row_stride(nb) - nb; a skew of 16
                            # was measured and made no difference on MI355X)


MAX_PR = 4                  # GPX_MAX_PR of csrc/gpx_internal.h: the segmented trailing update addresses at most 4 process rows


def choose_grid(world):
    """Pr x Pc with Pr <= Pc, Pr the largest divisor of `world` not above min(sqrt(world), MAX_PR): 8 -> 2x4, 4 -> 2x2, 2 -> 1x2,
    25 -> 1x25 (a 5x5 grid would be rejected by the update kernel inside the first step)."""
    pr = 1
    for c in range(1, min(int(world ** 0.5), MAX_PR) + 1):
        if world % c == 0:
            pr = c
    return pr, world // pr


class Grid2D:
    """Pure index logic of the 2-D block-cyclic layout (CPU-testable)."""

    def __init__(self, n, nb, Pr, Pc, rank):
        assert nb % TILE == 0 and 0 <= rank < Pr * Pc
        assert 1 <= Pr <= MAX_PR, "at most %d process rows (GPX_MAX_PR)" % MAX_PR
        self.n, self.nb, self.Pr, self.Pc, self.rank = int(n), int(nb), int(Pr), int(Pc), int(rank)
        self.pr, self.pc = rank // Pc, rank % Pc
        self.np = padded(n)
        self.nblk = num_blocks(n, nb)
        self.dsz = nb * nb + (nb // TILE) * TILE * TILE
        self.gld = nb + G_SKEW               # row stride of the packed rows (the D region keeps stride nb)
        self.piece_stride = self.dsz + max(self.local_rows(p) for p in range(Pr)) * self.gld

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
                out.append((self.piece_off(p), self.dsz + m * self.gld, p * self.Pc + kc))
            elif m > 0:
                out.append((self.piece_off(p) + self.dsz, m * self.gld, p * self.Pc + kc))
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
        aoff = self.piece_off(pr) + self.dsz + (liJ - self.li0(pr, k)) * nb * self.gld
        pj = J % self.Pr
        boff = self.piece_off(pj) + self.dsz + (J // self.Pr - self.li0(pj, k)) * nb * self.gld
        return self.row_off(pr, liJ), m, (J // self.Pc) * nb, self.height(J), aoff, boff


class DeviceOps2D(Emitter, DeviceOps):
    """Device primitives of the 2-D panel loop (gpx_dist2_*); the tests substitute a NumPy double.  Everything the loop
    enqueues goes through the Emitter, so the loop can be recorded once and replayed natively (Program)."""

    def alloc_local(self, geo):
        return _dev.DeviceMatrix.zeros(self.ctx, max(geo.local_rows(geo.pr), 1), max(geo.local_cols(geo.pc), 1))

    def alloc_buf(self, geo):
        return _dev.DeviceMatrix.zeros(self.ctx, geo.buf_elems(), 1, pad=False)

    def alloc_vec(self, n):
        return _dev.DeviceMatrix.zeros(self.ctx, max(int(n), 1), 1, pad=False)

    def reserve(self, nb, agg, mcols):
        """the context's panel-solve / streamed-evaluation scratches, sized before the first step (never mid-step)"""
        check(self.ctx.lib.gpx_dist2_reserve(self.ctx.h, int(nb), int(agg), int(mcols)))

    def kfill_local(self, spec, X, A, nugget, geo):
        nug, nlen = _dev._nugget_args(nugget, X.shape[0])
        # tests: hold the assembly of ONE rank back ("ms@rank"), so that a missing dependency on it shows every time
        spec_delay = os.environ.get("GPX_TEST_DELAY_FILL", "")
        delay = int(spec_delay.split("@")[0]) if spec_delay and int(spec_delay.split("@")[1]) == geo.rank else 0
        if delay:
            check(self.ctx.lib.gpx_dbg_spin(self.ctx.h, delay))
        check(self.ctx.lib.gpx_dist2_kfill(self.ctx.h, *spec.args(), X.h, dptr(nug), nlen, A.h, geo.nb, geo.Pr, geo.Pc,
                                           geo.pr, geo.pc))

    # stream / event plumbing.  A wait on an event that was recorded on the SAME stream is dropped: the stream is in order, so
    # it orders nothing -- and inside a stream capture it crashes hipStreamEndCapture on ROCm 7.2 when the stream is not the
    # capture's origin (scripts/_graph_bisect.py: PANEL records E_DIAGREADY after the early update and waits for it one step
    # later as the diagonal owner).
    _cur_stream = MAIN
    _ev_stream = None

    def stream(self, which):
        self._cur_stream = int(which)
        self._emit(OP["STREAM"], (), (which,))

    def record(self, ev):
        if self._ev_stream is None:
            self._ev_stream = {}
        self._ev_stream[int(ev)] = self._cur_stream
        self._emit(OP["RECORD"], (), (ev,))

    def wait(self, ev):
        if self._ev_stream is not None and self._ev_stream.get(int(ev)) == self._cur_stream:
            return
        self._emit(OP["WAIT"], (), (ev,))

    def begin(self):
        self._emit(OP["BEGIN"])

    def diag_factor(self, A, lr, lc, w, G, doff, nb, base, n_valid):
        self._emit(OP["DIAG_FACTOR"], (A, G), (lr, lc, w, doff, nb, base, n_valid))

    # the same in parts (round 4: only the factorisation itself stays on the chain across ranks, see dist2_potrf_enqueue)
    def diag_stage(self, A, lr, lc, w, G, doff, nb):
        self._emit(OP["DIAG_STAGE"], (A, G), (lr, lc, w, doff, nb))

    def diag_update(self, G, doff, h, S, soff, w, nb):
        """staged block (h x h) -= S[soff](h x w packed rows) times their transpose"""
        self._emit(OP["DIAG_UPDATE"], (G, S), (doff, h, soff, w, nb))

    def diag_factor_staged(self, A, lr, lc, w, G, doff, nb, base, n_valid):
        self._emit(OP["DIAG_FACTOR_STAGED"], (A, G), (lr, lc, w, doff, nb, base, n_valid))

    def diag_store(self, A, lr, lc, w, G, doff, nb, dslot=None):
        self._emit(OP["DIAG_STORE"], (A, G), (lr, lc, w, doff, nb, 0 if dslot is None else dslot + 1))

    def panel_inv(self, G, doff, nb, w):
        """explicit inverse of the factored diagonal block in G at doff, kept for this step's panel solves (built as soon as the
        block has arrived: off the panel chain)"""
        self._emit(OP["PANEL_INV"], (G,), (doff, nb, w))

    def panel_trsm(self, A, lr0, m, lc, w, G, doff, roff, nb, dslot=None, prepared=False, copy_back=True):
        """dslot (owner of the diagonal block only): its local block row -- the explicit inverse the solve builds is kept in A
        for the distributed substitution's diagonal solves.  prepared: panel_inv ran for this step's diagonal block; copy_back =
        False (with prepared, full-width blocks): the solved rows go into the packed buffer only, panel_copyback follows."""
        flag = 0 if not prepared else (1 if copy_back else 2)
        self._emit(OP["PANEL_TRSM"], (A, G), (lr0, m, lc, w, doff, roff, nb, 0 if dslot is None else dslot + 1, flag))

    def panel_copyback(self, A, lr0, m, lc, w, G, roff, nb):
        self._emit(OP["PANEL_COPYBACK"], (A, G), (lr0, m, lc, w, roff, nb))

    def panel_pack(self, A, lr0, m, lc, w, G, roff, nb):
        """rows of a FINISHED panel from the local matrix into the packed buffer (re-streamed factor, dist2_restream_enqueue)"""
        self._emit(OP["PANEL_PACK"], (A, G), (lr0, m, lc, w, roff, nb))

    def diag_pack(self, A, lr, lc, w, G, doff, nb):
        """the factored diagonal block + its leaf inverses from the local matrix into the D region of the packed buffer"""
        self._emit(OP["DIAG_PACK"], (A, G), (lr, lc, w, doff, nb))

    def update(self, A, lr0, m, lc0, n, G, aoff, boff, w, nb):
        self._emit(OP["UPDATE"], (A, G), (lr0, m, lc0, n, aoff, boff, w, nb))

    def update_multi(self, A, lr0, m, lc0, n, geo, Gs, ks, below_diag):
        """A[lr0:lr0+m, lc0:lc0+n] -= contributions of the panels ks (packed buffers Gs) on / below the global diagonal."""
        if m <= 0 or n <= 0 or not ks:
            return
        self._emit(OP["UPDATE_MULTI"], (A,),
                   (lr0, m, lc0, n, geo.nb, geo.Pr, geo.Pc, geo.pr, geo.pc, geo.piece_stride, len(ks) | (int(bool(below_diag)) << 8)),
                   extra=[int(g.h.value) for g in Gs] + list(ks))
        if self.prog is not None:
            self.prog.keep.extend(Gs)

    def unpack_rows(self, G, roff, m, w, nb, L, first_block, stride, col0):
        self._emit(OP["UNPACK_ROWS"], (G, L), (roff, m, w, nb, first_block, stride, col0))

    def unpack_diag(self, G, doff, w, nb, L, r0, c0=None):
        """c0: column of the block inside L when L is a WINDOW of block columns (default: the diagonal, c0 = r0)."""
        self._emit(OP["UNPACK_DIAG"], (G, L), (doff, w, nb, r0, 0 if c0 is None else c0 + 1))

    def ivar_step(self, K, k, nb, B):
        self._emit(OP["IVAR_STEP"], (K, B), (k, nb))

    def ivar_group(self, K, k0, k1, nb, B, c0=None):
        """c0: column of panel k0 inside K when K is a window of block columns (default k0 * nb)."""
        self._emit(OP["IVAR_GROUP"], (K, B), (k0, k1, nb, 0 if c0 is None else c0 + 1))

    def fwd_group(self, K, k0, k1, nb, v, c0=None):
        """the group's step of the forward substitution on the vector v, against the same (window of the) factor"""
        self._emit(OP["FWD_GROUP"], (K, v), (k0, k1, nb, 0 if c0 is None else c0 + 1))

    def alloc_window(self, n, cols):
        return _dev.DeviceMatrix.zeros(self.ctx, n, cols)

    # alpha / log-det straight from a REPLICATED factor (every rank holds it: no exchange, the single-GPU sweeps)
    def replica_solve(self, L, y0, alpha):
        _dev.potrs_dev(self.ctx, L, y0, alpha)

    def replica_logdet(self, L):
        return _dev.logdet(self.ctx, L)

    def trsv_diag(self, A, lr, lc, w, v, voff, transposed):
        self._emit(OP["TRSV_DIAG"], (A, v), (lr, lc, w, voff, int(transposed)))

    def gemv(self, A, lr0, m, lc, w, x, xoff, acc, aoff, transposed):
        self._emit(OP["GEMV"], (A, x, acc), (lr0, m, lc, w, xoff, aoff, int(transposed)))

    def logdet_acc(self, A, lr, lc, w, n_valid, acc):
        self._emit(OP["LOGDET_ACC"], (A, acc), (lr, lc, w, n_valid))

    def vec_op(self, dst, doff, src, soff, n, mode):
        self._emit(OP["VEC_OP"], (dst, src), (doff, soff, n, mode))

    def spin(self, ms):
        self._emit(OP["SPIN"], (), (ms,))

    def spin_us(self, us):
        self._emit(OP["SPIN"], (), (int(us), 1))

    def vec_to_host(self, v, n):
        return v.to_host()[:n, 0]

    def vec_from_host(self, v, a):
        a = as_f64(np.ravel(a))
        check(self.ctx.lib.gpx_mat_write(self.ctx.h, v.h, 0, a.size, dptr(a)))


# event kinds of the 2-D pipeline (per step k)
(E_COLREADY, E_DFACT, E_DBC, E_PIECE, E_ARRIVED, E_STORED, E_UPD, E_DIAGREADY, E_EARLYSOLVED, E_EARLY, E_COL2,
 E_PANELDONE, E_BULK, E_IVAR, E_PIECE0, E_ARR0, E_COLREADY0) = range(17)   # the last three: FIRST row chunk of a panel (round 5)
N_EVENT_KINDS = 17


def _ev2(kind, k):
    return N_EVENT_KINDS * (k + 1) + kind


EV_FORK, EV_PRE, EV_JOIN0 = 1, 2, 3            # event ids below N_EVENT_KINDS are free (ids of step k start at N_EVENT_KINDS (k + 1))
ALL_SIDE_STREAMS = (PANEL, COMM, BACK, EVAL, BULK)


def default_agg(streamed=True, world=8, nb=512):
    """Panels per aggregated trailing update (GPX_DIST_AGG; K = agg * nb per launch).  4 -> K = 2048 at nb = 512 where the
    evaluation is streamed underneath the factorisation (the rank is throughput-bound: longer K, fewer launches); 2 for the
    factorisation alone, whose time on a real grid is the chain ACROSS ranks: shorter-lived bulk tiles give the chain's kernels
    their slots sooner (paced replay of the 2 x 4 grid at C4: 51 ms with 2, 54 with 4 or 1; fit + IVAR: 113 with either).  Blocks
    of 1024 (default_nb) carry half as many: the same K."""
    # (1-2 ranks: each rank carries half or all of the trailing updates -- throughput-bound like the streamed case; world-1 RCCL
    # bench 695 ms with 4 panels per update, 710 with 2)
    dflt = 4 if (streamed or world < 4) else 2
    if nb >= 1024:
        dflt = max(1, dflt // 2)
    return max(1, min(8, int(os.environ.get("GPX_DIST_AGG", str(dflt)))))


def default_nb(n, world=8, streamed=False):
    """Block size of the 2-D block-cyclic layout (GPX_DIST_NB).  512 from N = 8192 (256 / 128 below: at least a few block steps
    per rank).  Eight ranks and more, factorisation without a streamed evaluation, N >= 16384: 1024 -- the time of such a grid is
    the chain ACROSS ranks, a sum over the block steps of ~20 small dependent launches and two hops, and half as many steps
    with twice the work each is shorter (paced replay of the 2 x 4 grid at C4, all ranks, diagonal chain paced: 44.6 ms against
    48-50 at 512; 2048: 47.8, the panel solves and near updates of a step then outweigh what the fewer steps save; a 2 x 2 grid is
    throughput-bound and indifferent: 61.5 / 61.8 ms -- profiles/r04_dist_paced_sweeps2.txt)."""
    env = os.environ.get("GPX_DIST_NB")
    if env:
        return int(env)
    if n >= 16384 and world >= 8 and not streamed:
        return 1024
    return 512 if n >= 8192 else (256 if n >= 2048 else 128)


def ring_size(agg):
    """Packed panel buffers the loop cycles through: a panel's buffer is read until the bulk update of ITS group has run,
    which may finish one group late (it runs beside the next group's chain) -- at least two groups of buffers.  Round 4: at
    least 8 of them whatever the group size: the communication stream may not overwrite a buffer before the bulk
    update that reads it has run, so a SHORT ring ties the chain across ranks to the slowest rank's backlog of bulk updates
    (paced replay, two panels per update: with 4 buffers the last panel reached the two busiest ranks 8 ms after the others)."""
    return max(2 * agg, 8)


def dist2_potrf(ops, comm, geo, A, G, L=None, on_stored=None, agg=None, window=0, E=None):
    """2-D block-cyclic right-looking Cholesky of the distributed matrix A (in place: A ends as the block-cyclic factor)
    with look-ahead, a CRITICAL-PATH-FIRST diagonal chain and AGGREGATED trailing updates.  G = ring of packed panel
    buffers (geo.buf_elems() doubles each, len(G) >= ring_size(agg), or 2 with agg = 1); L (optional) = full-size matrix
    that receives every finished panel (replicated factor for the evaluation phase).

    What serialises a distributed factorisation is the chain diag(k) -> panel(k) -> update of column k+1 -> diag(k+1).  Only
    ONE nb x nb block of panel k enters diag(k+1): L[k+1, k].  So, per step k,
      diagonal chain   owner of (k,k) factors it; the holder of block row k+1 solves THAT block first and sends it along its
                       process row (one small ncclBroadcast on the row sub-communicator); the owner of (k+1,k+1) applies it
                       to the diagonal block (already up to date through panel k-1, see "near" below) and can factor at
                       once: potrf(nb) + two small hops per step;
      panel chain      meanwhile the rest of panel k is solved, every piece goes to every rank over all links
                       (gpx_comm_panel_bcast);
      near updates     (MAIN) column k+1 below its diagonal block gets panel k (releases panel k+1's solve); column k+2 is
                       brought up to date through panel k in one launch over the panels of the current group;
      group end        every `agg` steps: the next `agg` block columns (the ones that become "near" during the next group)
                       get the whole group at once (MAIN, K = agg nb), and everything to the right of them gets it on the
                       CU-masked BULK stream -- ONE launch over the whole local trailing matrix, beside the next group's
                       chain.  (Round 2 applied every panel to every block column separately: K = nb, one launch per block
                       column, 39 TF/s at C4 on one rank; profiles/r03_dist_w1_before.txt.)
    Streams per rank: PANEL (diagonal factor, solves, the early diagonal update), COMM (collectives, enqueued in the same
    order on every rank of every communicator), MAIN (near updates), BULK (aggregated updates), BACK (copies into L, then the
    streamed-evaluation hook `on_stored(k)`).  A panel buffer is rewritten one ring later, after everything that reads it.
    Returns 0 or the 1-based index of the first non-positive pivot (agreed by all)."""
    dist2_potrf_enqueue(ops, comm, geo, A, G, L=L, on_stored=on_stored, agg=agg, window=window, E=E)
    return dist2_potrf_finish(ops, comm, None if window else L)


def dist2_potrf_finish(ops, comm, L=None):
    info = ops.info()  # synchronises every stream
    if L is not None:
        ops.finish(L)
    allinfo = comm.allgather(np.array([float(info)]))[:, 0]
    bad = [int(v) for v in allinfo if v > 0]
    return min(bad) if bad else 0


def dist2_potrf_enqueue(ops, comm, geo, A, G, L=None, on_stored=None, agg=None, window=0, E=None):
    """The asynchronous part of dist2_potrf: pure enqueue, no host read -- recordable (Program).

    window > 0: L is not a full-size replica but a WINDOW of `window` block columns (padded N rows x window * nb columns);
    panel k lands in column slot k % window, behind the streamed-evaluation step that consumed the slot's previous panel
    (event E_IVAR of that panel's group, recorded by the hook: streamed_ivar_hook).  window must be a multiple of the group
    size and at least two groups."""
    nb, Pr, Pc, pr, pc = geo.nb, geo.Pr, geo.Pc, geo.pr, geo.pc
    nblk = geo.nblk
    q = default_agg() if agg is None else int(agg)
    R = len(G)
    assert R >= (ring_size(q) if q > 1 else 2), "panel-buffer ring too short for the aggregation depth"
    # E (optional, round 4): a ring of small buffers (len(G) of them, nb x row-stride doubles) that receive the block row the
    # NEXT diagonal needs (L[k+1, k], sent along its process row ahead of the panel).  Without it that block lands in the panel
    # buffer, where the panel broadcast writes the same bytes again a moment later -- and the communication stream of the next
    # diagonal's owner has to hold the panel broadcast back until the kernel reading the block is done.  The panel
    # broadcast is a collective: one rank entering it late delays every rank -- 1.0-1.3 ms on each step, on the chain across
    # ranks (paced replay: the last panel reached the diagonal owners of process row 1 8 ms after everybody else).
    if E is not None and (len(E) < len(G) or not hasattr(comm, "bcast_grp2")):
        E = None
    window = int(window)
    assert window == 0 or (L is not None and on_stored is not None and window % q == 0 and window >= 2 * q), \
        "a window of the factor needs the streamed evaluation that consumes it, and two whole groups of column slots"
    # Where the group-end bulk update runs (GPX_DIST_BULK: chunks | eval | bulk | main):
    #   eval    (default) whole, on the low-priority unmasked stream beside the next group's chain and near updates: MAIN stays
    #           free for the near updates that gate the panel chain.
    #   chunks  on MAIN, cut into `q` column ranges of equal work that are issued one per step BEHIND the near updates of the
    #           following steps -- one GEMM stream, the big launches never share the chip with each other.  Expected to win
    #           where the GPU is the bound (1-2 ranks); measured it does not (the thin near updates -- 2 to 4 rounds of tiles,
    #           42-57 TF/s -- then run alone instead of underneath a bulk launch).
    #   bulk    whole, on the second CU-masked stream (its own queue; default where the evaluation is streamed underneath the
    #           factorisation -- `on_stored` given, 4 ranks and more: there EVAL already carries the per-group IVAR solves and the
    #           bulk update would queue behind them);
    #   main    whole, on MAIN.
    # Measured (replay, C4 fit + IVAR, ms at 1 / 2 / 4 / 8 ranks, MAIN at normal priority): eval 202 / 114 / 192 / 103,
    # bulk 244 / 134 / 185 / 100, chunks 225 / 124 / 191 / 103; fit alone at 4 / 8 ranks: eval 74 / 46, bulk 87 / 51 (the masked
    # stream has 224 of the 256 CUs) -- profiles/r03_dist_replay_*.json.
    mode = os.environ.get("GPX_DIST_BULK", "bulk" if on_stored is not None else "eval")
    bulk_stream = {"bulk": BULK, "eval": EVAL, "main": MAIN, "chunks": MAIN}[mode]
    chunked = mode == "chunks"
    at_step = getattr(comm, "at_step", None)
    hoist_inv = os.environ.get("GPX_DIST2_HOIST_INV", "1") == "1" and hasattr(ops, "panel_inv")
    # GPX_DIST_GATE_BULK: 0 = the group-end bulk update starts as soon as the group's panels are there; 1 = on the ranks that hold
    # the NEXT block column, behind their update of that column; 2 = on those ranks, behind their whole holder step (issued one
    # step later, behind E_PIECE of that step): a chain kernel beside a chip-filling update runs at half its rate or less
    gate_level = int(os.environ.get("GPX_DIST_GATE_BULK", "2"))
    gate_bulk = gate_level >= 1
    deferred_bulk = []           # (group end k, panels) whose bulk update is issued behind this rank's next holder step
    late_copyback = os.environ.get("GPX_DIST2_LATE_COPYBACK", "1") == "1"
    # Round 4, the diagonal chain (potrf(k) -> broadcast -> inverse -> block row k+1 solved -> broadcast -> last update of block
    # (k+1, k+1) -> potrf(k+1)) is ~20 small dependent launches per step, and their latencies -- 2-3 times the idle-chip figure
    # beside the trailing updates -- summed over the steps ARE the factorisation time of a grid (scripts/dist_replay.py
    # --paced-grid).  Staged: the owner of the next diagonal block copies it into the packed buffer BEFORE the block row it waits
    # for arrives, applies that last update to the copy, factors the copy in place, and copies factor + leaf inverses back into
    # the local matrix behind the event that releases the panel broadcast: three launches less per step on that chain.
    staged = os.environ.get("GPX_DIST2_STAGED_DIAG", "1") == "1" and hasattr(ops, "diag_stage")
    # Round 5: the panel travels in TWO ROW CHUNKS.  Per step the holder column used to add
    # [near update of column k] + [panel solve] + [hand-over] to the chain across ranks, each over the WHOLE piece (early steps
    # at C4 on a 2 x 4 grid: 0.88 + 0.36 + 0.33 ms).  But chunk c of panel k+1 needs chunk c of panel k only (the same local
    # rows: updated by it, then solved) -- plus the diagonal block, which has its own early path.  So the piece of every process
    # row is cut at a local row S: chunk 0 = the rows above S (solved, broadcast, applied to column k+1 first), chunk 1 = the
    # rest, one stage behind on each of the three streams (PANEL solve / COMM broadcast / MAIN near update).  S stays put for
    # `chunk_hold` steps (chunk 0 shrinks from the top as the factorisation moves down) and is then re-centred; a step whose
    # chunk 0 is not covered by the previous step's chunk-0 update waits for the whole column (one bubble per re-centring).
    # Chunk 0 always holds the first block row of its piece: that is the block row the near updates multiply with.
    nchunk = int(os.environ.get("GPX_DIST_PANEL_CHUNKS", "2"))
    chunk_hold = max(1, int(os.environ.get("GPX_DIST_CHUNK_HOLD", "4")))
    chunk_min = int(os.environ.get("GPX_DIST_CHUNK_MIN_BLOCKS", "4"))     # pieces shorter than this many block rows stay whole

    def chunk_rows0(p, k):
        """rows of piece p of panel k that travel in chunk 0 (all of them where the panel is not cut)"""
        m = geo.piece_rows(p, k)
        if nchunk < 2 or m == 0:
            return m
        k0 = (k // chunk_hold) * chunk_hold
        m0 = geo.piece_rows(p, k0)
        if m0 < chunk_min * nb:
            return m
        S = geo.row_off(p, geo.li0(p, k0)) + ((m0 // 2 + nb - 1) // nb) * nb         # local row of the cut
        s = S - geo.row_off(p, geo.li0(p, k))
        return min(max(s, min(m, nb)), m)

    def chunk_pieces(k, c):
        """(offset, count, world root) of the regions chunk c of panel k delivers; the diagonal region travels with chunk 0"""
        kr_, kc_ = k % Pr, k % Pc
        out = []
        for p in range(Pr):
            m, s = geo.piece_rows(p, k), chunk_rows0(p, k)
            r0, r1 = (0, s) if c == 0 else (s, m)
            if c == 0 and p == kr_:
                out.append((geo.piece_off(p), geo.dsz + r1 * geo.gld, p * Pc + kc_))
            elif r1 > r0:
                out.append((geo.piece_off(p) + geo.dsz + r0 * geo.gld, (r1 - r0) * geo.gld, p * Pc + kc_))
        return out

    def chunk_hi0(k):
        """one past the last LOCAL row (this process row) that chunk 0 of panel k solves / updates"""
        return geo.row_off(pr, geo.li0(pr, k)) + chunk_rows0(pr, k)

    def group_end(k):
        return min((k // q + 1) * q - 1, nblk - 1)

    def my_col_range(Ja, Jb):
        """(lc0, n, first J) of this rank's local block columns with global index in [Ja, Jb]."""
        Jb = min(Jb, nblk - 1)
        if Ja > Jb:
            return 0, 0, None
        lja = 0 if Ja <= pc else (Ja - pc + Pc - 1) // Pc
        ljb = (Jb - pc) // Pc if Jb >= pc else -1
        if ljb < lja:
            return 0, 0, None
        n = sum(geo.height(lj * Pc + pc) for lj in range(lja, ljb + 1))
        return lja * nb, n, lja * Pc + pc

    def update_cols(Ja, Jb, ks, below_diag=False, rows=None):
        """rows = (lo, hi): only the local rows in [lo, hi) (a row chunk of the panel; block-aligned)"""
        lc0, n, J0 = my_col_range(Ja, Jb)
        if n == 0:
            return
        li = geo.blocks_before(pr, Pr, J0 + (1 if below_diag else 0))   # first local block row with I >= J0 (> J0)
        lr0 = geo.row_off(pr, li)
        hi = geo.local_rows(pr)
        if rows is not None:
            lr0, hi = max(lr0, rows[0]), min(hi, rows[1])
        m = hi - lr0
        if m > 0:
            ops.update_multi(A, lr0, m, lc0, n, geo, [G[kk % R] for kk in ks], list(ks), below_diag)

    def split_cols(Ja, Jb, parts):
        """[Ja, Jb] cut into <= parts consecutive ranges of about equal trailing-update work (block column J costs ~ the
        number of block rows below it)."""
        Jb = min(Jb, nblk - 1)
        cols = list(range(Ja, Jb + 1))
        if not cols:
            return []
        wts = [nblk - J for J in cols]
        total, out, acc, start = float(sum(wts)), [], 0.0, 0
        for i, wgt in enumerate(wts):
            acc += wgt
            if acc >= total * (len(out) + 1) / parts or i == len(cols) - 1:
                out.append((cols[start], cols[i]))
                start = i + 1
        return [r for r in out if r[0] <= r[1]]

    bulk_recorded = set()
    pending = []                 # chunks of the last group's bulk update still to be issued (chunked mode)
    last_chunk_step = {}         # group end -> step at which its last chunk was issued
    ops.stream(MAIN)
    # FORK: every other stream starts behind this point of MAIN (behind the assembly queued there).  Costs nothing when the
    # rows are issued one by one, and is what lets the whole step be captured as one hipGraph (every stream of a capture must
    # branch off the capturing stream and rejoin it: the JOIN rows at the end).
    ops.record(EV_FORK)
    for s_ in ALL_SIDE_STREAMS:
        ops.stream(s_)
        ops.wait(EV_FORK)
    ops.stream(MAIN)
    ops.begin()
    ops.record(_ev2(E_DIAGREADY, 0))  # the assembly was queued on MAIN
    ops.record(_ev2(E_COLREADY, 0))
    for k in range(nblk):
        kr, kc = k % Pr, k % Pc
        g = G[k % R]
        w = geo.height(k)
        lr, lc = (k // Pr) * nb, (k // Pc) * nb
        holder = pc == kc
        owner = holder and pr == kr
        nxt = k + 1 < nblk
        r1, c1 = (k + 1) % Pr, (k + 1) % Pc        # process row of block row k+1 / column of the next diagonal owner
        h1 = geo.height(k + 1) if nxt else 0
        early_off = geo.piece_off(r1) + geo.dsz      # L[k+1, k] is the first block of piece r1
        if at_step is not None:
            at_step(geo, k)

        def wait_free(kk=k):
            """everything that read the buffer of step kk one ring ago is done (near updates + group-end updates of that panel's
            group, the copy into L, the panel stream's solves)"""
            old = kk - R
            if old >= 0:
                ge = group_end(old)
                ops.wait(_ev2(E_UPD, last_chunk_step.get(ge, ge)))
                if ge in bulk_recorded:
                    ops.wait(_ev2(E_BULK, ge))
                ops.wait(_ev2(E_STORED, old))
                ops.wait(_ev2(E_PANELDONE, old))

        # ---- diagonal chain -------------------------------------------------------------------------------------
        if owner:
            ops.stream(PANEL)
            ops.wait(_ev2(E_DIAGREADY, k))
            if not staged:
                wait_free()
                ops.diag_factor(A, lr, lc, w, g, geo.piece_off(kr), nb, k * nb, geo.n)
            else:
                if k == 0:                                               # (later blocks were staged one step earlier, below)
                    ops.diag_stage(A, lr, lc, w, g, geo.piece_off(kr), nb)
                ops.diag_factor_staged(A, lr, lc, w, g, geo.piece_off(kr), nb, k * nb, geo.n)
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
            if hoist_inv:
                ops.panel_inv(g, geo.piece_off(kr), nb, w)                # needs the diagonal block only: ahead of the column
            lr0, m = geo.row_off(pr, geo.li0(pr, k)), geo.piece_rows(pr, k)
            s0 = chunk_rows0(pr, k)                                      # rows of this rank's piece in chunk 0
            two = bool(chunk_pieces(k, 1))                               # the panel travels in two chunks (all ranks agree)
            # chunk 0's rows have column k's update through panel k-1 once THAT panel's chunk-0 update ran (E_COLREADY0) -- if it
            # covered them: after a re-centring of the cut it did not, and the solve waits for the whole column
            covered = k >= 1 and bool(chunk_pieces(k - 1, 1)) and lr0 + s0 <= chunk_hi0(k - 1)
            ops.wait(_ev2(E_COLREADY0 if covered else E_COLREADY, k))
            roff = geo.piece_off(pr) + geo.dsz
            dslot = (k // Pr) if owner else None                         # the owner keeps the block's explicit inverse
            keep_late = owner and staged and hoist_inv                    # ... behind E_PIECE (diag_store) instead of ahead of the solve
            # (full-width blocks with the inverse at hand: the solved rows go to the packed buffer only and are copied back into
            # the local matrix BEHIND the events that release the two broadcasts -- the copies are not the chain's business)
            late = late_copyback and hoist_inv and w == nb and nb > TILE and hasattr(ops, "panel_copyback")
            pk = dict(prepared=True, copy_back=not late) if hoist_inv else {}
            ds = None if keep_late else dslot
            e1 = 0
            if nxt and pr == r1:                                         # block row k+1 first: the next diagonal needs it
                e1 = h1
                ops.panel_trsm(A, lr0, h1, lc, w, g, geo.piece_off(kr), roff, nb, **pk)
                ops.record(_ev2(E_EARLYSOLVED, k))
            c0 = s0 if two else m                                        # chunk 0 ends here (>= e1: it holds the first block row)
            ops.panel_trsm(A, lr0 + e1, c0 - e1, lc, w, g, geo.piece_off(kr), roff + e1 * geo.gld, nb, ds, **pk)
            ops.record(_ev2(E_PIECE0, k))
            if two:
                if covered:
                    ops.wait(_ev2(E_COLREADY, k))
                ops.panel_trsm(A, lr0 + c0, m - c0, lc, w, g, geo.piece_off(kr), roff + c0 * geo.gld, nb, **pk)
            ops.record(_ev2(E_PIECE, k))
            while deferred_bulk:                                         # the previous group's bulk update, held back until here
                bk, bks = deferred_bulk.pop(0)
                ops.stream(bulk_stream)
                if bulk_stream != MAIN:
                    ops.wait(_ev2(E_COLREADY, 0))
                    ops.wait(_ev2(E_ARRIVED, bk))
                    ops.wait(_ev2(E_PIECE, k))
                update_cols(bk + 3 + q, nblk - 1, bks)
                ops.record(_ev2(E_BULK, bk))
                bulk_recorded.add(bk)
                ops.stream(PANEL)
            if owner and staged:
                ops.diag_store(A, lr, lc, w, g, geo.piece_off(kr), nb, dslot if keep_late else None)
            if late and m > 0:
                ops.panel_copyback(A, lr0, m, lc, w, g, roff, nb)
        if nxt and pr == r1:
            ops.stream(COMM)
            if holder:
                ops.wait(_ev2(E_EARLYSOLVED, k))
            else:
                wait_free()
            if E is not None:
                comm.bcast_grp2(g, early_off, E[k % R], 0, h1 * geo.gld, kc, ROW)   # ... into the early buffer of this step
            else:
                comm.bcast_grp(g, early_off, h1 * geo.gld, kc, ROW)      # L[k+1, k] along the process row of block row k+1
            ops.record(_ev2(E_EARLY, k))
            if pc == c1:                                                 # owner of the next diagonal block
                ops.stream(PANEL)
                if not staged:
                    ops.wait(_ev2(E_EARLY, k))
                if k >= 1:
                    ops.wait(_ev2(E_COL2, k + 1))                        # contributions of the panels before k
                else:
                    # k = 0: nothing else orders this stream behind the ASSEMBLY of A (queued on MAIN); without it the
                    # update could land on the block before the fill wrote it and be overwritten -- seen on the first
                    # step of a fresh process only, when the fill kernel's first launch is slow (1 run in 12)
                    ops.wait(_ev2(E_COLREADY, 0))
                src, soff = (E[k % R], 0) if E is not None else (g, early_off)
                if staged:
                    g1, doff1 = G[(k + 1) % R], geo.piece_off(r1)
                    wait_free(k + 1)
                    ops.diag_stage(A, ((k + 1) // Pr) * nb, ((k + 1) // Pc) * nb, h1, g1, doff1, nb)
                    ops.wait(_ev2(E_EARLY, k))
                    ops.diag_update(g1, doff1, h1, src, soff, w, nb)
                else:
                    ops.update(A, ((k + 1) // Pr) * nb, h1, ((k + 1) // Pc) * nb, h1, src, soff, soff, w, nb)
                ops.record(_ev2(E_DIAGREADY, k + 1))
        ops.stream(PANEL)
        ops.record(_ev2(E_PANELDONE, k))
        # ---- panel chain ----------------------------------------------------------------------------------------
        ops.stream(COMM)
        two = bool(chunk_pieces(k, 1))
        if holder:
            ops.wait(_ev2(E_PIECE0 if two else E_PIECE, k))
        else:
            wait_free()
        # Ranks of process row r1 that are no column holders already HAVE L[k+1, k] (row broadcast above) and their PANEL
        # stream may be reading it (early diagonal update) while the panel broadcast lands the same bytes on it again:
        # order the overwrite behind that read instead of relying on the bytes being identical.
        if nxt and pr == r1 and pc == c1 and not holder and E is None:
            ops.wait(_ev2(E_DIAGREADY, k + 1))
        at_chunk = getattr(comm, "at_chunk", None)                        # (measurement communicators: which chunk comes next)
        if two:
            if at_chunk is not None:
                at_chunk(0, False)
            comm.panel_bcast(g, chunk_pieces(k, 0))                      # chunk 0 of every piece (+ the diagonal region) ...
            ops.record(_ev2(E_ARR0, k))
            if holder:
                ops.wait(_ev2(E_PIECE, k))
            if at_chunk is not None:
                at_chunk(1, True)
            comm.panel_bcast(g, chunk_pieces(k, 1))                      # ... and the rest, one stage behind
        else:
            if at_chunk is not None:
                at_chunk(0, True)
            comm.panel_bcast(g, geo.pieces(k))                           # every piece to every rank, all links
            ops.record(_ev2(E_ARR0, k))
        ops.record(_ev2(E_ARRIVED, k))
        ops.stream(BACK)
        ops.wait(_ev2(E_ARRIVED, k))
        if L is not None:
            col = (k % window) * nb if window else k * nb
            if window and k >= window:
                ops.wait(_ev2(E_IVAR, group_end(k - window)))            # the evaluation step that read this column slot
            ops.unpack_diag(g, geo.piece_off(kr), w, nb, L, k * nb, col if window else None)
            for p in range(Pr):
                m = geo.piece_rows(p, k)
                if m > 0:
                    ops.unpack_rows(g, geo.piece_off(p) + geo.dsz, m, w, nb, L, p + geo.li0(p, k) * Pr, Pr, col)
        ops.record(_ev2(E_STORED, k))
        if on_stored is not None:
            on_stored(k)
        # ---- trailing updates -----------------------------------------------------------------------------------
        ops.stream(MAIN)
        g0 = (k // q) * q
        if two and nxt and pc == c1:
            # this rank solves panel k+1: its column k+1 gets panel k chunk by chunk (the B operand -- block row k+1 of the panel
            # -- is the first block row of its piece and travels with chunk 0)
            cut = chunk_hi0(k)
            ops.wait(_ev2(E_ARR0, k))
            update_cols(k + 1, k + 1, [k], below_diag=True, rows=(0, cut))
            ops.record(_ev2(E_COLREADY0, k + 1))
            ops.wait(_ev2(E_ARRIVED, k))
            update_cols(k + 1, k + 1, [k], below_diag=True, rows=(cut, geo.local_rows(pr)))
        else:
            ops.wait(_ev2(E_ARRIVED, k))
            update_cols(k + 1, k + 1, [k], below_diag=True)              # look-ahead: releases panel k+1's solve
        if nxt and pc == c1:
            ops.record(_ev2(E_COLREADY, k + 1))
        update_cols(k + 2, k + 2, list(range(g0, k + 1)))                # two ahead: up to date through panel k
        if k + 2 < nblk and (k + 2) % Pc == pc:
            ops.record(_ev2(E_COL2, k + 2))
        if pending and k != group_end(k):                                # chunked bulk of the previous group, one per step
            Ja, Jb, pks, pge = pending.pop(0)
            update_cols(Ja, Jb, pks)
            last_chunk_step[pge] = k
        if k == group_end(k):
            while pending:                                               # (only a short last group leaves any)
                Ja, Jb, pks, pge = pending.pop(0)
                update_cols(Ja, Jb, pks)
                last_chunk_step[pge] = k
            ks = list(range(g0, k + 1))
            if k + 3 < nblk:
                # the block columns that turn "near" during the next group: here, ahead of everything else.  They were last
                # written by the previous group's bulk update (other stream).
                prev = g0 - 1
                if prev in bulk_recorded:
                    ops.wait(_ev2(E_BULK, prev))
                update_cols(k + 3, k + 2 + q, ks)
            if k + 3 + q < nblk and chunked:
                pending = [(Ja, Jb, ks, k) for Ja, Jb in split_cols(k + 3 + q, nblk - 1, q)]
                Ja, Jb, pks, pge = pending.pop(0)
                update_cols(Ja, Jb, pks)
                last_chunk_step[k] = k
            elif k + 3 + q < nblk and gate_level >= 2 and bulk_stream != MAIN and nxt and pc == c1:
                deferred_bulk.append((k, ks))                            # this rank holds column k+1: issued behind that step's solve
            elif k + 3 + q < nblk:
                ops.stream(bulk_stream)
                if bulk_stream != MAIN:
                    ops.wait(_ev2(E_COLREADY, 0))                        # this rank's own assembly of A (queued on MAIN)
                    ops.wait(_ev2(E_ARRIVED, k))                         # COMM is in order: implies the group's earlier panels
                    if gate_bulk and nxt and pc == c1:
                        # this rank's update of column k+1 gates the NEXT panel's solve -- the chain across ranks; a chip-filling
                        # bulk launch issued at the same moment halves its rate (paced replay: the steps that follow a group end
                        # took twice as long as the others).  The bulk update starts behind it.
                        ops.wait(_ev2(E_COLREADY, k + 1))
                update_cols(k + 3 + q, nblk - 1, ks)
                ops.record(_ev2(E_BULK, k))
                bulk_recorded.add(k)
                ops.stream(MAIN)
        if k == nblk - 1:
            while pending:
                Ja, Jb, pks, pge = pending.pop(0)
                update_cols(Ja, Jb, pks)
                last_chunk_step[pge] = k
        ops.record(_ev2(E_UPD, k))
    for i_, s_ in enumerate(ALL_SIDE_STREAMS):                           # JOIN
        ops.stream(s_)
        ops.record(EV_JOIN0 + i_)
    ops.stream(MAIN)
    for i_ in range(len(ALL_SIDE_STREAMS)):
        ops.wait(EV_JOIN0 + i_)


def dist2_restream_enqueue(ops, comm, geo, A, G, L, on_stored=None, agg=None, window=0):
    """The FINISHED block-cyclic factor A once more to every rank, panel by panel -- the evaluation phase of a process that keeps
    no replica of the factor (round 5; SURVEY 8e(1) "above N = 16384 keep L distributed"; the role of the fork inside
    GP.evaluateVariance, gp.py:244-258).  Per panel k: the holder column packs its pieces out of the local matrix (the owner adds
    the diagonal block and its leaf inverses), the panel broadcast delivers every piece to every rank, the pieces are copied into
    L -- a WINDOW of `window` block columns consumed group by group by `on_stored` (streamed_ivar_hook: one right-looking solve
    step of the rank's slice of K(X, Z) per group, E_IVAR releases the slot), or with window = 0 a full-size matrix (the replica
    assembled on demand for the entry points that need a dense factor).  No arithmetic on the factor, no trailing updates: per
    rank N^2/2 doubles received, nothing resident but A / W, the window and the ring of packed buffers.  Pure enqueue."""
    nb, Pr, Pc, pr, pc = geo.nb, geo.Pr, geo.Pc, geo.pr, geo.pc
    nblk = geo.nblk
    q = default_agg() if agg is None else int(agg)
    R = len(G)
    window = int(window)
    assert window == 0 or (L is not None and on_stored is not None and window % q == 0 and window >= 2 * q)
    at_step = getattr(comm, "at_step", None)
    ops.stream(MAIN)
    ops.record(EV_FORK)
    for s_ in ALL_SIDE_STREAMS:
        ops.stream(s_)
        ops.wait(EV_FORK)
    for k in range(nblk):
        kr, kc = k % Pr, k % Pc
        g = G[k % R]
        w = geo.height(k)
        lr, lc = (k // Pr) * nb, (k // Pc) * nb
        holder = pc == kc
        if at_step is not None:
            at_step(geo, k)
        if holder:
            ops.stream(PANEL)
            if k >= R:
                ops.wait(_ev2(E_STORED, k - R))                           # the buffer's previous panel has been copied out
            if pr == kr:
                ops.diag_pack(A, lr, lc, w, g, geo.piece_off(kr), nb)
            lr0, m = geo.row_off(pr, geo.li0(pr, k)), geo.piece_rows(pr, k)
            if m > 0:
                ops.panel_pack(A, lr0, m, lc, w, g, geo.piece_off(pr) + geo.dsz, nb)
            ops.record(_ev2(E_PIECE, k))
        ops.stream(COMM)
        if holder:
            ops.wait(_ev2(E_PIECE, k))
        elif k >= R:
            ops.wait(_ev2(E_STORED, k - R))
        comm.panel_bcast(g, geo.pieces(k))
        ops.record(_ev2(E_ARRIVED, k))
        ops.stream(BACK)
        ops.wait(_ev2(E_ARRIVED, k))
        if L is not None:
            col = (k % window) * nb if window else k * nb
            if window and k >= window:
                ops.wait(_ev2(E_IVAR, min(((k - window) // q + 1) * q - 1, nblk - 1)))   # the step that read this column slot
            ops.unpack_diag(g, geo.piece_off(kr), w, nb, L, k * nb, col if window else None)
            for p in range(Pr):
                m = geo.piece_rows(p, k)
                if m > 0:
                    ops.unpack_rows(g, geo.piece_off(p) + geo.dsz, m, w, nb, L, p + geo.li0(p, k) * Pr, Pr, col)
        ops.record(_ev2(E_STORED, k))
        if on_stored is not None:
            on_stored(k)
    for i_, s_ in enumerate(ALL_SIDE_STREAMS):                           # JOIN
        ops.stream(s_)
        ops.record(EV_JOIN0 + i_)
    ops.stream(MAIN)
    for i_ in range(len(ALL_SIDE_STREAMS)):
        ops.wait(EV_JOIN0 + i_)


def dist2_potrs(ops, comm, geo, A, yv, acc_r, acc_c, out, skip_forward=False):
    """alpha = K^-1 y on the block-cyclic factor A: block forward substitution (partial sums reduced along the process
    row of the diagonal owner, the solved block broadcast down its process column), then the transposed sweep with the
    roles of rows and columns exchanged; the blocks of alpha (one per diagonal owner) are assembled on every rank by one
    ncclAllReduce.  yv: device vector (padded N) holding y on every rank -- overwritten; acc_r / acc_c: scratch vectors
    of local_rows / local_cols doubles; out: device vector (padded N) that receives alpha everywhere.  Pure enqueue.
    skip_forward: yv holds w = L^-1 y on every rank already (the streamed evaluation carried the forward substitution along,
    streamed_ivar_hook) -- only the transposed sweep runs: half the block steps, half the collectives."""
    nb, Pr, Pc, pr, pc = geo.nb, geo.Pr, geo.Pc, geo.pr, geo.pc
    ops.stream(MAIN)
    ops.vec_op(acc_r, 0, None, 0, max(geo.local_rows(pr), 1), 2)
    ops.vec_op(acc_c, 0, None, 0, max(geo.local_cols(pc), 1), 2)
    ops.vec_op(out, 0, None, 0, geo.np, 2)
    for k in (() if skip_forward else range(geo.nblk)):         # L w = y  (skip_forward: yv already holds w on every rank)
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


def dist2_logdet_enqueue(ops, geo, A, scal):
    ops.stream(MAIN)
    ops.vec_op(scal, 0, None, 0, 1, 2)
    for k in range(geo.nblk):
        if k % geo.Pr == geo.pr and k % geo.Pc == geo.pc:
            ops.logdet_acc(A, (k // geo.Pr) * geo.nb, (k // geo.Pc) * geo.nb, geo.height(k), geo.height(k), scal)


def dist2_logdet(ops, comm, geo, A, scal):
    """log det K = 2 sum log L_ii: every diagonal owner adds its blocks, one ncclAllReduce of a scalar."""
    dist2_logdet_enqueue(ops, geo, A, scal)
    return float(comm.allreduce_host(np.array([ops.vec_to_host(scal, 1)[0]]))[0])


def streamed_ivar_hook(ops, geo, L, B, q, window=0, stream=None, fwd=None):
    """`on_stored` hook of dist2_potrf_enqueue for the STREAMED evaluation: one right-looking solve step of B = K(X, Z_local)
    per GROUP of stored panels (K = q nb updates), behind the copy of the group's last panel into L.  With window > 0, L is the
    window of block columns described at dist2_potrf_enqueue: the step reads the group at its column slot and records E_IVAR,
    which releases the slot; `fwd` (device vector holding y) takes the same step of the forward substitution -- no rank ever
    holds the whole factor (above the size where a replica is cheap,
    "keep L distributed": every panel reaches every rank anyway for the trailing update, the evaluation consumes it then)."""
    last = geo.nblk - 1
    if stream is None:
        stream = EVAL

    def hook(k):
        if k % q == q - 1 or k == last:
            if stream != BACK:
                ops.stream(stream)
                ops.wait(_ev2(E_STORED, k))
            else:
                ops.stream(BACK)                            # same stream as the copies: already ordered
            k0 = (k // q) * q
            c0 = (k0 % window) * geo.nb if window else None
            ops.ivar_group(L, k0, k, geo.nb, B, c0)
            if fwd is not None:           # the forward substitution L w = y rides along (every rank: the complete w)
                ops.fwd_group(L, k0, k, geo.nb, fwd, c0)
            if window:
                ops.record(_ev2(E_IVAR, k))
    return hook


class DistFitIvar2D:
    """bench.py's multi-GPU step on the 2-D block-cyclic layout: distributed fit (local assembly + dist2_potrf), alpha
    by distributed substitution, logdet / y^T alpha through all-reduce, IVAR with the evaluation points sharded over the
    ranks (streamed underneath the factorisation from 4 ranks).

    Memory per rank: the local share of the working matrix (N^2 / W), the ring of packed panel buffers (2 agg x N x nb), the
    rank's slab of the cross matrix (N x M / W) -- and the finished factor in one of two forms:
      streamed evaluation (default from 4 ranks AND N >= 98304; round 3: from 4 ranks)  a WINDOW of 2 agg block columns (N x 2 agg nb: 1.07 GB at C4): every panel
          reaches every rank for the trailing update anyway, the evaluation's solve step consumes it right then, nothing reads
          it later.  The factor itself stays distributed (block-cyclic A); N per node is bounded by A / W + B, not by N^2.
      evaluation after the fit (the default below N = 98304), C5, the class API   a REPLICATED copy (N^2: 8.6 GB at C4, 34 GB at C5 of the 288 GB), against
          which the evaluation / the gradient slabs run with no exchange."""

    def __init__(self, ctx, comm, spec, Xh, yh, Zh, noise, nb=None, ops=None, streamed=None, grid=None, agg=None,
                 fit_only=False, replicate=None, cyclic=False):
        self.ctx, self.comm, self.spec = ctx, comm, spec
        self.ops = ops or DeviceOps2D(ctx)
        Pr, Pc = grid or choose_grid(comm.world)
        comm.set_grid(Pr, Pc)
        self.n, self.noise = Xh.shape[0], float(noise)
        # Evaluation streamed underneath the factorisation (against a window of the factor: no N x N replica) or after it
        # (against the rank's replica).  Round 3 streamed from 4 ranks because a rank's BUSY time was lower that way; the paced
        # replay of round 4 says the opposite for the time of the GRID: the streamed solve's GEMMs slow the chain
        # across ranks (2 x 4 at C4: 121-133 ms streamed against 53 + 62 + 4 = 119 ms fit-then-evaluate; 2 x 2: 206 against 200).
        # Streaming remains the memory-saving form: default from N = 98304 (a replica of 77 GB), where a
        # rank's own work also dwarfs the chain.  GPX_DIST_STREAM_IVAR=0/1 overrides.
        env = os.environ.get("GPX_DIST_STREAM_IVAR")
        big = Xh.shape[0] >= 98304
        self.streamed = (comm.world >= 4 and big) if streamed is None else bool(streamed)
        if env is not None:
            self.streamed = env == "1"
        # (nb = None: default_nb -- decided from N, the world size and the evaluation schedule alone: the same on every rank)
        nb = default_nb(self.n, comm.world, self.streamed and Zh.shape[0] > 0) if nb is None else int(nb)
        self.nb = nb
        self.geo = Grid2D(self.n, nb, Pr, Pc, comm.rank)
        self.agg = default_agg(self.streamed and Zh.shape[0] > 0, comm.world, nb) if agg is None else int(agg)
        self.fit_only = bool(fit_only)       # replay: factorisation (+ streamed evaluation) only
        self.yh = np.ascontiguousarray(yh, dtype=np.float64)
        self.m = Zh.shape[0]
        self.X = self.ops.points(Xh)
        lo, hi = eval_slice(self.m, comm.rank, comm.world)
        self.Zloc = self.ops.points(Zh[lo:hi]) if hi > lo else None
        self.A = self.ops.alloc_local(self.geo)
        self.G = [self.ops.alloc_buf(self.geo) for _ in range(ring_size(self.agg))]
        # early buffers (one per ring slot): the block row the next diagonal needs, received out of the panel buffer's way
        self.E = ([self.ops.alloc_vec(nb * self.geo.gld) for _ in self.G]
                  if os.environ.get("GPX_DIST_EARLY_BUF", "1") == "1" and hasattr(comm, "bcast_grp2") else None)
        # The finished factor: with the STREAMED evaluation nothing reads a panel after its group's solve step, so the rank
        # keeps a window of two groups of block columns (N x 2 agg nb) instead of an N x N replica (`replicate`
        # forces the replica; the C5 gradient needs it).
        self.replicate = (not self.streamed) if replicate is None else bool(replicate)
        if not self.streamed:
            self.replicate = True
        self.window = 0 if self.replicate else 2 * self.agg
        # cyclic (round 5, the class API's distributed-factor mode): the finished factor stays block-cyclic in A -- no replica,
        # no window until an evaluation asks for one; alpha / log det by distributed substitution, evaluation by re-streaming the
        # panels (cyclic_posterior), a dense replica only on demand (assemble_dense)
        self.cyclic_only = bool(cyclic)
        self.generation = 0              # fits of this runner so far: a CyclicFactor knows which one it belongs to
        self._win = None
        if self.cyclic_only:
            self.replicate, self.window, self.L = False, 0, None
        elif self.window:
            self.L = self.ops.alloc_window(self.n, self.window * nb)
        else:
            self.L = self.ops.alloc_matrix(self.n)
        # The forward substitution L w = y rides along the streamed evaluation when EVERY rank streams (all have evaluation
        # points): each rank then ends the factorisation with the complete w and the distributed substitution keeps only its
        # backward sweep (half the block steps and collectives).  Decided from (M, world) alone: the same on every rank.
        self.fused_fwd = bool(self.window) and Zh.shape[0] >= comm.world
        self.yv = self.ops.alloc_vec(self.geo.np)
        self.y0 = self.ops.alloc_vec(self.geo.np)
        ypad = np.zeros(self.geo.np)
        ypad[:self.n] = self.yh
        self.ops.vec_from_host(self.y0, ypad)
        self.alpha = self.ops.alloc_vec(self.geo.np)
        self.acc_r = self.ops.alloc_vec(self.geo.local_rows(self.geo.pr))
        self.acc_c = self.ops.alloc_vec(self.geo.local_cols(self.geo.pc))
        self.scal = self.ops.alloc_vec(8)
        self.B = self.ops.alloc_cross(self.n, hi - lo) if (self.streamed and hi > lo) else None
        if hasattr(self.ops, "reserve"):
            self.ops.reserve(nb, self.agg, (hi - lo) if self.B is not None else 0)
        # recorded programs (device ops + a recordable communicator only; `force_interpret` interprets every step)
        self.programs = None
        self.recordable = type(self.ops) is DeviceOps2D and getattr(comm, "recordable", False)
        # hipGraph replay of the recorded programs: EXPERIMENTAL, off (scripts/graph_capture_bisect.py sets `use_graph`).  The programs fork every stream
        # off MAIN and join it again, so they are capturable in principle (gpx_program_capture), and small ones are (four
        # panels: 68 nodes, one hipGraphLaunch per step) -- but on ROCm 7.2 hipStreamEndCapture SEGFAULTS on the full loop's
        # cross-stream event pattern: first on a side stream waiting for an event recorded on that same stream (now dropped
        # by DeviceOps2D.wait), then on the panel stream's buffer-reuse waits from the ninth panel on
        # (scripts/graph_capture_bisect.py cuts the recorded program to the first failing row; scripts/graph_check.hip shows the
        # elementary patterns work).  RCCL under capture is unvalidated on top of that.  Row-by-row replay stays the product path.
        self.use_graph = False
        self.host_ms = {}
        self._runs = {}

    # the three asynchronous phases of a step, written once, either executed directly or recorded
    def _hook(self):
        if self.B is None:
            return None
        return streamed_ivar_hook(self.ops, self.geo, self.L, self.B, self.agg, self.window,
                                  fwd=self.yv if self.fused_fwd else None)

    def _enqueue_factor(self):
        hook = self._hook()
        L = None if (self.window and hook is None) else self.L       # a rank without evaluation points keeps no window
        if getattr(self, "cyclic_only", False):                      # the factor stays block-cyclic in A: nothing is copied
            L, hook = None, None
        dist2_potrf_enqueue(self.ops, self.comm, self.geo, self.A, self.G, L=L, on_stored=hook, agg=self.agg,
                            window=self.window if hook is not None else 0, E=self.E)

    def _enqueue_solve(self):
        ops, geo = self.ops, self.geo
        ops.stream(MAIN)
        if not self.fused_fwd:
            ops.vec_op(self.yv, 0, self.y0, 0, geo.np, 0)
        dist2_potrs(ops, self.comm, geo, self.A, self.yv, self.acc_r, self.acc_c, self.alpha, skip_forward=self.fused_fwd)
        dist2_logdet_enqueue(ops, geo, self.A, self.scal)

    def _record(self):
        progs = {}
        for name, fn in (("factor", self._enqueue_factor), ("solve", self._enqueue_solve)):
            if name == "solve" and self.fit_only:
                continue
            prog = Program()
            self.ops.prog = self.comm.prog = prog
            try:
                fn()
            finally:
                self.ops.prog = self.comm.prog = None
            progs[name] = prog
        self.programs = progs

    def _run(self, name, fn):
        if not self.recordable:
            fn()
            return
        if self.programs is None:
            self._record()
        prog = self.programs[name]
        n = self._runs.get(name, 0)
        self._runs[name] = n + 1
        if self.use_graph and n >= 1 and not getattr(self, "force_interpret", False):   # (profiled steps need the launches)
            if prog.graph is None:      # second use: scratch buffers and kernel attributes exist, nothing allocates any more
                self.ctx.sync()
                prog.capture(self.ctx)
            self.host_ms[name] = prog.launch(self.ctx)
        else:
            self.host_ms[name] = prog.run(self.ctx)

    def fit(self):
        """Distributed assembly + factorisation (+ the streamed evaluation solve when enabled); raises NotPositiveDefinite."""
        ops, comm, geo = self.ops, self.comm, self.geo
        ops.stream(MAIN)
        ops.kfill_local(self.spec, self.X, self.A, self.noise, geo)
        if self.B is not None:
            ops.stream(BACK)
            ops.cross_fill(self.spec, self.X, self.Zloc, self.B)   # independent of the factorisation
            ops.record(EV_PRE)          # the program forks every stream off MAIN: MAIN carries the cross fill's completion
            ops.stream(MAIN)
            ops.wait(EV_PRE)
        if self.fused_fwd:
            ops.stream(MAIN)                                       # (the program forks every stream off MAIN behind this)
            ops.vec_op(self.yv, 0, self.y0, 0, geo.np, 0)
        self.generation += 1
        self._run("factor", self._enqueue_factor)
        info = dist2_potrf_finish(ops, comm, None if (self.window or getattr(self, "cyclic_only", False)) else self.L)
        if info:
            from ._lib import NotPositiveDefinite
            raise NotPositiveDefinite(info)

    def solve(self):
        """alpha (host, on every rank) and the log marginal likelihood.  With a replicated factor on every rank (evaluation
        after the fit, C5) they come from the replica -- the single-GPU sweeps, no exchange: the distributed substitution is a
        chain of 2 N / nb block steps with two collectives each (11 ms at C4 before any communication; the local sweeps:
        2.3 ms).  Without a replica (streamed evaluation against a window of the factor) they come from the block-cyclic
        factor by distributed substitution; GPX_DIST_SOLVE=dist forces that form everywhere."""
        ops, comm = self.ops, self.comm
        if self.replicate and os.environ.get("GPX_DIST_SOLVE", "local") != "dist":
            ops.stream(MAIN)
            ops.replica_solve(self.L, self.y0, self.alpha)
            logdet = ops.replica_logdet(self.L)
            alpha = ops.vec_to_host(self.alpha, self.n)
            ll = -0.5 * hdot(self.yh, alpha) - 0.5 * logdet - self.n / 2.0 * np.log(2 * np.pi)
            return ll, alpha
        self._run("solve", self._enqueue_solve)
        logdet = float(comm.allreduce_host(np.array([ops.vec_to_host(self.scal, 1)[0]]))[0])
        alpha = ops.vec_to_host(self.alpha, self.n)
        ll = -0.5 * hdot(self.yh, alpha) - 0.5 * logdet - self.n / 2.0 * np.log(2 * np.pi)
        return ll, alpha

    def step(self):
        ops, comm = self.ops, self.comm
        self.fit()
        if self.fit_only:
            part = float(np.sum(ops.variances(self.spec, self.Zloc, self.B, self.n))) if self.B is not None else 0.0
            return 0.0, part
        ll, _ = self.solve()
        part = 0.0
        if self.B is not None:     # the solve finished with the last panel (dist2_potrf synchronised every stream)
            part = float(np.sum(ops.variances(self.spec, self.Zloc, self.B, self.n)))
        elif self.Zloc is not None:
            part = float(np.sum(ops.posterior_var(self.spec, self.L, self.X, self.Zloc)))
        iv = abs(ordered_sum(comm.allgather(np.array([part]))[:, 0]) / max(self.m, 1))
        return ll, iv


    # ---- round 5: the block-cyclic factor as THE factor (no replica) ------------------------------------------------------------
    def solve_for(self, y):
        """(alpha = K^-1 y on the host, log det K) from the block-cyclic factor by distributed substitution (dist2_potrs) --
        identical on every rank (the blocks of alpha are assembled by collectives)."""
        self.yh = np.ascontiguousarray(y, dtype=np.float64)
        ypad = np.zeros(self.geo.np)
        ypad[:self.n] = self.yh
        self.ops.vec_from_host(self.y0, ypad)
        self._run("solve", self._enqueue_solve)
        logdet = float(self.comm.allreduce_host(np.array([self.ops.vec_to_host(self.scal, 1)[0]]))[0])
        return self.ops.vec_to_host(self.alpha, self.n), logdet

    def resident_bytes(self):
        """what this rank holds for the factor: its share of the matrix, the ring of packed buffers (+ early buffers), the window"""
        geo = self.geo
        b = 8 * geo.local_rows(geo.pr) * geo.local_cols(geo.pc) + 8 * len(self.G) * geo.buf_elems()
        if self.E is not None:
            b += 8 * len(self.E) * self.nb * geo.gld
        if self._win is not None:
            b += 8 * geo.np * 2 * self.agg * self.nb
        if self.L is not None:
            b += 8 * geo.np * (self.window * self.nb if self.window else geo.np)
        return int(b)

    def cyclic_posterior(self, Zloc, alpha=None, want_mean=True, want_var=True):
        """(mean, signed variance) at this rank's slice `Zloc` (host, may be empty) of the evaluation points against the
        block-cyclic factor: K(X, Zloc) is solved group by group against a WINDOW of the factor while the panels are re-streamed
        (dist2_restream_enqueue).  A COLLECTIVE when want_var: every rank must call it, also with an empty slice."""
        ops, comm, geo = self.ops, self.comm, self.geo
        mloc = int(Zloc.shape[0])
        Z = ops.points(Zloc) if mloc else None
        mean = var = None
        if want_mean:
            mean = ops.cross_mean(self.spec, self.X, Z, alpha) if mloc else np.zeros(0)
        if want_var:
            q, B, hook, win = self.agg, None, None, None
            if mloc:
                B = ops.alloc_cross(self.n, mloc)
                if hasattr(ops, "reserve"):
                    ops.reserve(self.nb, q, mloc)
                ops.stream(MAIN)
                ops.cross_fill(self.spec, self.X, Z, B)
                if self._win is None:
                    self._win = ops.alloc_window(self.n, 2 * q * self.nb)
                win = self._win
                hook = streamed_ivar_hook(ops, geo, win, B, q, 2 * q)
            dist2_restream_enqueue(ops, comm, geo, self.A, self.G, win, on_stored=hook, agg=q, window=2 * q if mloc else 0)
            ops.info()                                   # synchronises every stream
            var = ops.variances(self.spec, Z, B, self.n) if mloc else np.zeros(0)
        return mean, var

    def assemble_dense(self):
        """A replica of the factor assembled on demand (a COLLECTIVE: the panels are re-streamed into a full-size matrix) for the
        entry points that need a dense factor on every rank (greedy designs, the gradient slabs, compvar = 2, point derivatives).
        The returned matrix is an ordinary factored matrix and belongs to the caller."""
        ops = self.ops
        L = ops.alloc_matrix(self.n)
        ops.stream(MAIN)
        ops.begin()
        dist2_restream_enqueue(ops, self.comm, self.geo, self.A, self.G, L, on_stored=None, agg=self.agg, window=0)
        ops.info()
        ops.finish(L)
        return L


class CyclicFactor:
    """What Session.factor hands the class API in the distributed-factor mode (GPX_DIST_FACTOR=cyclic): the fit of one
    (kernel, nodes, nugget) whose factor lives block-cyclic on the ranks.  solve / posterior run distributed; dense() assembles a
    replica once for everything else.  The runner's matrix is reused by the NEXT fit of the same size: a factor that is asked
    after that re-fits itself first (same inputs, same result)."""
    is_cyclic = True

    def __init__(self, sess, run, spec, nodes, nugget):
        self.sess, self.run, self.spec = sess, run, spec
        self.nodes, self.nugget = nodes, nugget
        self.generation = run.generation
        self._dense = None

    def _current(self):
        if self.run.generation != self.generation:
            self.sess._refit(self)
        return self.run

    def solve(self, y):
        return self._current().solve_for(y)

    def posterior(self, Zloc, alpha, want_mean, want_var):
        return self._current().cyclic_posterior(Zloc, alpha, want_mean, want_var)

    def dense(self):
        if self._dense is None:
            self._dense = self._current().assemble_dense()
            self.sess.stats["dense_assembled"] = self.sess.stats.get("dense_assembled", 0) + 1
        return self._dense


# =====================================================================================================================
# BASELINE config C5 on N GPUs: hyper-parameter gradient of the log marginal likelihood + mutual-information design
# =====================================================================================================================
def dist_lml_grad(ctx, comm, spec, L, X, alpha, be=None, slabs=None, form=None):
    """Gradient of the log marginal likelihood w.r.t. (cl_0..cl_{d-1}, signalSize, noise [raw: the caller scales by 2 noise,
    gp.py:463-464]) with the TRACES SHARDED over the ranks (gp.py:444-466 builds an (N, N, d+2) array and needs all of
    K^-1).  Every rank holds the complete factor L (replicated by the distributed fit); rank r forms only the row slab
    [b_r, b_{r+1}) of K^-1 -- two triangular solves against the TRAILING factor L[b_r:, b_r:], no N x N inverse anywhere --
    and reduces its share of tr((alpha alpha^T - K^-1) dK/dtheta) (gpx_lml_grad_slab).  The slab boundaries equalise the work
    (~(N - b)^2 per row).  One exchange: d+2 partial sums per rank, added in rank order (deterministic).
    `be`: the device backend (gpexp_amd.device; the CPU tests pass a NumPy double with the same four functions)."""
    be = be or _dev
    n = X.shape[0]
    # A rank's share is cut further into `sub` slabs of equal work (the solves of a slab run against the trailing factor below
    # its FIRST row: one slab over all rows costs 2 N^3, many slabs approach 2 N^3 / 3 -- single GPU, N = 65536: 7.8 s with one
    # slab, 3.5 s with 16); about 16 slabs in total, at least one per rank.
    sub = max(1, 16 // comm.world) if slabs is None else max(1, int(slabs))
    # Round 5: the ROWS form (the default where the backend has it): rank r takes rows of L^-1 -- one right solve against the
    # leading block of the factor + one large lower SYRK per slab instead of the slab form's two solves with K = 1024 block
    # inverses (gpx_lml_grad_rows; the slab form measured 54 TF/s at C5, 0.69 of the MFMA roof).  The rank whose range ends at the
    # padded order allocates an N x N accumulator on top of its replica, so the ranks' memory needs differ: a rank that runs out
    # of memory must not leave the others waiting in the exchange below -- every rank reports whether its share
    # succeeded, and when ANY failed ALL fall back to the slab form, which needs no N x N buffer.
    rows_form = hasattr(be, "lml_grad_rows") and form != "slabs"
    sums = np.zeros(getattr(spec, 'nsums', spec.d + 2))
    if rows_form:
        ok, err = 1.0, None
        try:
            b = be.lml_grad_rows_bounds(n, comm.world, sub)     # one range of rows per rank, `sub` sub-slabs inside (one trace)
            if b[comm.rank + 1] > b[comm.rank]:
                sums = sums + be.lml_grad_rows(ctx, spec, L, X, alpha, b[comm.rank], b[comm.rank + 1], sub)
        except GpxError as e:
            if "hipMalloc" not in str(e) and "memory" not in str(e).lower():
                ok = -1.0          # not an allocation failure: re-raised below, after the other ranks have been told
            else:
                ok = 0.0
            err = e
        flags = comm.allgather(np.array([ok]))
        worst = min(float(f[0]) for f in flags)
        if worst < 0.0:
            raise err if err is not None else RuntimeError("dist_lml_grad: another rank failed in its share of the gradient")
        if worst == 0.0:
            if hasattr(ctx, "trim"):
                ctx.trim()
            rows_form = False
            sums = np.zeros(getattr(spec, 'nsums', spec.d + 2))
    if not rows_form:
        b = be.lml_grad_slab_bounds(n, comm.world * sub)
        for i in range(comm.rank * sub, (comm.rank + 1) * sub):
            if b[i + 1] > b[i]:
                sums = sums + be.lml_grad_slab(ctx, spec, L, X, alpha, b[i], b[i + 1])
    allsums = comm.allgather(sums)
    tot = np.zeros(getattr(spec, 'nsums', spec.d + 2))
    for r in range(comm.world):
        tot += allsums[r]
    return be.lml_grad_from_sums(spec, tot)


def mi_owner(m, s, world):
    """Rank whose slice of the M candidates holds index s (eval_slice)."""
    for r in range(world):
        lo, hi = eval_slice(m, r, world)
        if lo <= s < hi:
            return r
    raise ValueError("index outside the candidate set")


def merge_argmax(values, indices):
    """First-maximum rule across ranks: highest value, ties -> lowest global index (np.argmax semantics)."""
    best = None
    for v, i in zip(values, indices):
        if best is None or v > best[0] or (v == best[0] and i < best[1]):
            best = (float(v), int(i))
    return best


def dist_mi_greedy(ctx, comm, spec, cand_host, noise, nsel, start=0, be=None):
    """Greedy mutual-information design (experimentalDesign.py:259-285, 753-785) with the candidate SCORING sharded: the
    M x M inverse of the candidate covariance is down-dated by rows -- rank r owns the contiguous rows [lo, hi) of it and
    scores exactly those candidates; per pick the owner of the chosen row broadcasts it (M doubles, ncclBroadcast), every
    rank down-dates and scores its rows, and the ranks exchange one (ratio, index) pair each; the winner follows np.argmax
    (highest ratio, ties to the lowest index), so the picks are identical to the single-GPU gpx_mi_greedy.  The numerator
    chain (a Cholesky row per pick, O(M) each) is replicated.  Returns (indices, ratios)."""
    be = be or _dev
    m = cand_host.shape[0]
    lo, hi = eval_slice(m, comm.rank, comm.world)
    Cp = be.points(ctx, cand_host)
    st = be.MiState(ctx, spec, Cp, noise, nsel, start, lo, hi)
    row = be.alloc_vector(ctx, max(m, 1))
    picks, ratios = [int(start)], []
    for cur in range(int(nsel) - 1):
        s = picks[-1]
        st.row(cur, row)                                        # numerator chain everywhere; the owner stages its row P[s, :]
        comm.bcast_grp(row, 0, m, mi_owner(m, s, comm.world), WORLD)
        val, idx = st.score(cur, row)                           # down-date my rows, ratios of my candidates, local first-max
        pairs = comm.allgather(np.array([val, float(idx)]))
        best = merge_argmax(pairs[:, 0], pairs[:, 1].astype(np.int64))
        st.select(cur + 1, best[1])
        picks.append(best[1])
        ratios.append(best[0])
    return np.array(picks, dtype=np.int64), np.array(ratios)


class DistFitGrad2D(DistFitIvar2D):
    """BASELINE config C5 on the 2-D layout: distributed fit, alpha + log marginal likelihood, the hyper-parameter gradient
    with its traces sharded over the ranks (dist_lml_grad) and -- optionally -- a greedy MI design over `cand` candidates with
    the scoring sharded (dist_mi_greedy)."""

    def __init__(self, ctx, comm, spec, Xh, yh, noise, nb=None, cand=None, nsel=8, be=None, **kw):
        super().__init__(ctx, comm, spec, Xh, yh, np.zeros((0, Xh.shape[1])), noise, nb=nb, streamed=False, **kw)
        self.cand, self.nsel, self.be = cand, int(nsel), be
        self.times = {}

    def step(self):
        import time
        t0 = time.perf_counter()
        self.fit()
        ll, alpha = self.solve()
        t1 = time.perf_counter()
        grad = dist_lml_grad(self.ctx, self.comm, self.spec, self.L, self.X, alpha, be=self.be)
        t2 = time.perf_counter()
        picks = None
        if self.cand is not None:
            picks, _ = dist_mi_greedy(self.ctx, self.comm, self.spec, self.cand, self.noise, self.nsel, be=self.be)
        t3 = time.perf_counter()
        self.times = {"fit_ms": 1e3 * (t1 - t0), "lml_grad_ms": 1e3 * (t2 - t1), "mi_ms": 1e3 * (t3 - t2)}
        return ll, grad, picks


# =====================================================================================================================
# The class API under a multi-process launch
# =====================================================================================================================
# Role replaced: the fork-and-gather INSIDE GP.evaluateVariance (gp.py:244-258 -> parallel_utilities.py:26-80) -- in the
# reference the parallel backend sits behind the class API, so it does here.  Contract: SPMD.  Every rank runs the same
# script on the same host data (one process per GPU, `python -m torch.distributed.run --nproc-per-node 8 demo.py`) and every
# rank gets the same arrays back, bit for bit:
#   fit        GP.train / addNodesAndComputeCovariance / computeLogLike / loglikeParams and every cost function that refits:
#              2-D block-cyclic distributed assembly + factorisation (DistFitIvar2D.fit); every panel reaches every rank for
#              the trailing update anyway, so every rank also assembles a REPLICA of the finished factor (N^2 doubles: 8.6 GB
#              at C4, 34 GB at C5 of the 288 GB) -- the object the class API keeps as its "precision matrix" (gp.py:181).
#              coeff / log det then come from the replica by the single-GPU sweeps (no exchange, identical on every rank).
#   evaluate   GP.evaluate(compvar=0/1) / evaluateVariance / costFunctionGP_IVAR.evaluate / greedyIVARStep: the evaluation
#              (candidate) points are split in contiguous slices (eval_slice: the chunking parallelizeMcForLoop intends,
#              parallel_utilities.py:46-60), each rank solves its slice against its replica, and the slices are
#              all-gathered (posterior MEAN and the per-point variance VECTOR on every rank, not only the IVAR sum).
#   gradient   loglikeParams(returnDeriv=1): dist_lml_grad (row slabs of K^-1);  MI design: dist_mi_greedy.
# Small problems stay replicated (every rank computes everything, no exchange): below GPX_DIST_MIN_N training points
# (default 2048) a distributed factorisation is latency, not throughput, and below GPX_DIST_MIN_M evaluation points
# (default 4096) so is the gather.  All ranks take these decisions from the same numbers.
_session = None
_auto_tried = False


class Session:
    """Link between the GPEXP class API and the distributed runners of this module.  `be` = the device backend module
    (gpexp_amd.device; the CPU tests pass a NumPy double), `ops_factory()` -> DeviceOps2D (or its double)."""

    def __init__(self, ctx, comm, be=None, ops_factory=None, min_n=None, min_m=None, check=None):
        self.ctx, self.comm = ctx, comm
        self.be = be or _dev
        self.ops_factory = ops_factory or (lambda: DeviceOps2D(ctx))
        self.min_n = 2048 if min_n is None else int(min_n)
        self.min_m = 4096 if min_m is None else int(min_m)
        self.check = True if check is None else bool(check)
        self._runner = None          # (key, DistFitIvar2D): buffers + recorded programs of the last problem size
        self._scratch_runner = None  # the same for fits nobody keeps (likelihood evaluations) in the distributed-factor mode
        self.stats = dict(fits=0, evals=0, grads=0)
        # Round 5, distributed-factor mode (GPX_DIST_FACTOR=cyclic | replica; default: cyclic from 65536
        # training points, where two N x N replicas per rank stop being cheap): the fit leaves the factor block-cyclic on the
        # ranks -- N^2 / W per rank + the ring of packed panel buffers -- and the class API works on THAT: coeff / log-marginal by
        # distributed substitution, evaluate / evaluateVariance / the IVAR cost by re-streaming the panels against a window
        # (dist2_restream_enqueue); a dense replica is assembled only when an entry point needs one (CyclicFactor.dense).
        self.factor_mode = os.environ.get("GPX_DIST_FACTOR", "auto")
        self.cyclic_min_n = 65536

    @property
    def rank(self):
        return self.comm.rank

    @property
    def world(self):
        return self.comm.world

    # ---- decisions (functions of sizes only: identical on every rank) ----
    def use_fit(self, n):
        return self.world > 1 and n >= max(self.min_n, 1)

    def use_eval(self, m):
        return self.world > 1 and m >= max(self.min_m, self.world)

    def nb_for(self, n):
        return default_nb(n, self.world, False)

    def use_cyclic(self, n):
        return self.factor_mode == "cyclic" or (self.factor_mode == "auto" and n >= self.cyclic_min_n)

    # ---- the SPMD contract, checked ----
    def agree(self, what, *arrays):
        """Every rank must have passed the same data: one all-gather of a 96-bit digest; raises on EVERY rank otherwise
        (a rank that went on alone would hang the others in the next collective).  GPX_DIST_CHECK=0 switches it off."""
        if not self.check:
            return
        import hashlib
        h = hashlib.blake2b(digest_size=12)
        for a in arrays:
            a = np.ascontiguousarray(a)
            h.update(str((a.dtype.str, a.shape)).encode())
            h.update(a.tobytes())
        raw = h.digest()
        mine = np.array([float(int.from_bytes(raw[0:6], "little")), float(int.from_bytes(raw[6:12], "little"))])
        allv = self.comm.allgather(mine)
        if not (np.all(allv[:, 0] == allv[0, 0]) and np.all(allv[:, 1] == allv[0, 1])):
            raise RuntimeError("gpexp_amd.dist: the ranks disagree on %s -- under a multi-process launch every rank must run "
                               "the same calls on the same data (seed the generators alike); rank %d of %d"
                               % (what, self.rank, self.world))

    # ---- fit ----
    def _get_runner(self, n, d, nb):
        key = (int(n), int(d), int(nb))
        if self._runner is not None and self._runner[0] == key:
            return self._runner[1]
        self._runner = None           # release the previous size's buffers before allocating the next
        run = DistFitIvar2D(self.ctx, self.comm, None, np.zeros((n, d)), np.zeros(n), np.zeros((0, d)), 0.0, nb=nb,
                            ops=self.ops_factory(), streamed=False, replicate=True)
        self._runner = (key, run)
        return run

    def _get_cyclic_runner(self, n, d, nb, keep):
        slot = "_runner_c" if keep else "_scratch_runner"
        key = (int(n), int(d), int(nb))
        cur = getattr(self, slot, None)
        if cur is not None and cur[0] == key:
            return cur[1]
        setattr(self, slot, None)
        run = DistFitIvar2D(self.ctx, self.comm, None, np.zeros((n, d)), np.zeros(n), np.zeros((0, d)), 0.0, nb=nb,
                            ops=self.ops_factory(), streamed=False, cyclic=True)
        setattr(self, slot, (key, run))
        return run

    def _fit_cyclic(self, run, spec, nodes, nugget):
        run.spec = spec
        run.X = run.ops.points(nodes)
        run.noise = nugget if isinstance(nugget, np.ndarray) else float(nugget)
        run.fit()
        self.stats["fits"] += 1

    def _refit(self, fac):
        """the runner of a kept CyclicFactor has been used by a later fit: the same fit once more (deterministic)"""
        self._fit_cyclic(fac.run, fac.spec, fac.nodes, fac.nugget)
        fac.generation = fac.run.generation
        fac._dense = None
        self.stats["cyclic_refits"] = self.stats.get("cyclic_refits", 0) + 1

    def factor(self, spec, nodes, nugget, keep=True):
        """Distributed assembly + factorisation of K(nodes) + diag(nugget) -> (X, L): the device point set and the factor --
        this rank's replica (an ordinary factored matrix: every single-GPU entry point works on it) or, in the distributed-factor
        mode, a CyclicFactor.  keep = False: the caller drops the factor at once (a likelihood evaluation).  Raises
        NotPositiveDefinite on every rank alike (the pivot index is agreed by an all-gather)."""
        nodes = as_f64(nodes)
        n, d = nodes.shape
        self.agree("the training set / hyper-parameters", nodes, np.asarray(nugget, dtype=float), spec.hyp,
                   np.array([spec.kind, spec.d]))
        if self.use_cyclic(n):
            run = self._get_cyclic_runner(n, d, self.nb_for(n), keep)
            self._fit_cyclic(run, spec, nodes, nugget)
            nug = nugget.copy() if isinstance(nugget, np.ndarray) else nugget
            return run.X, CyclicFactor(self, run, spec, nodes.copy(), nug)
        run = self._get_runner(n, d, self.nb_for(n))
        run.spec = spec
        run.X = run.ops.points(nodes)
        run.noise = nugget if isinstance(nugget, np.ndarray) else float(nugget)
        # Round 5: the caller gets the runner's replica ITSELF, not a gpx_mat_clone of it (8.6 GB copied at C4 on every call,
        # every likelihood evaluation of an optimiser loop included, and two N x N matrices resident per rank).  Round 6 (ADVICE
        # r5): OWNERSHIP IS EXPLICIT -- keep = True hands the matrix over for good (GP.train keeps it; a state object may hold
        # only its C handle, which no reference count sees), and the runner takes a fresh one for its next fit, re-recording its
        # program (the recorded rows carry the matrix handle); keep = False (a likelihood evaluation: the caller is done with
        # the factor before it asks for the next) leaves it with the runner, which refits in place.
        if getattr(run, "_handed_over", False):
            run.L = run.ops.alloc_matrix(run.n)
            run.programs, run._runs = None, {}
            run._handed_over = False
            self.stats["replicas_handed_over"] = self.stats.get("replicas_handed_over", 0) + 1
        run.fit()
        run._handed_over = bool(keep)
        self.stats["fits"] += 1
        return run.X, run.L

    # ---- evaluation ----
    def gather(self, local, m, width=1):
        """Concatenate the ranks' slices (eval_slice order) of an (m, width) quantity; `local` = this rank's rows."""
        world = self.world
        per = (m + world - 1) // world
        buf = np.zeros(per * width)
        loc = as_f64(local).reshape(-1)
        buf[:loc.size] = loc
        allv = self.comm.allgather(buf)
        out = np.empty((m, width))
        for r in range(world):
            lo, hi = eval_slice(m, r, world)
            out[lo:hi] = allv[r, :(hi - lo) * width].reshape(hi - lo, width)
        return out

    def posterior(self, spec, L, X, alpha, newpt, want_mean=True, want_var=True):
        """(mean, signed variance) at `newpt` (host array) on every rank; evaluation points sharded over the ranks."""
        be, ctx = self.be, self.ctx
        newpt = as_f64(newpt)
        m = newpt.shape[0]
        cyc = getattr(L, "is_cyclic", False)
        if not self.use_eval(m) and not cyc:
            return be.posterior(ctx, spec, L, X, alpha, be.points(ctx, newpt), want_mean=want_mean, want_var=want_var)
        self.agree("the evaluation points", newpt)
        lo, hi = eval_slice(m, self.rank, self.world)
        cols = []
        if cyc:      # the factor is block-cyclic on the ranks: every rank takes part in the re-stream, also with an empty slice
            mean, var = L.posterior(newpt[lo:hi], alpha, want_mean, want_var)
            if hi > lo:
                cols = [c for c in (mean if want_mean else None, var if want_var else None) if c is not None]
        elif hi > lo:
            mean, var = be.posterior(ctx, spec, L, X, alpha, be.points(ctx, newpt[lo:hi]), want_mean=want_mean,
                                     want_var=want_var)
            cols = [c for c in (mean if want_mean else None, var if want_var else None) if c is not None]
        width = int(want_mean) + int(want_var)
        local = np.stack(cols, axis=1) if cols else np.zeros((0, width))
        full = self.gather(local, m, width)
        self.stats["evals"] += 1
        return (full[:, 0].copy() if want_mean else None), (full[:, width - 1].copy() if want_var else None)

    def ivar(self, spec, L, X, mc, cache=None):
        """Signed mean posterior variance over the MC points (costFunctionGP_IVAR.evaluate, experimentalDesign.py:100-117):
        each rank sums the variances of its slice, partial sums are added in rank order."""
        be, ctx = self.be, self.ctx
        m = mc.shape[0]
        lo, hi = eval_slice(m, self.rank, self.world)
        part = 0.0
        if getattr(L, "is_cyclic", False):     # block-cyclic factor: the slice's variances by the re-streamed solve (collective)
            self.agree("the Monte-Carlo points", mc)
            _, var = L.posterior(as_f64(mc[lo:hi]), None, False, True)
            part = float(np.sum(var))
            self.stats["evals"] += 1
            return ordered_sum(self.comm.allgather(np.array([part]))[:, 0]) / max(m, 1)
        if hi > lo:
            Z = getattr(cache, "_mc_slice_dev", None) if cache is not None else None
            if Z is None:
                self.agree("the Monte-Carlo points", mc)
                Z = be.points(ctx, as_f64(mc[lo:hi]))
                if cache is not None:
                    cache._mc_slice_dev = Z
            part = be.ivar(ctx, spec, L, X, Z) * (hi - lo)
        elif cache is None or getattr(cache, "_mc_slice_dev", None) is None:
            self.agree("the Monte-Carlo points", mc)
            if cache is not None:
                cache._mc_slice_dev = False
        self.stats["evals"] += 1
        return ordered_sum(self.comm.allgather(np.array([part]))[:, 0]) / max(m, 1)

    def lml_grad(self, spec, L, X, alpha):
        self.stats["grads"] += 1
        return dist_lml_grad(self.ctx, self.comm, spec, L, X, alpha, be=self.be)

    def greedy_ivar_step(self, spec, L, X, cand, mc, noise):
        """(best index, costs[M]) of one discrete greedy-IVAR step with the CANDIDATES sharded (dist_greedy_ivar_step) and the
        cost vector gathered; the winner follows np.argmin (lowest cost, ties -> lowest global index)."""
        be, ctx = self.be, self.ctx
        cand = as_f64(cand)
        m = cand.shape[0]
        self.agree("the candidate / Monte-Carlo points", cand, as_f64(mc))
        lo, hi = eval_slice(m, self.rank, self.world)
        Z = be.points(ctx, as_f64(mc))
        if hi > lo:
            cpts = be.points_slice(ctx, cand, lo, hi) if hasattr(be, "points_slice") else be.points(ctx, cand[lo:hi])
            best, costs = be.greedy_ivar_step(ctx, spec, L, X, cpts, Z, noise)
            mine = np.array([costs[best], float(lo + best)])
        else:
            costs, mine = np.zeros(0), np.array([np.inf, float(m)])
        pairs = self.comm.allgather(mine)
        _, idx = merge_argmin(pairs[:, 0], pairs[:, 1].astype(np.int64))
        return idx, self.gather(costs, m)[:, 0].copy()

    def greedy_ivar(self, spec, L, X, cand, mc, noise, nsel, want_all=False):
        """nsel picks of discrete greedy IVAR with the CANDIDATES sharded and the state resident (dist_greedy_ivar)."""
        self.agree("the candidate / Monte-Carlo points", as_f64(cand), as_f64(mc))
        return dist_greedy_ivar(self.ctx, self.comm, spec, L, X, as_f64(cand), as_f64(mc), noise, nsel, want_all=want_all,
                                be=self.be)

    def mi_greedy(self, spec, cand, noise, nsel, start=0):
        self.agree("the candidate points", as_f64(cand))
        return dist_mi_greedy(self.ctx, self.comm, spec, as_f64(cand), noise, nsel, start=start, be=self.be)

    def close(self):
        self._runner = None
        try:
            self.comm.close()
        except Exception:
            pass


def attach(comm=None, ctx=None, **opts):
    """Route the class API of this process through the distributed runners.  Without arguments: the communicator of the
    launcher's process group (RANK / WORLD_SIZE / MASTER_* -> RCCL).  Called implicitly on the first use of a GP under
    WORLD_SIZE > 1 (GPX_DIST_ATTACH=0 keeps the process single-GPU); explicit calls are for tests and embedders."""
    global _session, _auto_tried
    _auto_tried = True
    if _session is not None:
        return _session
    if ctx is None and comm is None:
        ctx = _dev.context()
    if comm is None:
        comm = init_from_env(ctx)
    _session = Session(ctx, comm, **opts)
    import atexit
    atexit.register(detach)
    return _session


def detach():
    global _session
    if _session is not None:
        s, _session = _session, None
        s.close()


def session():
    """The active Session or None (single-process use: nothing changes)."""
    global _auto_tried
    if _session is None and not _auto_tried:
        _auto_tried = True
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("GPX_DIST_ATTACH", "1") != "0":
            attach()
    return _session
