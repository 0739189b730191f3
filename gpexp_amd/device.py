"""Thin Python objects over the C ABI: Context (one per process/GPU), DeviceMatrix, and the
function wrappers the GPEXP-API classes call.  All arithmetic happens in libgpx_hip.so."""
import ctypes as C
import atexit
import os

import numpy as np

from . import _lib
from ._lib import check, as_f64, dptr, c_vp, c_i64, GpxError

K_SE, K_MATERN32, K_MATERN52, K_MEHLER = 0, 1, 2, 3
PROF_CLASSES = ["kfill", "gemm", "leaf", "trsv", "reduce", "greedy", "comm", "kcross"]

_ctx = None


class Context:
    """Owns the gpx_ctx for this process.  device defaults to LOCAL_RANK (one process per GPU)."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        h = c_vp()
        rc = self.lib.gpx_create(int(device), C.byref(h))
        if rc != 0:
            raise RuntimeError("gpexp_amd needs an MI355X: " + self.lib.gpx_last_error().decode())
        self.h = h
        self.device = int(device)

    def close(self):
        if self.h:
            self.lib.gpx_destroy(self.h)
            self.h = None

    def sync(self):
        check(self.lib.gpx_sync(self.h))

    def guard_violations(self):
        """Pooled blocks found overwritten outside their bounds (GPX_ALLOC_GUARD=1), -1 when the mode is off."""
        return int(self.lib.gpx_dbg_guard_violations(self.h))

    def trim(self):
        check(self.lib.gpx_trim(self.h))

    def info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        mem = c_i64()
        clk = C.c_int()
        check(self.lib.gpx_device_info(self.h, name, 256, C.byref(cus), C.byref(mem), C.byref(clk)))
        return dict(name=name.value.decode(), cus=cus.value, hbm_bytes=mem.value, clock_mhz=clk.value)

    # ---- streams / events (0 main, 1 side high-priority, 2 communication) ----
    def stream(self, which):
        check(self.lib.gpx_stream_select(self.h, int(which)))

    def record(self, ev):
        check(self.lib.gpx_event_record(self.h, int(ev)))

    def wait(self, ev):
        check(self.lib.gpx_event_wait(self.h, int(ev)))

    # ---- profiling ----
    def profile(self, on):
        check(self.lib.gpx_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.gpx_profile_reset(self.h))

    def profile_get(self):
        out = {}
        for i, nm in enumerate(PROF_CLASSES):
            n = c_i64()
            ms = C.c_double()
            fl = C.c_double()
            by = C.c_double()
            check(self.lib.gpx_profile_get(self.h, i, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
            out[nm] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return out


def _shutdown():
    # destroy the context (streams, pool) while the HIP runtime and any profiler tool library are still loaded: leaving it
    # to interpreter teardown crashes at exit under rocprofv3; matrices that outlive it see ctx.h == None and do nothing
    global _ctx
    if _ctx is not None and _ctx.h:
        try:
            _ctx.close()
        except Exception:
            pass


def context():
    """Process-wide context, created on first use (fails loudly without a GPU)."""
    global _ctx
    if _ctx is None:
        _ctx = Context()
        atexit.register(_shutdown)
    return _ctx


class DeviceMatrix:
    """Library-owned fp64 device matrix (row-major, padded to multiples of 128 when pad=True)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    @classmethod
    def from_host(cls, ctx, a, pad=False):
        a = as_f64(a)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        h = c_vp()
        check(ctx.lib.gpx_mat_from_host(ctx.h, dptr(a), a.shape[0], a.shape[1], 1 if pad else 0, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def zeros(cls, ctx, rows, cols, pad=True):
        h = c_vp()
        check(ctx.lib.gpx_mat_alloc(ctx.h, rows, cols, 1 if pad else 0, C.byref(h)))
        return cls(ctx, h)

    @property
    def shape(self):
        r, c, ld = c_i64(), c_i64(), c_i64()
        check(self.ctx.lib.gpx_mat_shape(self.h, C.byref(r), C.byref(c), C.byref(ld)))
        return (r.value, c.value)

    def to_host(self, tri=0):
        r, c = self.shape
        out = np.empty((r, c), dtype=np.float64)
        check(self.ctx.lib.gpx_mat_to_host(self.ctx.h, self.h, dptr(out), int(tri)))
        return out

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.gpx_mat_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def clone(ctx, M):
    """Device-to-device duplicate of a matrix including its factor state (gpx_mat_clone)."""
    h = c_vp()
    check(ctx.lib.gpx_mat_clone(ctx.h, M.h, C.byref(h)))
    return DeviceMatrix(ctx, h)


class KernelSpec:
    """Flat (kind, d, hyp[]) form of a covariance kernel, as the C ABI takes it."""

    def __init__(self, kind, d, hyp):
        self.kind = int(kind)
        self.d = int(d)
        self.hyp = as_f64(np.asarray(hyp, dtype=float).ravel())

    def args(self):
        return (self.kind, self.d, dptr(self.hyp), int(self.hyp.size))

    @property
    def nlen(self):
        """Length-type hyper-parameters in the log-marginal gradient: d correlation lengths (squared exponential) or the one rho
        (isotropic Matern)."""
        return self.d if self.kind == K_SE else 1

    @property
    def nsums(self):
        """Trace sums the gradient entry points return: nlen + signalSize + noise."""
        return self.nlen + 2


def points(ctx, x):
    """Upload an (n, d) point set unpadded."""
    x = as_f64(x)
    assert x.ndim == 2
    return DeviceMatrix.from_host(ctx, x, pad=False)


def points_slice(ctx, x, lo, hi):
    """Rows [lo, hi) of the host point set `x` on the device, carrying the bounding box of ALL of x (gpx_points_set_box): what a
    rank uploads of a sharded candidate / evaluation set, so that its fills use the same centring as the unsharded call."""
    x = as_f64(x)
    P = points(ctx, x[lo:hi])
    if hi > lo and x.shape[0] > 0:
        blo, bhi = np.ascontiguousarray(x.min(axis=0)), np.ascontiguousarray(x.max(axis=0))
        check(ctx.lib.gpx_points_set_box(ctx.h, P.h, dptr(blo), dptr(bhi), int(x.shape[1])))
    return P


def _nugget_args(nugget, n):
    """float -> scalar on the diagonal; ndarray (n,) -> per point (gp_kernel_utilities.py:62-65)."""
    if isinstance(nugget, float):
        if nugget == 0.0:
            return None, 0
        a = as_f64([nugget])
        return a, 1
    if isinstance(nugget, np.ndarray):
        a = as_f64(nugget.ravel())
        assert a.size == n, "per-point nugget must have one entry per point"
        return a, n
    raise TypeError("nugget must be a float or an ndarray (an int raises in the reference too, "
                    "gp_kernel_utilities.py:62-67)")


def kfill(ctx, spec, X, Z=None, nugget=0.0):
    """K(X,X)+diag(nugget) (Z None) or K(X,Z); X, Z are DeviceMatrix point sets."""
    n = X.shape[0]
    nug, nlen = _nugget_args(nugget, n) if Z is None else (None, 0)
    h = c_vp()
    check(ctx.lib.gpx_kfill(ctx.h, *spec.args(), X.h, Z.h if Z is not None else None, dptr(nug), nlen, C.byref(h)))
    return DeviceMatrix(ctx, h)


def kfill_into(ctx, spec, X, K, Z=None, nugget=0.0):
    n = X.shape[0]
    nug, nlen = _nugget_args(nugget, n) if Z is None else (None, 0)
    check(ctx.lib.gpx_kfill_into(ctx.h, *spec.args(), X.h, Z.h if Z is not None else None, dptr(nug), nlen, K.h))
    return K


def kdiag(ctx, spec, Z):
    out = np.empty(Z.shape[0])
    check(ctx.lib.gpx_kdiag(ctx.h, *spec.args(), Z.h, dptr(out)))
    return out


def kernel_eval(ctx, spec, x1, x2):
    """Paired / one-vs-n evaluation on host arrays (Kernel.evaluate, kernels.py:49-65)."""
    x1, x2 = as_f64(x1), as_f64(x2)
    out = np.empty(max(x1.shape[0], x2.shape[0]))
    check(ctx.lib.gpx_kernel_eval(ctx.h, *spec.args(), dptr(x1), x1.shape[0], dptr(x2), x2.shape[0], dptr(out)))
    return out


def potrf(ctx, K):
    check(ctx.lib.gpx_potrf(ctx.h, K.h))
    return K


def potrf_policy(ctx, piv_min=0.0, skip=False):
    """Pivot policy of the factorisations that follow (gpx_potrf_policy); (0, False) is the library default.  Returns the
    policy that was in force before, so that a caller can put it back (the policy is state of the shared context)."""
    prev = getattr(ctx, "_potrf_policy", (0.0, False))
    check(ctx.lib.gpx_potrf_policy(ctx.h, float(piv_min), 1 if skip else 0))
    ctx._potrf_policy = (float(piv_min), bool(skip))
    return prev


def potrf_dropped(ctx):
    n = C.c_int()
    check(ctx.lib.gpx_potrf_dropped(ctx.h, C.byref(n)))
    return n.value


def refit_rows(ctx, spec, X, nugget, L_old, keep):
    """Cholesky factor of K(X,X)+diag(nugget) re-using the leading `keep` (multiple of 128) rows of the factor `L_old`."""
    nug, nlen = _nugget_args(nugget, X.shape[0])
    h = c_vp()
    check(ctx.lib.gpx_refit_rows(ctx.h, *spec.args(), X.h, dptr(nug), nlen, L_old.h if keep > 0 else None, int(keep),
                                 C.byref(h)))
    return DeviceMatrix(ctx, h)


def matvec(ctx, A, v):
    """A @ v for a resident DeviceMatrix (deterministic row reduction)."""
    v = as_f64(v)
    assert v.shape == (A.shape[1],)
    out = np.empty(A.shape[0])
    check(ctx.lib.gpx_matvec(ctx.h, A.h, dptr(v), dptr(out)))
    return out


class FitcModel:
    """Device-resident FITC model (gpx_fitc_*): chol(Quu), Kuf, G and chol(Quu + Kuf G^-1 Kfu)."""

    def __init__(self, ctx, spec, X, S, noise):
        self.ctx, self.X = ctx, X
        h = c_vp()
        check(ctx.lib.gpx_fitc_fit(ctx.h, *spec.args(), X.h, S.h, float(noise), C.byref(h)))
        self.h = h
        self.n = X.shape[0]

    def solve(self, y):
        y = as_f64(y)
        coeff = np.empty(self.n)
        quad = C.c_double()
        check(self.ctx.lib.gpx_fitc_solve(self.ctx.h, self.h, dptr(y), dptr(coeff), C.byref(quad)))
        return coeff, quad.value

    def logdet(self):
        out = C.c_double()
        check(self.ctx.lib.gpx_fitc_logdet(self.ctx.h, self.h, C.byref(out)))
        return out.value

    def posterior(self, coeff, Z, want_mean=True, want_var=True):
        m = Z.shape[0]
        mean = np.empty(m) if want_mean else None
        var = np.empty(m) if want_var else None
        co = as_f64(coeff) if want_mean else None
        check(self.ctx.lib.gpx_fitc_posterior(self.ctx.h, self.h, self.X.h, dptr(co), Z.h, dptr(mean), dptr(var)))
        return mean, var

    def var_grad(self, spec, Z, noise_deriv=None, eval_bias=None, dk_bias=None):
        """(N*d, M) matrix d var(z_m) / d X[j][l] with the FITC precision (gp.py:282-341 reads `precisionMatrix`, gp.py:194-206)."""
        out = np.empty((self.n * spec.d, Z.shape[0]))
        nd = as_f64(noise_deriv) if noise_deriv is not None else None
        eb = as_f64(eval_bias) if eval_bias is not None else None
        db = as_f64(dk_bias) if dk_bias is not None else None
        check(self.ctx.lib.gpx_fitc_var_grad(self.ctx.h, self.h, *spec.args(), self.X.h, Z.h, dptr(nd), dptr(eb), dptr(db), dptr(out)))
        return out

    def var_grad_newpt(self, spec, Z):
        """(M*d,) vector d var(z_m) / d z_m with the FITC precision (gp.py:261-280)."""
        out = np.empty(Z.shape[0] * spec.d)
        check(self.ctx.lib.gpx_fitc_var_grad_newpt(self.ctx.h, self.h, *spec.args(), self.X.h, Z.h, dptr(out)))
        return out

    def dense(self, cov=True, prec=True):
        c = np.empty((self.n, self.n)) if cov else None
        p = np.empty((self.n, self.n)) if prec else None
        check(self.ctx.lib.gpx_fitc_dense(self.ctx.h, self.h, dptr(c), dptr(p)))
        return c, p

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.gpx_fitc_free(self.ctx.h, self.h)
            self.h = None
        except Exception:
            pass


def potrs(ctx, L, y):
    y = as_f64(y)
    out = np.empty_like(y)
    check(ctx.lib.gpx_potrs(ctx.h, L.h, dptr(y), dptr(out)))
    return out


def padded_vector(ctx, v, n_padded=None):
    """Upload a vector as an (n_padded x 1) device matrix, zero padded to a multiple of 128."""
    v = as_f64(np.ravel(v))
    npad = n_padded or (max(v.size, 1) + 127) // 128 * 128
    buf = np.zeros(npad)
    buf[:v.size] = v
    return DeviceMatrix.from_host(ctx, buf.reshape(-1, 1), pad=False)


def potrs_dev(ctx, L, y_dev, alpha_dev):
    """alpha_dev <- K^-1 y_dev, asynchronous on the currently selected stream (see Context.stream)."""
    check(ctx.lib.gpx_potrs_dev(ctx.h, L.h, y_dev.h, alpha_dev.h))


def logdet(ctx, L):
    v = C.c_double()
    check(ctx.lib.gpx_logdet(ctx.h, L.h, C.byref(v)))
    return v.value


def potri(ctx, L):
    h = c_vp()
    check(ctx.lib.gpx_potri(ctx.h, L.h, C.byref(h)))
    return DeviceMatrix(ctx, h)


def posterior(ctx, spec, L, X, alpha, Z, want_mean=True, want_var=True):
    m = Z.shape[0]
    mean = np.empty(m) if want_mean else None
    var = np.empty(m) if want_var else None
    al = as_f64(alpha) if (alpha is not None and want_mean) else None
    check(ctx.lib.gpx_posterior(ctx.h, *spec.args(), L.h, X.h, dptr(al), Z.h, dptr(mean), dptr(var)))
    return mean, var


def posterior_cov(ctx, spec, L, X, Z):
    m = Z.shape[0]
    cov = np.empty((m, m))
    check(ctx.lib.gpx_posterior_cov(ctx.h, *spec.args(), L.h, X.h, Z.h, dptr(cov)))
    return cov


def ivar(ctx, spec, L, X, Z, keep=False):
    """Mean posterior variance over Z.  keep=True -> (cost, W): W = L^-1 K(X, Z) stays on the device for ivar_grad at the same
    design (None when Z does not fit one evaluation chunk)."""
    v = C.c_double()
    if not keep:
        check(ctx.lib.gpx_ivar(ctx.h, *spec.args(), L.h, X.h, Z.h, C.byref(v)))
        return v.value
    h = c_vp()
    check(ctx.lib.gpx_ivar_keep(ctx.h, *spec.args(), L.h, X.h, Z.h, C.byref(v), C.byref(h)))
    return v.value, (DeviceMatrix(ctx, h) if h.value else None)


def ivar_update(ctx, spec, L, X, Z, W, keep):
    """The cost after a refit that kept the leading `keep` rows of the factor: W (kept for the previous design) is updated in
    place from row `keep` on (gpx_ivar_update)."""
    v = C.c_double()
    check(ctx.lib.gpx_ivar_update(ctx.h, *spec.args(), L.h, X.h, Z.h, W.h, int(keep), C.byref(v)))
    return v.value


def fit_ivar(ctx, spec, K, X, Z):
    """Factor the assembled covariance K in place and return the (signed) IVAR over Z, the evaluation solve streamed
    underneath the factorisation (gpx_fit_ivar).  Raises NotPositiveDefinite like potrf."""
    v = C.c_double()
    check(ctx.lib.gpx_fit_ivar(ctx.h, *spec.args(), K.h, X.h, Z.h, C.byref(v)))
    return v.value


def greedy_var(ctx, spec, Cpts, nsel, keep=(), weights=None):
    keep = np.ascontiguousarray(np.asarray(list(keep), dtype=np.int64))
    out = np.empty(int(nsel), dtype=np.int64)
    w = as_f64(weights) if weights is not None else None
    check(ctx.lib.gpx_greedy_var(ctx.h, *spec.args(), Cpts.h, dptr(w),
                                 keep.ctypes.data_as(_lib.c_ip), int(keep.size), int(nsel),
                                 out.ctypes.data_as(_lib.c_ip)))
    return out


def greedy_ivar_step(ctx, spec, L, X, Cpts, Z, noise, want_costs=True):
    m = Cpts.shape[0]
    costs = np.empty(m) if want_costs else None
    best = c_i64()
    check(ctx.lib.gpx_greedy_ivar_step(ctx.h, *spec.args(), L.h, X.h, Cpts.h, Z.h, float(noise), dptr(costs),
                                       C.byref(best)))
    return best.value, costs


def greedy_ivar(ctx, spec, L, X, Cpts, Z, noise, nsel, want_all=False):
    """nsel picks of discrete greedy IVAR with resident state (gpx_greedy_ivar): (indices, winner costs[, all costs nsel x M])."""
    nsel = int(nsel)
    idx = np.empty(nsel, dtype=np.int64)
    cost = np.empty(nsel)
    allc = np.empty((nsel, Cpts.shape[0])) if want_all else None
    check(ctx.lib.gpx_greedy_ivar(ctx.h, *spec.args(), L.h, X.h, Cpts.h, Z.h, float(noise), nsel,
                                  idx.ctypes.data_as(_lib.c_ip), dptr(cost), dptr(allc)))
    return (idx, cost, allc) if want_all else (idx, cost)


class GivarState:
    """One rank's state of a candidate-sharded greedy-IVAR run (gpx_givar_*): W_C = L^-1 K(X, C_local), cov(Z, C_local | design)
    and the score vectors stay on the device; a pick is one pivot pack (from the winner's owner) applied by every rank."""

    def __init__(self, ctx, spec, L, X, Cpts, Z, noise, nsel):
        self.ctx, self.Cpts, self.Z, self.L, self.X = ctx, Cpts, Z, L, X      # (kept alive: the library reads C's points per pick)
        self.m = Cpts.shape[0]
        h = c_vp()
        check(ctx.lib.gpx_givar_begin(ctx.h, *spec.args(), L.h, X.h, Cpts.h, Z.h, float(noise), int(nsel), C.byref(h)))
        self.h = h
        self.pivot_elems = int(ctx.lib.gpx_givar_pivot_elems(h))

    def score(self, want_all=False):
        v, i = C.c_double(), c_i64()
        allc = np.empty(self.m) if want_all else None
        check(self.ctx.lib.gpx_givar_score(self.ctx.h, self.h, C.byref(v), C.byref(i), dptr(allc)))
        return v.value, i.value, allc

    def pack(self, s, buf):
        check(self.ctx.lib.gpx_givar_pack(self.ctx.h, self.h, int(s), buf.h))

    def apply(self, buf):
        check(self.ctx.lib.gpx_givar_apply(self.ctx.h, self.h, buf.h))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.gpx_givar_end(self.ctx.h, self.h)
            self.h = None
        except Exception:
            pass


def mi_greedy(ctx, spec, Cpts, noise, nsel, start=0):
    out = np.empty(int(nsel), dtype=np.int64)
    ratios = np.empty(max(int(nsel) - 1, 0))
    check(ctx.lib.gpx_mi_greedy(ctx.h, *spec.args(), Cpts.h, float(noise), int(nsel), int(start),
                                out.ctypes.data_as(_lib.c_ip), dptr(ratios) if ratios.size else None))
    return out, ratios


# From this order on the gradient is formed slab by slab (no N x N inverse is ever allocated).  Measured warm, MI355X (scripts/
# probe_lml_grad.py): N = 32768: potri form 0.365 s, 32 slabs 0.485 s; N = 65536: 2.82 s against 3.48 s -- but the potri form
# then holds 210 GB of the 288 GB (inverse + the recursion's scratch), so the default stays on the side of memory.
LML_GRAD_SLAB_MIN = 49152


def lml_grad_full(ctx, spec, L, X, alpha):
    """gpx_lml_grad: explicit inverse (potri) + one fused trace pass over it -- allocates an N x N matrix."""
    alpha = as_f64(alpha)
    out = np.empty(spec.hyp.size + 1)
    check(ctx.lib.gpx_lml_grad(ctx.h, *spec.args(), L.h, X.h, dptr(alpha), dptr(out)))
    return out


def lml_grad_linv(ctx, spec, L, X, alpha):
    """Raw trace sums (d+2) over all rows through ONE explicit L^-1 and the lower triangle of K^-1 written over it
    (gpx_lml_grad_linv): the single-GPU form for large N when two more N x N buffers fit."""
    alpha = as_f64(alpha)
    out = np.empty(spec.nsums)
    check(ctx.lib.gpx_lml_grad_linv(ctx.h, *spec.args(), L.h, X.h, dptr(alpha), dptr(out)))
    return out


def lml_grad_linv_fits(ctx, n):
    """Room for gpx_lml_grad_linv's buffers (L^-1 / K^-1: N^2, U: N^2, recursion scratch N^2/4) beside the factor itself and
    whatever else is resident: at most 60 % of the HBM for 3.4 N^2 doubles (N = 65536: 117 of 288 GB)."""
    if not hasattr(ctx, "_hbm_bytes"):
        ctx._hbm_bytes = ctx.info()["hbm_bytes"]
    return 3.4 * 8.0 * float(n) ** 2 <= 0.6 * ctx._hbm_bytes


def lml_grad(ctx, spec, L, X, alpha, slabs=None):
    """[d/d hyp_0 .. d/d hyp_{n-1}, raw d/d noise] of the log marginal likelihood (squared exponential: gp.py:444-466; the
    isotropic Materns: round 6, absent in the reference).
    Large factors: the traces are summed over `slabs` row slabs of K^-1 of equal work (gpx_lml_grad_slab: two triangular solves
    against the trailing factor per slab) -- the same flops as potri in the limit, but no N x N inverse in memory (34 GB at
    N = 65536, whose first allocation alone costs seconds) and the form the multi-GPU path shards; small ones: gpx_lml_grad."""
    n = X.shape[0]
    if slabs is None:
        slabs = 32 if n >= LML_GRAD_SLAB_MIN else 0
    if not slabs:
        return lml_grad_full(ctx, spec, L, X, alpha)
    if os.environ.get("GPX_LML_GRAD_FORM", "linv") == "linv" and lml_grad_linv_fits(ctx, n):
        #  the static size test knows nothing of what else the process holds -- the factor, a cached previous factor,
        # pooled blocks, a session's replica --, so the 2 N^2 + N^2/4 doubles of scratch can fail to allocate where the slab form
        # below (no N x N buffer) still runs: an out-of-memory error of the linv form falls through to it
        try:
            return lml_grad_from_sums(spec, lml_grad_linv(ctx, spec, L, X, alpha))
        except GpxError as e:
            if "hipMalloc" not in str(e) and "memory" not in str(e).lower():
                raise
            ctx.trim()
    # next: the rows form over the whole range (one N x N accumulator instead of the linv form's 2 N^2 + N^2 / 4; 2.97 s against the
    # slab loop's 3.5 s at N = 65536); the slab loop needs no N x N buffer at all and is what remains when that does not fit either
    if os.environ.get("GPX_LML_GRAD_FORM", "linv") in ("linv", "rows"):
        try:
            npad = (n + 127) // 128 * 128
            return lml_grad_from_sums(spec, lml_grad_rows(ctx, spec, L, X, alpha, 0, npad, 16))
        except GpxError as e:
            if "hipMalloc" not in str(e) and "memory" not in str(e).lower():
                raise
            ctx.trim()
    b = lml_grad_slab_bounds(n, int(slabs))
    sums = np.zeros(spec.nsums)
    for r0, r1 in zip(b[:-1], b[1:]):
        if r1 > r0:
            sums += lml_grad_slab(ctx, spec, L, X, alpha, r0, r1)
    return lml_grad_from_sums(spec, sums)


class MiState:
    """One rank's state of a row-sharded greedy MI run (gpx_mi_*): rows [lo, hi) of the inverse are kept current."""

    def __init__(self, ctx, spec, Cpts, noise, nsel, start, lo, hi):
        self.ctx, self.Cpts = ctx, Cpts
        self.picks = [int(start)]
        h = c_vp()
        check(ctx.lib.gpx_mi_begin(ctx.h, *spec.args(), Cpts.h, float(noise), int(nsel), int(start), int(lo), int(hi), C.byref(h)))
        self.h = h

    def row(self, cur, rowbuf):
        check(self.ctx.lib.gpx_mi_row(self.ctx.h, self.h, int(cur), int(self.picks[cur]), rowbuf.h))

    def score(self, cur, rowbuf):
        v = C.c_double()
        i = c_i64()
        check(self.ctx.lib.gpx_mi_score(self.ctx.h, self.h, int(cur), rowbuf.h, C.byref(v), C.byref(i)))
        return v.value, i.value

    def select(self, slot, idx):
        check(self.ctx.lib.gpx_mi_select(self.ctx.h, self.h, int(slot), int(idx)))
        assert slot == len(self.picks)
        self.picks.append(int(idx))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.gpx_mi_end(self.ctx.h, self.h)
            self.h = None
        except Exception:
            pass


def alloc_vector(ctx, n):
    """Zeroed device vector of n doubles (an unpadded n x 1 matrix)."""
    return DeviceMatrix.zeros(ctx, max(int(n), 1), 1, pad=False)


def lml_grad_slab_bounds(n, parts):
    """Row boundaries (multiples of 128, inside the padded order) that cut the trace sums of the log-marginal gradient into
    `parts` slabs of equal WORK: slab [r0, r1) costs ~ (N - r0)^2 (r1 - r0), so r_i = N (1 - (1 - i/parts)^(1/3))."""
    npad = (max(n, 1) + 127) // 128 * 128
    q = 1024 if npad >= 8192 else 128     # large factors: on the boundaries of the 1024-order block inverses the solves use
    b = [int(round(npad * (1.0 - (1.0 - i / parts) ** (1.0 / 3.0)) / q)) * q for i in range(parts + 1)]
    b[0], b[-1] = 0, npad
    for i in range(1, parts + 1):
        b[i] = max(b[i], b[i - 1])
    return b


def lml_grad_slab(ctx, spec, L, X, alpha, r0, r1):
    """Raw trace sums (d+2) of the log-marginal gradient over the row slab [r0, r1) of K^-1 (gpx_lml_grad_slab)."""
    alpha = as_f64(alpha)
    out = np.empty(spec.nsums)
    check(ctx.lib.gpx_lml_grad_slab(ctx.h, *spec.args(), L.h, X.h, dptr(alpha), int(r0), int(r1), dptr(out)))
    return out


def lml_grad_rows_bounds(n, parts, nsub=1):
    """Row boundaries (multiples of 128, inside the padded order) that cut the ROWS of L^-1 into `parts` ranges of equal COST for
    gpx_lml_grad_rows when every range is worked in `nsub` sub-slabs of equal height: a sub-slab [c0, c1) costs (c1 - c0) c1^2.
    Found by bisection on the cost per range (the ranges are laid out greedily from row 0; the cost of a range grows with its
    end)."""
    npad = (max(n, 1) + 127) // 128 * 128
    nsub = max(1, int(nsub))

    def cost(a, b):
        h = (b - a) / nsub
        return sum(h * (a + (j + 1) * h) ** 2 for j in range(nsub))

    def layout(t):
        b = [0.0]
        for _ in range(parts):
            lo, hi = b[-1], float(npad)
            if cost(b[-1], hi) <= t:
                b.append(hi)
                continue
            for _ in range(60):
                mid = 0.5 * (lo + hi)
                if cost(b[-1], mid) < t:
                    lo = mid
                else:
                    hi = mid
            b.append(hi)
        return b

    lo, hi = 0.0, cost(0.0, float(npad))
    for _ in range(60):
        t = 0.5 * (lo + hi)
        if layout(t)[-1] >= npad - 1e-6:
            hi = t
        else:
            lo = t
    b = [int(round(v / 128)) * 128 for v in layout(hi)]
    b[0], b[-1] = 0, npad
    for i in range(1, parts + 1):
        b[i] = min(max(b[i], b[i - 1]), npad)
    return b


def lml_grad_rows(ctx, spec, L, X, alpha, r0, r1, nsub=1):
    """Raw trace sums (d+2) of the log-marginal gradient contributed by the rows [r0, r1) of L^-1 (gpx_lml_grad_rows; the range
    is worked in `nsub` sub-slabs of equal work whose products accumulate in one matrix, traced once)."""
    alpha = as_f64(alpha)
    out = np.empty(spec.nsums)
    check(ctx.lib.gpx_lml_grad_rows(ctx.h, *spec.args(), L.h, X.h, dptr(alpha), int(r0), int(r1), int(nsub), dptr(out)))
    return out


def lml_grad_from_sums(spec, sums):
    """[d/d cl_0 .. d/d cl_{d-1} (Matern: d/d rho), d/d signalSize, raw d/d noise] from the summed slab traces (as gpx_lml_grad
    returns them)."""
    d = spec.nlen
    g = np.empty(d + 2)
    g[:d] = 0.5 * sums[:d] / spec.hyp[:d]
    g[d] = 0.5 * sums[d] / spec.hyp[d]
    g[d + 1] = 0.5 * sums[d + 1]
    return g


def ivar_grad(ctx, spec, L, X, Z, noise_deriv=None, W=None):
    """d IVAR / d design coordinates, flattened (N*d) in the reference's row order (point-major); noise_deriv (N, d) =
    d noise(x_j)/d x_j of a heteroscedastic noise model; W = the forward solve kept by ivar(..., keep=True) for the same
    L, X, Z."""
    out = np.empty(X.shape[0] * spec.d)
    nd = as_f64(noise_deriv) if noise_deriv is not None else None
    check(ctx.lib.gpx_ivar_grad_w(ctx.h, *spec.args(), L.h, X.h, Z.h, dptr(nd), W.h if W is not None else None, dptr(out)))
    return out


def ivar_grad_rows(ctx, spec, L, X, Z, W, r0):
    """d IVAR / d coordinates of the design points r0.. only (flattened, (N - r0)*d), from the kept forward solve W
    (gpx_ivar_grad_rows: squared exponential, homoscedastic)."""
    out = np.empty((X.shape[0] - int(r0)) * spec.d)
    check(ctx.lib.gpx_ivar_grad_rows(ctx.h, *spec.args(), L.h, X.h, Z.h, W.h, int(r0), dptr(out)))
    return out


def var_grad(ctx, spec, L, X, Z, noise_deriv=None, eval_bias=None, dk_bias=None):
    """(N*d, M) matrix d var(z_m) / d X[j][l] (GP.evaluateVarianceDerivative, gp.py:282-341)."""
    out = np.empty((X.shape[0] * spec.d, Z.shape[0]))
    nd = as_f64(noise_deriv) if noise_deriv is not None else None
    eb = as_f64(eval_bias) if eval_bias is not None else None
    db = as_f64(dk_bias) if dk_bias is not None else None
    check(ctx.lib.gpx_var_grad(ctx.h, *spec.args(), L.h, X.h, Z.h, dptr(nd), dptr(eb), dptr(db), dptr(out)))
    return out


def var_grad_newpt(ctx, spec, L, X, Z):
    """(M*d,) vector d var(z_m) / d z_m (GP.evaluateVarianceDerivWRTnewpt, gp.py:261-280)."""
    out = np.empty(Z.shape[0] * spec.d)
    check(ctx.lib.gpx_var_grad_newpt(ctx.h, *spec.args(), L.h, X.h, Z.h, dptr(out)))
    return out


def kfill_plan(ctx, spec, X, Z=None):
    """(exact, center): which distance form an assembly between X and Z takes (test hook, gpx_dbg_kfill_plan)."""
    ex = C.c_int()
    cen = np.zeros(spec.d)
    check(ctx.lib.gpx_dbg_kfill_plan(ctx.h, *spec.args(), X.h, Z.h if Z is not None else None, C.byref(ex), dptr(cen)))
    return bool(ex.value), cen


def dbg_gemm_tri(ctx, A, B, Cm, bt, accumulate, tri):
    check(ctx.lib.gpx_dbg_gemm_tri(ctx.h, A.h, B.h, Cm.h, int(bt), int(accumulate), int(tri)))


def dbg_gemm_ksplit(ctx, A, B, Cm, mode, parts):
    check(ctx.lib.gpx_dbg_gemm_ksplit(ctx.h, A.h, B.h, Cm.h, int(mode), int(parts)))


def dbg_gemm(ctx, A, B, Cm, bt, accumulate, lower=False):
    check(ctx.lib.gpx_dbg_gemm(ctx.h, A.h, B.h, Cm.h, int(bt), int(accumulate), int(lower)))
