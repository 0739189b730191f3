"""TEST-ONLY NumPy double of `gpexp_amd.device` for running the GPEXP CLASS API (gpexp_amd.gp / experimentalDesign) through
the distributed session (gpexp_amd.dist.Session) on CPU over gloo: the worker swaps the `_dev` name of the product modules for
an instance of `NumpyDevice`, so GP.train / evaluate / computeLogLike / the design functions run their real host logic --
routing, sharding, gathers, first-min / first-max merges, the SPMD agreement check -- against dense NumPy arithmetic built on
the oracle's kernel functions.  Nothing here ships: the product has no CPU path (tests/test_host_cpu.py checks that).
"""
import numpy as np

from gpexp_amd import device as _real
from gpexp_amd import dist
from gpexp_amd._lib import NotPositiveDefinite
from dist_worker import NumpyMat, NumpyOps2D, NumpyC5   # noqa: E402  (tests/)
from oracle import gpexp_oracle as orc

KIND_NAME = {0: "se", 1: "matern32", 2: "matern52", 3: "mehler"}


def oracle_spec(spec):
    """KernelSpec (kind, d, hyp) -> the oracle's dict."""
    d, hyp = spec.d, np.asarray(spec.hyp, dtype=float)
    kind = KIND_NAME[spec.kind]
    if kind == "se":
        return dict(kind="se", cl=[float(v) for v in hyp[:d]], signalSize=float(hyp[d]), d=d)
    if kind in ("matern32", "matern52"):
        return dict(kind=kind, rho=float(hyp[0]), signalSize=float(hyp[1]), d=d)
    return dict(kind="mehler", t=[float(v) for v in hyp[:d]], d=d)


def _cov(spec, X, nugget):
    n = X.shape[0]
    if n == 0:
        return np.zeros((0, 0))
    nz = np.broadcast_to(np.asarray(nugget, dtype=float), (n,)) if np.ndim(nugget) else float(nugget) * np.ones(n)
    return orc.cov_matrix(oracle_spec(spec), X, 0.0, row_loop=False) + np.diag(nz)


class _Ctx:
    lib = None
    h = 1
    device = 0

    def sync(self):
        pass


class _HostMat:
    def __init__(self, a):
        self.a = a

    def to_host(self, tri=0):
        return self.a.copy()


class NumpyDevice(NumpyC5):
    """The functions of gpexp_amd.device the class API calls, on host arrays.  A factor is a NumpyMat whose leading n x n block is
    the lower Cholesky factor (the replica the distributed double assembles is padded with an identity block, like the device's)."""
    K_SE, K_MATERN32, K_MATERN52, K_MEHLER = _real.K_SE, _real.K_MATERN32, _real.K_MATERN52, _real.K_MEHLER
    KernelSpec = _real.KernelSpec
    LML_GRAD_SLAB_MIN = _real.LML_GRAD_SLAB_MIN

    def __init__(self):
        super().__init__()
        self._ctx = _Ctx()
        self.calls = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def context(self):
        return self._ctx

    def points(self, ctx, x):
        x = np.array(x, dtype=float)
        assert x.ndim == 2
        return x

    def clone(self, ctx, M):
        return NumpyMat(M.a.copy())

    def kdiag(self, ctx, spec, Z):
        return orc.kernel_diag(oracle_spec(spec), Z)

    def kernel_eval(self, ctx, spec, x1, x2):
        return orc.kernel_eval(oracle_spec(spec), np.asarray(x1, float), np.asarray(x2, float))

    def kfill(self, ctx, spec, X, Z=None, nugget=0.0):
        if Z is None:
            return NumpyMat(_cov(spec, X, nugget))
        return NumpyMat(orc.cross_matrix(oracle_spec(spec), Z, X).T)

    def kfill_into(self, ctx, spec, X, K, Z=None, nugget=0.0):
        K.a = self.kfill(ctx, spec, X, Z=Z, nugget=nugget).a
        return K

    def potrf_policy(self, ctx, piv_min=0.0, skip=False):
        prev = getattr(ctx, "_potrf_policy", (0.0, False))
        ctx._potrf_policy = (float(piv_min), bool(skip))
        return prev

    def potrf_dropped(self, ctx):
        return 0

    def potrf(self, ctx, K):
        self._count("potrf")
        try:
            K.a = np.linalg.cholesky(K.a)
        except np.linalg.LinAlgError:
            raise NotPositiveDefinite(1)
        return K

    def refit_rows(self, ctx, spec, X, nugget, L_old, keep):
        return self.potrf(ctx, NumpyMat(_cov(spec, X, nugget)))

    @staticmethod
    def _L(L, n):
        Lt = np.tril(L.a[:n, :n])
        assert not np.isnan(Lt).any(), "the factor handed to the class API is incomplete"
        return Lt

    def potrs(self, ctx, L, y):
        y = np.asarray(y, dtype=float)
        Lt = self._L(L, len(y))
        return np.linalg.solve(Lt.T, np.linalg.solve(Lt, y))

    def logdet(self, ctx, L):
        return 2.0 * float(np.sum(np.log(np.diag(L.a))))

    def potri(self, ctx, L):
        n = L.a.shape[0]
        Lt = self._L(L, n)
        return _HostMat(np.linalg.inv(Lt @ Lt.T))

    def posterior(self, ctx, spec, L, X, alpha, Z, want_mean=True, want_var=True):
        self._count("posterior")
        s = oracle_spec(spec)
        n = X.shape[0]
        kz = orc.cross_matrix(s, Z, X)                      # (M, N)
        mean = kz @ np.asarray(alpha, float) if (want_mean and alpha is not None) else None
        var = None
        if want_var:
            W = np.linalg.solve(self._L(L, n), kz.T)
            var = orc.kernel_diag(s, Z) - np.sum(W * W, axis=0)
        return mean, var

    def posterior_cov(self, ctx, spec, L, X, Z):
        s = oracle_spec(spec)
        W = np.linalg.solve(self._L(L, X.shape[0]), orc.cross_matrix(s, Z, X).T)
        kzz = np.array([orc.kernel_eval(s, Z, Z[j:j + 1]) for j in range(len(Z))]).T
        return kzz - W.T @ W

    def ivar(self, ctx, spec, L, X, Z):
        self._count("ivar")
        return float(np.mean(self.posterior(ctx, spec, L, X, None, Z, want_mean=False)[1]))

    def lml_grad(self, ctx, spec, L, X, alpha, slabs=None):
        sums = self.lml_grad_slab(ctx, spec, L, X, np.asarray(alpha, float), 0, dist.padded(X.shape[0]))
        return self.lml_grad_from_sums(spec, sums)

    def greedy_var(self, ctx, spec, Cpts, nsel, keep=(), weights=None):
        self._count("greedy_var")
        return np.array(orc.greedy_var(oracle_spec(spec), Cpts, int(nsel), weights=weights, keep_start=list(keep)), dtype=np.int64)

    def greedy_ivar_step(self, ctx, spec, L, X, Cpts, Z, noise, want_costs=True):
        self._count("greedy_ivar_step")
        s = oracle_spec(spec)
        costs = np.array([orc.ivar(s, np.vstack((X, Cpts[j:j + 1])), Z, float(noise)) for j in range(len(Cpts))])
        return int(np.argmin(costs)), costs

    def GivarState(self, ctx, spec, L, X, Cpts, Z, noise, nsel):
        return _GivarState(self, spec, L, X, Cpts, Z, noise, nsel)

    def greedy_ivar(self, ctx, spec, L, X, Cpts, Z, noise, nsel, want_all=False):
        st = _GivarState(self, spec, L, X, Cpts, Z, noise, nsel)
        buf = NumpyMat(np.zeros(st.pivot_elems))
        idx, costs = [], []
        for t in range(int(nsel)):
            c, i, _ = st.score()
            idx.append(i)
            costs.append(c)
            if t + 1 < nsel:
                st.pack(i, buf)
                st.apply(buf)
        return np.array(idx, dtype=np.int64), np.array(costs)

    def mi_greedy(self, ctx, spec, Cpts, noise, nsel, start=0):
        idx, ratios = orc.greedy_mi(oracle_spec(spec), Cpts, float(noise), int(nsel), start=int(start))
        return np.array(idx, dtype=np.int64), ratios


class _GivarState:
    """NumPy double of device.GivarState: same interface and pivot-pack layout, dense arithmetic from the definitions."""

    def __init__(self, be, spec, L, X, Cpts, Z, noise, nsel):
        s = oracle_spec(spec)
        self.s, self.C, self.noise, self.nsel, self.cur = s, Cpts, float(noise), int(nsel), 0
        n = X.shape[0]
        Lt = be._L(L, n)
        self.np_, self.zp, self.d = dist.padded(n), dist.padded(len(Z)), spec.d
        self.Wc = np.zeros((self.np_, len(Cpts)))
        self.Wc[:n] = np.linalg.solve(Lt, orc.cross_matrix(s, Cpts, X).T)
        Wz = np.linalg.solve(Lt, orc.cross_matrix(s, Z, X).T)
        self.G = np.zeros((self.zp, len(Cpts)))
        self.G[:len(Z)] = orc.cross_matrix(s, Cpts, Z).T - Wz.T @ self.Wc[:n]
        self.v = orc.kernel_diag(s, Cpts) - np.sum(self.Wc ** 2, axis=0)
        self.s0 = float(np.sum(orc.kernel_diag(s, Z) - np.sum(Wz ** 2, axis=0)))
        self.nmc = len(Z)
        self.U = np.zeros((self.nsel, len(Cpts)))
        self.pivot_elems = 2 + self.d + self.zp + self.np_ + self.nsel

    def score(self, want_all=False):
        den = self.v + self.noise
        with np.errstate(divide="ignore", invalid="ignore"):
            cost = np.where(den > 1e-300, np.abs((self.s0 - np.sum(self.G ** 2, axis=0) / den) / self.nmc), np.inf)
        cost = np.where(np.isnan(cost), np.inf, cost)           # (gpx_givar_score: an undefined cost leaves the race)
        i = int(np.argmin(cost))
        return float(cost[i]), i, (cost if want_all else None)

    def pack(self, sidx, buf):
        d, zp, np_ = self.d, self.zp, self.np_
        delta = self.v[sidx] + self.noise
        b = buf.a
        b[0] = delta
        b[2:2 + d] = self.C[sidx]
        r = self.G[:, sidx] / np.sqrt(delta)
        b[1] = float(r @ r)
        b[2 + d:2 + d + zp] = r
        b[2 + d + zp:2 + d + zp + np_] = self.Wc[:, sidx]
        b[2 + d + zp + np_:2 + d + zp + np_ + self.nsel] = self.U[:, sidx]

    def apply(self, buf):
        d, zp, np_ = self.d, self.zp, self.np_
        b = buf.a
        delta, cs = b[0], b[2:2 + d].reshape(1, d)
        r, w, uc = b[2 + d:2 + d + zp], b[2 + d + zp:2 + d + zp + np_], b[2 + d + zp + np_:2 + d + zp + np_ + self.nsel]
        u = (orc.kernel_eval(self.s, self.C, cs) - w @ self.Wc - uc[:self.cur] @ self.U[:self.cur]) / np.sqrt(delta)
        self.U[self.cur] = u
        self.v = self.v - u * u
        self.G = self.G - np.outer(r, u)
        self.s0 -= b[1]
        self.cur += 1


class ApiOps2D(NumpyOps2D):
    """NumpyOps2D whose assembly takes the KernelSpec / nugget of the call (the class API changes both between fits)."""

    def __init__(self):
        super().__init__(None)

    # the evaluation against a block-cyclic factor (DistFitIvar2D.cyclic_posterior) takes the kernel of the call too
    def cross_fill(self, spec, X, Z, B):
        B.a[:] = 0.0
        B.a[:X.shape[0], :] = orc.cross_matrix(oracle_spec(spec), Z, X).T

    def cross_mean(self, spec, X, Z, alpha):
        return orc.cross_matrix(oracle_spec(spec), Z, X) @ np.asarray(alpha, dtype=float)

    def variances(self, spec, Z, B, n):
        return orc.kernel_diag(oracle_spec(spec), Z) - np.sum(B.a[:n] ** 2, axis=0)

    def kfill_local(self, spec, X, A, nugget, geo):
        n, nb = X.shape[0], geo.nb
        full = np.eye(geo.np)
        full[:n, :n] = _cov(spec, X, nugget)
        A.a[:] = np.nan
        for I in range(geo.pr, geo.nblk, geo.Pr):
            for J in range(geo.pc, geo.nblk, geo.Pc):
                if I >= J:
                    A.a[(I // geo.Pr) * nb:(I // geo.Pr) * nb + geo.height(I),
                        (J // geo.Pc) * nb:(J // geo.Pc) * nb + geo.height(J)] = \
                        full[I * nb:I * nb + geo.height(I), J * nb:J * nb + geo.height(J)]
