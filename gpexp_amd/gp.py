"""Gaussian-process model with the reference's interface (gpExp/gp.py, class GP), computed on the GPU.

What changes underneath: the reference builds K row by row in Python, inverts it with
`numpy.linalg.pinv` (SVD) and evaluates every posterior variance as a separate `k^T P k` matvec.  Here K is
assembled by one HIP kernel, factored once (K = L L^T, blocked fp64-MFMA Cholesky), and every later quantity is a
triangular solve or a column reduction against L:

    coeff = K^-1 y                     gpx_potrs        (gp.py:101)
    log det K = 2 sum log L_ii         gpx_logdet       (gp.py:434)
    mean = K(Z,X) coeff                gpx_posterior    (gp.py:137)
    var_j = k(z_j,z_j) - |L^-1 k_j|^2  gpx_posterior    (gp.py:142-144, 253-255)

`covarianceMatrix` and `precisionMatrix` remain available as attributes (external code reads them,
experimentalDesign.py:241-242) but are materialised lazily, on first access, from the device.

Rank-deficient K (e.g. noise 0.0 with coincident points): pinv truncates silently, Cholesky cannot.  Policy: a pivot
at round-off level (<= 1e-13 * largest diagonal entry) fails the factorisation, which is then repeated with those
points DROPPED (gpx_potrf_policy: unit pivot, zero column, zero row of the inverse -- every solve returns 0 in that
component, as if the point were not in the training set) and a RuntimeWarning says how many.  For an exactly duplicated
point with equal function values that is the pinv answer (posterior mean and variance to 1e-8, reference fixture
`rankdef`); `coeff` differs in how it is split between the duplicates (pinv: equal halves; here: all on the first).

FITC (gp.py:182-210, 401-426): `GP(kernel, noise, FITC=fraction)` draws the inducing points with
`np.random.permutation` exactly as the reference does (seed it the same way and the same points come out) and keeps
chol(Quu), Kuf, G and chol(Quu + Kuf G^-1 Kfu) on the device (gpx_fitc_*); predictions use the Woodbury form, the
N x N `covarianceMatrix` / `precisionMatrix` are only built when read.  The reference compares `self.fitcnodes == None`,
which raises for an ndarray, so there a GP instance survives only ONE FITC operation; here refits simply reuse
`fitcnodes`.  `generateSamples` stays out of scope.
"""
import copy
import warnings

import numpy as np

from . import device as _dev
from . import dist as _dist
from ._lib import NotPositiveDefinite, hdot as _hdot
from .gp_kernel_utilities import calculateCovarianceMatrix, _check_nugget  # noqa: F401  (re-export like the reference)

try:  # the reference prefers nlopt and falls back to SciPy (gp.py:28-41); only the SciPy branch is provided
    from scipy.optimize import fmin_l_bfgs_b as bfgs
except ImportError:  # pragma: no cover
    bfgs = None
NLOPT = False


class GP:
    """Zero-prior-mean GP regression model."""

    coeff = None
    noise = None
    pts = None
    FITC = None
    fitcnodes = None

    def __init__(self, kernel_in, noiseIn, **kwargs):
        try:
            self.kernel = copy.deepcopy(kernel_in)
        except Exception:
            print("warning ")
            self.kernel = copy.copy(kernel_in)
        self.noise = noiseIn  # added to the diagonal as a VARIANCE (gp.py:68, 178)
        if 'FITC' in kwargs:
            self.FITC = kwargs['FITC']  # fraction of the nodes used as inducing points (gp.py:69-70, 187)
        self._fitc = None    # device FITC model (gpx_fitc_*)
        self._X = None       # device point set
        self._Lc = None      # distributed-factor mode (dist.CyclicFactor): the factor lives block-cyclic on the ranks ...
        self._Ld = None      # ... and a dense factor is assembled only when something asks for `_L`
        self._L = None       # device Cholesky factor of K(pts) + nugget
        self._nugget = None  # nugget that went into _L
        self._K_host = None
        self._P_host = None
        self.jitter = 0.0    # kept for compatibility: the rank-deficient policy no longer perturbs the diagonal
        self.dropped = 0     # points dropped by the rank-deficient policy in the last factorisation
        # f2: the last factorisation, kept so that a refit whose leading points (and hyper-parameters) did
        # not change -- the design loop pins earlier points by bounds, experimentalDesign.py:722-724 -- only assembles and
        # factors the trailing rows.  (host copy of the points, nugget, hyper-parameter key, device factor)
        self._fcache = None
        self.reuseFactor = True

    # shallow copies (costFunctionGP_IVAR does copy.copy(gp), experimentalDesign.py:64) share device state,
    # which is immutable once built: a refit replaces the handles instead of mutating them.

    # ---- lazy dense attributes -------------------------------------------------------------------------
    @property
    def covarianceMatrix(self):
        if self._K_host is None and self._fitc is not None:
            self._K_host = self._fitc.dense(cov=True, prec=False)[0]
        elif self._K_host is None and self.pts is not None:
            self._K_host = calculateCovarianceMatrix(self.kernel, self.pts, self._nugget)
        return self._K_host

    @covarianceMatrix.setter
    def covarianceMatrix(self, value):
        self._K_host = value

    @property
    def precisionMatrix(self):
        if self._P_host is None and self._fitc is not None:
            self._P_host = self._fitc.dense(cov=False, prec=True)[1]
        elif self._P_host is None and self._has_factor():
            self._P_host = _dev.potri(_dev.context(), self._L).to_host(tri=2)   # lower triangle valid: mirrored
        return self._P_host

    @precisionMatrix.setter
    def precisionMatrix(self, value):
        self._P_host = value

    # the dense device factor.  Under a multi-process launch in the distributed-factor mode (dist.Session.use_cyclic) the fit
    # leaves a CyclicFactor instead: train / evaluate / evaluateVariance / the likelihood / the IVAR cost work on that, and a
    # dense replica is assembled the first time anything else reads `_L`.
    @property
    def _L(self):
        if self._Ld is None and self._Lc is not None:
            self._Ld = self._Lc.dense()
        return self._Ld

    @_L.setter
    def _L(self, value):
        if getattr(value, "is_cyclic", False):
            self._Lc, self._Ld = value, None
        else:
            self._Lc, self._Ld = None, value

    def _has_factor(self):
        return self._Lc is not None or self._Ld is not None

    # ---- helpers --------------------------------------------------------------------------------------------
    def gpPriorMean(self, pts):
        return np.zeros((pts.shape[0]))

    def _factor(self, nodes, nugget, remember=True):
        """Assemble K(nodes)+diag(nugget) and factor it in place on the device -> (X, L, 0.0).  `remember=False`
        (likelihood evaluations: every call has new hyper-parameters) neither consults nor replaces the kept factor."""
        _check_nugget(nugget)
        nodes = np.asarray(nodes, dtype=float)
        assert nodes.ndim == 2 and nodes.shape[1] == self.kernel.dimension, \
            (" Incorrect dimension of input points fed to kernel ", nodes.shape)
        ctx = _dev.context()
        spec = self.kernel._spec()
        X = _dev.points(ctx, nodes)
        # Rank-deficient policy (module docstring): pivots at round-off level count as failed, and the retry DROPS those
        # points, which is what the reference's pinv does with an exactly duplicated point (gp.py:181).
        diag = float(spec.hyp[-1]) if spec.kind != _dev.K_MEHLER else float(np.max(_dev.kdiag(ctx, spec, X)))
        tau = 1e-13 * (diag + float(np.max(np.asarray(nugget, dtype=float))))
        self.dropped = 0
        prev_policy = getattr(ctx, "_potrf_policy", (0.0, False))   # a policy the user set on the shared context survives
        try:
            _dev.potrf_policy(ctx, tau, False)
            self._last_refit = None    # (factor the last fit started from, rows it kept): costFunctionGP_IVAR updates its kept solve
            hit = self._cached_factor(nodes, nugget, spec)
            if hit is not None:
                return X, hit, 0.0
            keep = self._reusable_rows(nodes, nugget, spec) if remember else 0
            if keep > 0:
                try:
                    old = self._fcache[3]
                    L = _dev.refit_rows(ctx, spec, X, nugget, old, keep)
                    self._remember(nodes, nugget, spec, L)
                    self._last_refit = (old, keep)
                    return X, L, 0.0
                except NotPositiveDefinite:
                    pass  # fall through to the full path and its rank-deficient policy
            sess = _dist.session()
            if sess is not None and sess.use_fit(nodes.shape[0]):
                # multi-process launch: 2-D block-cyclic distributed fit, every rank keeps a replica of the factor
                # (dist.Session.factor).  A non-PD result is agreed by all ranks; they then all take the replicated
                # single-GPU path below with its rank-deficient policy.
                try:
                    Xd, L = sess.factor(spec, nodes, nugget, keep=remember)
                    if remember and not getattr(L, "is_cyclic", False):   # (rows are reused from a DENSE previous factor only)
                        self._remember(nodes, nugget, spec, L)
                    return Xd, L, 0.0
                except NotPositiveDefinite:
                    pass
            K = _dev.kfill(ctx, spec, X, nugget=nugget)
            try:
                L = _dev.potrf(ctx, K)
                if remember:
                    self._remember(nodes, nugget, spec, L)
                return X, L, 0.0
            except NotPositiveDefinite as first:
                if remember:
                    self._fcache = None
                _dev.kfill_into(ctx, spec, X, K, nugget=nugget)
                _dev.potrf_policy(ctx, tau, True)
                _dev.potrf(ctx, K)
                self.dropped = _dev.potrf_dropped(ctx)
                # once per (GP object, remember-mode): an optimiser loop over near-singular hyper-parameters would otherwise
                # raise it on every likelihood evaluation
                if not getattr(self, "_warned_dropped", {}).get(remember):
                    warnings.warn("covariance matrix not positive definite (pivot %d <= %.1e): %d point(s) whose "
                                  "conditional variance is at round-off level were dropped from the factor (the "
                                  "reference's pinv truncates the same directions)%s" %
                                  (first.pivot, tau, self.dropped,
                                   "" if remember else "; the log-likelihood returned is that of the REDUCED model -- the "
                                   "reference's value there is -0.5 y^T pinv(K) y - 0.5 slogdet(K) with slogdet -> -inf "
                                   "(gp.py:431-435), which no optimiser can use"), RuntimeWarning)
                    seen = dict(getattr(self, "_warned_dropped", {}))
                    seen[remember] = True
                    self._warned_dropped = seen
                return X, K, 0.0
        finally:
            _dev.potrf_policy(ctx, *prev_policy)

    @staticmethod
    def _spec_key(spec):
        return (spec.kind, spec.d, tuple(spec.hyp.tolist()))

    def _remember(self, nodes, nugget, spec, L):
        if self.reuseFactor and nodes.shape[0] >= 256:
            nug = None if np.ndim(nugget) == 0 else np.array(nugget, dtype=float, copy=True)
            self._fcache = (nodes.copy(), float(nugget) if nug is None else nug, self._spec_key(spec), L)
        else:
            self._fcache = None

    def _cached_factor(self, nodes, nugget, spec):
        """The kept factor itself when it IS the fit asked for: same kernel and hyper-parameters, bit-identical points, same
        nugget.  Read-only, so likelihood evaluations consult it too (computeLogLike right after train on the same points
        re-assembled and re-factored K: 2.4 of config C2's 6.3 ms)."""
        c = self._fcache
        if not self.reuseFactor or c is None or c[2] != self._spec_key(spec) or c[0].shape != nodes.shape:
            return None
        if np.ndim(nugget) == 0 and np.ndim(c[1]) == 0:
            if float(nugget) != c[1]:
                return None
        elif not np.array_equal(np.broadcast_to(np.asarray(nugget, dtype=float), (nodes.shape[0],)),
                                np.broadcast_to(np.asarray(c[1], dtype=float), (nodes.shape[0],))):
            return None
        return c[3] if np.array_equal(c[0], nodes) else None

    def _reusable_rows(self, nodes, nugget, spec):
        """Leading rows (multiple of 128) whose factor can be taken from the previous fit: same kernel and
        hyper-parameters, same nugget on those rows, bit-identical points.  0 = factor from scratch."""
        c = self._fcache
        if not self.reuseFactor or c is None or c[2] != self._spec_key(spec):
            return 0
        old, onug = c[0], c[1]
        m = min(old.shape[0], nodes.shape[0])
        if m < 128:
            return 0
        same = np.all(old[:m] == nodes[:m], axis=1)
        if np.ndim(nugget) == 0 and np.ndim(onug) == 0:
            if float(nugget) != onug:
                return 0
        else:
            a = np.broadcast_to(np.asarray(nugget, dtype=float), (nodes.shape[0],))[:m]
            b = np.broadcast_to(np.asarray(onug, dtype=float), (old.shape[0],))[:m]
            same = same & (a == b)
        p = m if same.all() else int(np.argmin(same))
        keep = (p // 128) * 128
        return keep if keep * 4 >= nodes.shape[0] else 0  # below a quarter of the rows the copy buys nothing

    # ---- training ---------------------------------------------------------------------------------------------
    def _fitc_model(self, nodes):
        """FITC branch of addNodesAndComputeCovariance / loglikeParams (gp.py:182-206, 401-426): inducing points are a
        random subset of the nodes, drawn once with np.random.permutation and then kept in `fitcnodes`."""
        nodes = np.asarray(nodes, dtype=float)
        assert nodes.ndim == 2 and nodes.shape[1] == self.kernel.dimension, \
            (" Incorrect dimension of input points fed to kernel ", nodes.shape)
        if self.fitcnodes is None:
            nNodes = len(nodes)
            nu = int(np.floor(nNodes * self.FITC))
            indu = np.random.permutation(nNodes)[0:nu]
            self.fitcnodes = np.array(nodes[indu], dtype=float)
        ctx = _dev.context()
        X = _dev.points(ctx, nodes)
        return X, _dev.FitcModel(ctx, self.kernel._spec(), X, _dev.points(ctx, self.fitcnodes), float(self.noise))

    def addNodesAndComputeCovariance(self, nodes, noiseIn=None):
        """Set the training locations and factor their covariance (no function values needed)."""
        if self.FITC is not None:
            if noiseIn is not None:
                print("NOT IMPLEMENTED YET")  # gp.py:208-209: per-point noise with FITC leaves the old state in place
            else:
                self._X, self._fitc = self._fitc_model(nodes)
                self._L = None
                self._nugget = self.noise
                self._K_host = None
                self._P_host = None
            self.pts = np.array(nodes, dtype=float, copy=True)
            return
        nugget = self.noise if noiseIn is None else noiseIn
        self._X, self._L, self.jitter = self._factor(nodes, nugget)
        self._nugget = nugget
        self._K_host = None
        self._P_host = None
        self.pts = np.array(nodes, dtype=float, copy=True)

    def train(self, pts, evalsIn, noiseIn=None):
        """Compute the GP coefficients K^-1 (y - prior mean); stored in `coeff`."""
        assert len(evalsIn.shape) == 1, "evaluations must be an (N,) array for training GP"
        evals = evalsIn - self.gpPriorMean(pts)
        self.addNodesAndComputeCovariance(pts, noiseIn)
        self.fVals = evals.copy()
        if self._fitc is not None:
            self.coeff = self._fitc.solve(evals)[0]
            return
        if self._Lc is not None:
            self.coeff = self._Lc.solve(evals)[0]          # distributed substitution on the block-cyclic factor
        else:
            self.coeff = _dev.potrs(_dev.context(), self._L, evals)

    # ---- prediction ---------------------------------------------------------------------------------------------
    def evaluate(self, newpt, compvar=0):
        """Posterior mean at `newpt`; compvar=1 also |variance| (gp.py:145), compvar=2 the full covariance."""
        assert newpt.shape[1] == self.kernel.dimension, "evaluation points for GP is incorrect shape"
        ctx = _dev.context()
        spec = self.kernel._spec()
        Z = _dev.points(ctx, newpt)
        if self._fitc is not None:
            mean, var = self._fitc.posterior(self.coeff, Z, want_mean=True, want_var=(compvar == 1))
            out = mean + self.gpPriorMean(newpt)
            if compvar == 1:
                return out, np.abs(var)
            elif compvar == 2:  # dense by definition (gp.py:146-152); built from the materialised precision
                kv = _dev.kfill(ctx, spec, Z, Z=self._X).to_host()
                return out, _dev.kfill(ctx, spec, Z, Z=Z).to_host() - kv @ (self.precisionMatrix @ kv.T)
            return out
        sess = _dist.session()
        if sess is not None and (self._Lc is not None or sess.use_eval(newpt.shape[0])):
            # evaluation points sharded over the ranks, mean and variance gathered on every rank (the role of
            # parallelizeMcForLoop, parallel_utilities.py:26-80)
            mean, var = sess.posterior(spec, self._Lc or self._L, self._X, self.coeff, newpt, want_mean=True,
                                       want_var=(compvar == 1))
        else:
            mean, var = _dev.posterior(ctx, spec, self._L, self._X, self.coeff, Z, want_mean=True,
                                       want_var=(compvar == 1))
        out = mean + self.gpPriorMean(newpt)
        if compvar == 1:
            return out, np.abs(var)
        elif compvar == 2:
            return out, _dev.posterior_cov(ctx, spec, self._L, self._X, Z)
        return out

    def evaluateVariance(self, newpt, parallel=1):
        """Signed posterior variance at `newpt` (gp.py:213-259).  `parallel` is accepted and ignored: the
        reference forks CPU processes above 500001 points (gp.py:244-258); here the parallel branch is the multi-process
        launch itself -- one process per GPU, evaluation points sharded, the variance vector gathered on every rank."""
        assert self.pts is not None
        assert newpt.shape[1] == self.kernel.dimension, "evaluation points for GP is incorrect shape"
        ctx = _dev.context()
        if self._fitc is not None:
            return self._fitc.posterior(None, _dev.points(ctx, newpt), want_mean=False, want_var=True)[1]
        sess = _dist.session()
        if sess is not None and (self._Lc is not None or sess.use_eval(newpt.shape[0])):   # gp.py:244-258: the reference's parallel branch
            return sess.posterior(self.kernel._spec(), self._Lc or self._L, self._X, None, newpt, want_mean=False,
                                  want_var=True)[1]
        _, var = _dev.posterior(ctx, self.kernel._spec(), self._L, self._X, None, _dev.points(ctx, newpt),
                                want_mean=False, want_var=True)
        return var

    # ---- variance gradients w.r.t. point locations, on the device ---------------------------------------
    def _point_derivative_ready(self, newpt):
        assert self.pts is not None, "must specify training points before running this"
        assert newpt.shape[1] == self.kernel.dimension, "evaluation points for GP is incorrect shape"
        if not self._has_factor() and self._fitc is None:
            raise NotImplementedError("variance derivatives need a fitted model")
        if not hasattr(self.kernel, "derivative") or (self.kernel._spec().kind == _dev.K_MEHLER
                                                      and self.kernel.dimension != 1):
            # the reference defines Kernel.derivative for the squared exponential and the 1-D Mehler kernel only
            raise AttributeError("derivative of %s not implemented" % type(self.kernel).__name__)

    def evaluateVarianceDerivWRTnewpt(self, newpt):
        """d var(newpt_i) / d newpt_i, flattened (gp.py:261-280): gpx_var_grad_newpt."""
        self._point_derivative_ready(newpt)
        ctx = _dev.context()
        if self._fitc is not None:    # the reference reads `precisionMatrix`: for FITC the Woodbury precision (gp.py:204-206)
            return self._fitc.var_grad_newpt(self.kernel._spec(), _dev.points(ctx, newpt))
        return _dev.var_grad_newpt(ctx, self.kernel._spec(), self._L, self._X, _dev.points(ctx, newpt))

    def evaluateVarianceDerivative(self, newpt, noiseFunc=None):
        """out[k*d+l, j] = d var(newpt_j) / d pts[k, l]  (gp.py:282-341): gpx_var_grad -- beta = K^-1 K(X,Z) by two
        triangular solves, one MFMA GEMM per coordinate, fused finish; the (N*d, M) result is the only thing that
        crosses to the host.  `noiseFunc` (callable with .deriv, demo2.py:45-58) adds its terms at coincident training
        points (gp.py:314-317) and, when the WHOLE evaluation set coincides with a training point, the terms of
        gp.py:318-320 (the reference tests `np.linalg.norm(p - newpt)`, a norm over all evaluation points)."""
        self._point_derivative_ready(newpt)
        ctx = _dev.context()
        nd = eb = db = None
        if noiseFunc is not None:
            nd = np.asarray(noiseFunc.deriv(self.pts), dtype=float).reshape(self.pts.shape)
            spread = newpt - newpt[:1]
            if np.linalg.norm(spread) < 2e-10:       # otherwise no training point can be within 1e-10 of all of them
                hit = np.array([np.linalg.norm(self.pts[zz:zz + 1] - newpt) < 1e-10 for zz in range(len(self.pts))])
                if hit.any():
                    eb = np.where(hit, np.asarray(noiseFunc(self.pts), dtype=float), 0.0)
                    db = np.where(hit[:, None], nd, 0.0)
        if self._fitc is not None:
            return self._fitc.var_grad(self.kernel._spec(), _dev.points(ctx, newpt), nd, eb, db)
        return _dev.var_grad(ctx, self.kernel._spec(), self._L, self._X, _dev.points(ctx, newpt), nd, eb, db)

    def generateSamples(self, x, noise=1e-10):
        raise NotImplementedError("generateSamples (SVD sampling, gp.py:343-371) is outside the GPU hot path")

    # ---- marginal likelihood -------------------------------------------------------------------------------------
    def computeLogLike(self, pts, evals):
        """Log marginal likelihood of (pts, evals) under the current hyper-parameters."""
        return self.loglikeParams(pts, evals)

    def loglikeParams(self, pts, evals, returnDeriv=0, noiseIn=None):
        """-1/2 y^T K^-1 y - 1/2 log det K - N/2 log 2 pi  [, {key: d/d key}] (gp.py:394-468).

        Does not touch the trained state.  `noiseIn` (per-point nugget) works here; in the reference that branch
        passes an unknown keyword and raises TypeError (gp.py:429-430)."""
        evals = np.asarray(evals, dtype=float)
        if self.FITC is not None and noiseIn is None:
            if returnDeriv == 1:
                raise NotImplementedError("no hyper-parameter gradient for the FITC likelihood (the reference's own is "
                                          "unrunnable, kernels.py:140-141)")
            _, model = self._fitc_model(pts)
            quad = model.solve(evals)[1]
            return -0.5 * quad - 0.5 * model.logdet() - len(evals) / 2.0 * np.log(2.0 * np.pi)
        nugget = self.noise if noiseIn is None else noiseIn
        X, L, _ = self._factor(pts, nugget, remember=False)
        ctx = _dev.context()
        if getattr(L, "is_cyclic", False):
            alpha, logdet = L.solve(evals)                 # distributed substitution; a dense factor only for the gradient
            if returnDeriv == 1:
                L = L.dense()
        else:
            alpha = _dev.potrs(ctx, L, evals)
            logdet = _dev.logdet(ctx, L)
        out = -0.5 * _hdot(evals, alpha) - 0.5 * logdet - len(evals) / 2.0 * np.log(2.0 * np.pi)
        if returnDeriv == 1:
            keys = list(self.kernel.hyperParam.keys()) + ['noise']
            sess = _dist.session()
            if sess is not None and sess.use_fit(len(evals)):     # traces sharded by row slabs of K^-1 (dist_lml_grad)
                grad = sess.lml_grad(self.kernel._spec(), L, X, alpha)
            else:
                grad = _dev.lml_grad(ctx, self.kernel._spec(), L, X, alpha)
            outD = dict(zip(keys, grad))
            outD['noise'] *= self.noise * 2.0  # gp.py:463-464
            return out, outD
        return out

    def getHypParamNames(self):
        return self.kernel.hyperParam.keys()

    def updateKernelParams(self, paramsIn):
        """Set new hyper-parameters; a 'noise' entry goes to `self.noise`, the rest to the kernel."""
        params = copy.copy(paramsIn)
        if 'noise' in params.keys():
            self.noise = copy.copy(params['noise'])
            del params['noise']
        self.kernel.updateHyperParameters(params)

    # ---- hyper-parameter fit: host driver around loglikeParams -----------------------------------
    def findOptParamsLogLike(self, pts, evals, paramsStart=None, paramLowerBounds=None, paramUpperBounds=None,
                             useNoise=None, maxiter=40, useLastParams=True, analyticGradient=False):
        """Maximise the marginal likelihood over the kernel hyper-parameters (+ noise unless `useNoise` is given);
        bounds default to [max(v/10, 1e-3), min(10 v, 10)], noise to [1e-12, 1] from 1e-5 (gp.py:498-590).

        `analyticGradient=True` (opt-in; squared-exponential kernel) hands L-BFGS-B the gradient from
        gpx_lml_grad instead of letting it difference the objective (gp.py:635, approx_grad=True): one factorisation per
        iterate instead of nparams+1.  The default reproduces the reference's numerical-gradient search."""
        if paramsStart is None:
            paramsStart = copy.deepcopy(self.kernel.hyperParam)
        if paramLowerBounds is None:
            paramLowerBounds = dict((k, np.max([v / 10.0, 1e-3])) for k, v in paramsStart.items())
        if paramUpperBounds is None:
            paramUpperBounds = dict((k, np.min([v * 10.0, 10.0])) for k, v in paramsStart.items())
        keys = list(paramsStart.keys())
        vals = [paramsStart[k] for k in keys]
        lbs = [paramLowerBounds[k] for k in keys]
        ubs = [paramUpperBounds[k] for k in keys]
        if useNoise is None:  # last key is the noise
            keys.append('noise')
            lbs.append(1e-12)
            ubs.append(1e0)
            vals.append(1e-5)

        def objFunc(in0, gradIn):
            self.updateKernelParams(dict(zip(keys, in0)))
            if gradIn.size > 0:
                margLogLike, derivs = self.loglikeParams(pts, evals, returnDeriv=1)
                if analyticGradient and 'noise' in derivs and 'noise' in keys:
                    # loglikeParams scales the noise entry by 2*noise (gp.py:463-464, a derivative w.r.t. sqrt(noise));
                    # the search variable is the noise VARIANCE, so the optimiser gets the unscaled trace term
                    derivs['noise'] /= 2.0 * self.noise
                gradIn[:] = -np.array([derivs[k] for k in keys])
            else:
                margLogLike = self.loglikeParams(pts, evals, returnDeriv=0)
            objFunc.last_x_value = in0.copy()
            objFunc.last_f_value = -margLogLike
            return -margLogLike

        paramsOut, optValue = self.chooseParams(lbs, ubs, vals, objFunc, maxiter=maxiter, useLastParams=useLastParams,
                                                analyticGradient=analyticGradient)
        params = dict(zip(keys, paramsOut))
        self.updateKernelParams(params)
        return params, optValue

    def chooseParams(self, paramLowerBounds, paramUpperBounds, startValues, costFunction, maxiter=40,
                     useLastParams=True, analyticGradient=False):
        """SciPy L-BFGS-B, factr=1e10, maxfun=maxiter (gp.py:615-639): numerical gradients as in the reference, or the
        cost function's own gradient when `analyticGradient` is set."""
        bounds = list(zip(paramLowerBounds, paramUpperBounds))

        def objFunc(x):
            return costFunction(x, np.empty(0))

        def objFuncWithGrad(x):
            g = np.empty(len(x))
            f = costFunction(x, g)
            return f, g

        if analyticGradient:
            sol = bfgs(objFuncWithGrad, np.array(startValues), bounds=bounds, factr=1e10, maxfun=maxiter)[0]
        else:
            sol = bfgs(objFunc, np.array(startValues), bounds=bounds, approx_grad=True, factr=1e10, maxfun=maxiter)[0]
        return sol, objFunc(sol)
