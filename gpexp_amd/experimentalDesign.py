"""Experimental-design cost functions and greedy designs with the reference's interface
(gpExp/experimentalDesign.py), evaluated on the GPU.

Hot-path pieces:
  costFunctionGP_IVAR.evaluate        refit on the design + mean posterior variance over the MC points
                                      (experimentalDesign.py:79-117)            -> gpx_potrf + gpx_ivar
  performGreedyVarExperimentalDesign  greedy maximum posterior variance (:787-845) -> gpx_greedy_var
  costFunctionGP_MI / performGreedyMIExperimentalDesign (:223-285, :753-785)       -> gpx_mi_greedy
  greedyIVARStep                      discrete one-step-lookahead IVAR (the composition SURVEY.md 8c pins;
                                      the reference has no such function)        -> gpx_greedy_ivar_step

Host drivers kept because demo.py needs them: ExperimentalDesign.boundsFunction,
ExperimentalDesignDerivative.begin / beginWithVarGreedy (SciPy SLSQP branch).  The Mercer-basis "version 0" of
IVAR is dead code in the reference (no kernel defines `eigenvalues`, :122,129,193) and is not provided; the
COBYLA / greedy-continuation drivers (:500-751) and the space-filling / Bayesian-optimisation helpers (:852-1003)
are outside the hot path.
"""
import copy
import itertools
import sys

import numpy as np

try:
    from scipy.optimize import fmin_slsqp as slsqp
except ImportError:  # pragma: no cover
    slsqp = None
import scipy.optimize as optimize

from . import device as _dev
from . import dist as _dist
from . import gp_kernel_utilities  # noqa: F401  (the reference module exposes it)

NLOPT = False

__all__ = ["costFunctionBase", "costFunctionGP_IVAR", "costFunctionGP_MI", "ExperimentalDesign",
           "ExperimentalDesignDerivative", "performGreedyVarExperimentalDesign",
           "performGreedyMIExperimentalDesign", "greedyIVARStep", "performGreedyIVARExperimentalDesign", "np"]


class costFunctionBase(object):

    def __init__(self, nInputs, space):
        self.numInputs = nInputs
        self.space = space


class costFunctionGP_IVAR(costFunctionBase):
    """Integrated posterior variance of a GP trained at the design points, estimated over MC points."""

    def __init__(self, gaussianProcess, nInputs, space, version=1, **kwargs):
        super(costFunctionGP_IVAR, self).__init__(nInputs, space)
        self.gaussianProcess = copy.copy(gaussianProcess)
        self.version = version
        if self.version != 1:
            raise NotImplementedError("IVAR version 0 needs a Mercer eigen-basis that no kernel class provides "
                                      "(dead code in the reference, experimentalDesign.py:119-146)")
        self._mc_dev = None  # MC points stay resident in HBM across optimiser evaluations
        if 'mcPoints' in kwargs:
            self.mcPoints = kwargs['mcPoints']
        else:
            self.mcPoints = space.sample((10000, space.dimension))

    # `mcPoints` is an attribute users reassign (the reference's drivers do): everything derived from the set -- its device copy, the
    # forward solve W kept for the gradient (N x M doubles: 2 GB at N = 8192, M = 32768) and the cost that goes with it -- is
    # dropped with it (the kept cost / gradient would otherwise belong to the previous set).
    @property
    def mcPoints(self):
        return self._mcPoints

    @mcPoints.setter
    def mcPoints(self, pts):
        self._mcPoints = pts
        self.nMC = len(pts)
        self.reset()

    def reset(self):
        """Drop the device state kept across optimiser iterations (MC points, forward solve, cost); the batch driver calls it
        when a batch ends."""
        self._mc_dev = None
        self._w_kept = None

    def _mc(self):
        if self._mc_dev is None:
            self._mc_dev = _dev.points(_dev.context(), self._mcPoints)
        return self._mc_dev

    def evaluate(self, inputPoints):
        """|mean_z var(z | design = inputPoints)| (experimentalDesign.py:100-117)."""
        assert inputPoints.shape == (self.numInputs, self.space.dimension), \
            ("inputPoints are the wrong size: ", inputPoints.shape)
        gp = self.gaussianProcess
        if self.space.noiseFunc is None:
            gp.addNodesAndComputeCovariance(inputPoints)
        else:
            gp.addNodesAndComputeCovariance(inputPoints, self.space.noiseFunc(inputPoints))
        if not gp._has_factor():   # FITC model: the same mean through the Woodbury precision (gp.py:246-255), as the reference
            return np.abs(np.mean(gp.evaluateVariance(self.mcPoints)))
        sess = _dist.session()
        if sess is not None and (gp._Lc is not None or sess.use_eval(self.nMC)):   # MC points sharded over the ranks, partial sums in rank order
            return np.abs(sess.ivar(gp.kernel._spec(), gp._Lc or gp._L, gp._X, self.mcPoints, cache=self))
        # An optimiser asks for the cost and then for its gradient at the same design (SLSQP: experimentalDesign.py:471-489):
        # from 1024 design points on the forward solve W = L^-1 K(X, Z) stays on the device for that gradient -- a third of
        # its work (design state across optimiser iterations).  Keyed on the factor OBJECT: the same design
        # gets the kept factor back from GP._cached_factor; one W at a time, (factor, W, signed cost).
        # ... and a batch loop moves the last points only (experimentalDesign.py:694-751): when the refit kept the leading rows of
        # the factor W belongs to, W keeps them too and only its trailing rows are solved again (gpx_ivar_update).
        prev, self._w_kept = getattr(self, "_w_kept", None), None
        if inputPoints.shape[0] >= 1024 and gp.reuseFactor and self.space.noiseFunc is None:
            ctx, spec = _dev.context(), gp.kernel._spec()
            lr = getattr(gp, "_last_refit", None)
            if prev is not None and prev[0] is gp._L:                  # the very fit W and the cost belong to
                self._w_kept = prev
                return np.abs(prev[2])
            if prev is not None and lr is not None and lr[0] is prev[0] and prev[1].shape[0] == inputPoints.shape[0]:
                cost = _dev.ivar_update(ctx, spec, gp._L, gp._X, self._mc(), prev[1], lr[1])
                self._w_kept = (gp._L, prev[1], cost)
                return np.abs(cost)
            del prev                                                   # (its W goes back to the pool before the next one is made)
            cost, W = _dev.ivar(ctx, spec, gp._L, gp._X, self._mc(), keep=True)
            if W is not None:
                self._w_kept = (gp._L, W, cost)
        else:
            cost = _dev.ivar(_dev.context(), gp.kernel._spec(), gp._L, gp._X, self._mc())
        return np.abs(cost)

    def derivative(self, inputPoints):
        """d IVAR / d design coordinates, flattened (experimentalDesign.py:168-179): gpx_ivar_grad --
        two triangular solves, one MFMA GEMM and a fused row reduction instead of the reference's (N*d x M) matrix --
        for the kernels the reference differentiates (squared exponential, 1-D Mehler), with the heteroscedastic terms of
        `space.noiseFunc` when there is one.  With `pinnedPoints` set (the batch driver: the leading points are fixed by equal
        bounds) the entries of the pinned points are returned as ZEROS -- not the full gradient."""
        gp = self.gaussianProcess
        nd = None
        if self.space.noiseFunc is None:
            gp.addNodesAndComputeCovariance(inputPoints)
        else:
            gp.addNodesAndComputeCovariance(inputPoints, noiseIn=self.space.noiseFunc(inputPoints))
            nd = np.asarray(self.space.noiseFunc.deriv(inputPoints), dtype=float).reshape(inputPoints.shape)
        gp._point_derivative_ready(self.mcPoints)
        if gp._fitc is not None:
            # FITC model: the reference sums evaluateVarianceDerivative over the MC points (experimentalDesign.py:171-177), which
            # reads the Woodbury precision (gp.py:322); gpx_fitc_var_grad, then the same mean over the MC points
            dv = gp._fitc.var_grad(gp.kernel._spec(), self._mc(), nd)
            return np.sum(dv, axis=1) / float(len(self.mcPoints))
        kept = getattr(self, "_w_kept", None)
        W = kept[1] if (kept is not None and nd is None and kept[0] is gp._L) else None    # (read only: it stays for the next cost)
        # `pinnedPoints` (set by the batch driver, experimentalDesign.py:719-724: the leading points are fixed by equal bounds, so
        # the optimiser never uses their entries): the gradient of the free points alone from the kept solve -- gpx_ivar_grad_rows,
        # 2 (N - r0) N M flops instead of 2 N^2 M; the pinned entries are returned as zeros
        r0 = (int(getattr(self, "pinnedPoints", 0)) // 128) * 128
        if W is not None and r0 > 0 and gp.kernel._spec().kind == _dev.K_SE:
            g = np.zeros(inputPoints.shape[0] * self.space.dimension)
            g[r0 * self.space.dimension:] = _dev.ivar_grad_rows(_dev.context(), gp.kernel._spec(), gp._L, gp._X, self._mc(), W, r0)
            return g
        return _dev.ivar_grad(_dev.context(), gp.kernel._spec(), gp._L, gp._X, self._mc(), nd, W=W)


class costFunctionGP_MI(costFunctionBase):
    """Mutual-information ratio var(c | A) / var(c | all \\ A \\ c) among a fixed candidate set."""

    def __init__(self, gaussianProcess, nInputs, space, nmc=None, mcpoints=None, square=False):
        super(costFunctionGP_MI, self).__init__(nInputs, space)
        self.gaussianProcess = gaussianProcess  # NOT copied: the reference mutates the caller's GP (:227,240)
        if nmc is not None:
            self.nMC = nmc
            self.mcPoints = np.copy(mcpoints)
        elif space.dimension == 2 and square is True:
            x = np.linspace(-1, 1, 10)
            self.nMC = len(x) * len(x)
            self.mcPoints = np.array(list(itertools.product(x, x)))
        else:
            self.nMC = 200
            self.mcPoints = space.sample((self.nMC, space.dimension))
        self.gaussianProcess.addNodesAndComputeCovariance(self.mcPoints)

    def add_candidates(self, nCandidates, candidates):
        self.nMC = nCandidates
        self.mcPoints = copy.deepcopy(candidates)
        self.gaussianProcess.addNodesAndComputeCovariance(self.mcPoints)

    # the reference caches dense copies that evaluate() never uses (:241-242); kept as lazy views
    @property
    def cov(self):
        return self.gaussianProcess.covarianceMatrix

    @property
    def invcov(self):
        return self.gaussianProcess.precisionMatrix

    def _cond_var(self, rows, point):
        """var(f(point) | noisy observations at mcPoints[rows]) through a fresh factorisation."""
        gp = self.gaussianProcess
        ctx = _dev.context()
        spec = gp.kernel._spec()
        if len(rows) == 0:
            return _dev.kernel_eval(ctx, spec, point, point)[0]
        X, L, _ = gp._factor(self.mcPoints[rows, :], gp.noise)
        _, var = _dev.posterior(ctx, spec, L, X, None, _dev.points(ctx, point), want_mean=False)
        return var[0]

    def evaluate(self, index, indexAdded):
        """Ratio for candidate `index` given the already selected `indexAdded` (experimentalDesign.py:249-285)."""
        point = self.mcPoints[index, :].reshape((1, self.space.dimension))
        added = list(indexAdded)
        left = np.setdiff1d(np.setdiff1d(np.arange(self.nMC), added), [index])
        return self._cond_var(added, point) / self._cond_var(list(left), point)


class ExperimentalDesign(object):
    """Base class of the continuous design optimisers."""
    nMCpoints = 10000

    def __init__(self, costFunction, nPoints, nDims, **kwargs):
        self.costFunction = costFunction
        self.nPoints = nPoints
        self.nDims = nDims
        super(ExperimentalDesign, self).__init__()

    def boundsFunction(self, optPoints):
        """+1 if every point has non-zero probability density, else -1 (experimentalDesign.py:312-343)."""
        if len(np.shape(optPoints)) == 1:
            optPoints = np.reshape(optPoints, (int(len(optPoints) / self.nDims), self.nDims))
        out = np.array(self.costFunction.space.probDensity(optPoints), dtype=float)
        out[out == 0.0] = -1e0
        return -1e0 if np.min(out) < 0.0 else 1e0


class ExperimentalDesignDerivative(ExperimentalDesign):
    """Gradient-based continuous design (SciPy SLSQP; the nlopt branch of the reference is not provided)."""

    def __init__(self, costFunction, nPoints, nDims):
        self.addObj = lambda x: 0
        self.addGrad = lambda x: 0
        super(ExperimentalDesignDerivative, self).__init__(costFunction, nPoints, nDims)

    def addPenaltyToObjective(self, addObj, addGrad):
        self.addObj = addObj
        self.addGrad = addGrad

    def objFunc(self, in1, gradIn):
        in0 = np.reshape(in1, (int(len(in1) / self.nDims), self.nDims))
        if gradIn.size > 0:
            gradIn[:] = self.costFunction.derivative(in0)
        out = self.costFunction.evaluate(in0) - 10.0 * np.min(np.array([self.boundsFunction(in0), 0.0]))
        sys.stdout.write("\r Optimization (Cost = %s, ||g||= %s) OK\n" % (str(out), str(None)))
        sys.stdout.flush()
        return out

    def constraint(self, in0, gradIn):
        if gradIn.size > 0:
            gradIn[:] = -optimize.approx_fprime(in0, self.boundsFunction, 1e-8)
        return -self.boundsFunction(in0)

    def beginWithVarGreedy(self, nodesKeep=None, lbounds=[], rbounds=[]):
        """Start SLSQP from a greedy maximum-variance design; `nodesKeep` are pinned in front (:379-409)."""
        kTemp = copy.copy(self.costFunction.gaussianProcess.kernel)
        if nodesKeep is not None:
            mcPoints = np.concatenate((nodesKeep, self.costFunction.mcPoints), axis=0)
            indKeep = np.arange(len(nodesKeep)).tolist()
        else:
            try:
                mcPoints = self.costFunction.mcPoints[:]
            except AttributeError:
                mcPoints = self.costFunction.space.sample((1000, self.costFunction.space.dimension))
            indKeep = []
        startVals = performGreedyVarExperimentalDesign(kTemp, mcPoints, self.nPoints, self.nDims,
                                                       indKeepStart=indKeep)
        return self.begin([startVals], lbounds, rbounds)

    def begin(self, startValues, lbounds=[], rbounds=[]):
        """Minimise the cost from every start with SLSQP (acc=1e-6) and return the best design (:463-497)."""

        def func(xIn, *args):
            in0 = np.reshape(xIn, (int(len(xIn) / self.nDims), self.nDims))
            return self.costFunction.evaluate(in0) - 10.0 * np.min(np.array([self.boundsFunction(in0), 0.0]))

        def grad(xIn, *args):
            in0 = np.reshape(xIn, (int(len(xIn) / self.nDims), self.nDims))
            return self.costFunction.derivative(in0)

        nvar = len(startValues[0]) * self.nDims
        if len(lbounds) == 0:
            bounds = list(zip(-100.0 * np.ones(nvar), 100.0 * np.ones(nvar)))
        else:
            bounds = list(zip(lbounds, rbounds))
        # `freeVariablesOnly` (opt-in; the reference hands SLSQP every coordinate, the pinned ones with equal bounds,
        # experimentalDesign.py:719-724): the leading points whose bounds coincide are taken out of the optimiser's variable
        # vector -- the same problem, but SLSQP's dense algebra is O(variables^3) per iteration and a batch of 64 new points on
        # top of 4000 pinned ones is 64 d variables, not 4064 d.  Cost and gradient still see the whole design.
        free0 = 0
        if getattr(self, "freeVariablesOnly", False) and len(lbounds) > 0:
            fixed = np.asarray(lbounds, dtype=float) == np.asarray(rbounds, dtype=float)
            k = int(np.argmin(fixed)) if not fixed.all() else nvar
            free0 = (k // self.nDims) * self.nDims
        sol = []
        obj = np.zeros((len(startValues)))
        for ii in range(len(startValues)):
            x0 = startValues[ii].reshape((nvar))
            if free0 > 0:
                head = np.asarray(lbounds, dtype=float)[:free0]
                tail = slsqp(lambda xf, *a: func(np.concatenate((head, xf))), x0[free0:],
                             fprime=lambda xf, *a: grad(np.concatenate((head, xf)))[free0:], bounds=bounds[free0:], acc=1e-6)
                pts = np.concatenate((head, np.atleast_1d(tail)))
            else:
                pts = slsqp(func, x0, fprime=grad, bounds=bounds, acc=1e-6)
            sol.append(pts)
            obj[ii] = func(pts)
        best = sol[int(np.argmin(obj))]
        return np.reshape(best, (int(len(best) / self.nDims), self.nDims))


class ExperimentalDesignGreedyWithDerivatives(ExperimentalDesignDerivative):
    """Batch-greedy continuous design (experimentalDesign.py:694-751, the branch without continuation): points are added
    `nPointsBatch` at a time; every batch is an SLSQP run over ALL points so far in which the earlier ones are pinned by
    equal lower/upper bounds (:719-724) and the new ones start from a greedy maximum-variance pick.

    SURVEY.md 8 f2: because the pinned points lead every trial design, each cost/gradient evaluation of a batch re-uses
    the factor of their covariance block (gpx_refit_rows, O(N^2 b) instead of O(N^3/3)) -- GP._factor does that
    transparently for the GP copy the cost function owns."""

    def __init__(self, costFunction, nPoints, nPointsBatch, nDims, **kwargs):
        super(ExperimentalDesignGreedyWithDerivatives, self).__init__(costFunction, nPoints, nDims)
        self.nPointsBatch = nPointsBatch
        if kwargs.get('useCont', 0) != 0:
            raise NotImplementedError("hyper-parameter continuation (experimentalDesign.py:500-692) is outside the GPU "
                                      "hot path")

    def begin(self, startValues=np.array([])):
        nPointsAdded = len(startValues)
        points = startValues.copy() if len(startValues) > 0 else np.zeros((0, self.nDims))
        tol = 1e-16
        err = 10000
        cf = self.costFunction
        while (nPointsAdded < self.nPoints) and (err > tol):
            print("Number of points so far ", nPointsAdded)
            nPointsAdded = nPointsAdded + self.nPointsBatch
            lbounds = -100.0 * np.ones((nPointsAdded * self.nDims))
            rbounds = 100.0 * np.ones((nPointsAdded * self.nDims))
            nPointsPrev = nPointsAdded - self.nPointsBatch
            lbounds[0:nPointsPrev * self.nDims] = points.reshape((nPointsPrev * self.nDims))
            rbounds[0:nPointsPrev * self.nDims] = points.reshape((nPointsPrev * self.nDims))
            currCost = costFunctionGP_IVAR(cf.gaussianProcess, nPointsAdded, cf.space, cf.version, mcPoints=cf.mcPoints)
            currCost.pinnedPoints = nPointsPrev          # (large designs: cost and gradient touch the new batch's rows only)
            expCurr = ExperimentalDesignDerivative(currCost, nPointsAdded, self.nDims)
            expCurr.freeVariablesOnly = getattr(self, "freeVariablesOnly", False)
            points = expCurr.beginWithVarGreedy(nodesKeep=points, lbounds=lbounds, rbounds=rbounds)
            err = currCost.evaluate(points)
            print("Current Error ", err)
            currCost.reset()                             # the batch is over: its forward solve (N x M doubles) leaves the device
        return points


def performGreedyMIExperimentalDesign(costFuncMI, nPoints, start=0):
    """Greedy mutual-information design among costFuncMI.mcPoints, seeded with [start] (:753-785).

    The reference evaluates two fresh pinv's per remaining candidate per step; gpx_mi_greedy carries the
    numerator as an incremental Cholesky row and the denominator as diag((K_SS+noise I)^-1) with rank-one
    down-dates of the inverse, so a step costs O(M^2)."""
    gp = costFuncMI.gaussianProcess
    sess = _dist.session()
    if sess is not None and sess.use_fit(len(costFuncMI.mcPoints)):   # scoring sharded by rows of the inverse (dist_mi_greedy)
        idx, _ = sess.mi_greedy(gp.kernel._spec(), costFuncMI.mcPoints, float(gp.noise), int(nPoints), int(start))
        return costFuncMI.mcPoints[list(idx), :]
    ctx = _dev.context()
    C = _dev.points(ctx, costFuncMI.mcPoints)
    idx, _ = _dev.mi_greedy(ctx, gp.kernel._spec(), C, float(gp.noise), int(nPoints), int(start))
    return costFuncMI.mcPoints[list(idx), :]


def performGreedyVarExperimentalDesign(kernel, mcPoints, nPoints, dimension, weights=None, indKeepStart=[]):
    """Greedy maximum-posterior-variance ("entropy") design among mcPoints (:787-845).

    As in the reference the caller's `indKeepStart` list is extended in place -- that aliasing is how callers
    get the indices back (:808) -- and mcPoints[indices] is returned.  First pick from an empty start is the
    arg-max of the (weighted) prior variance; ties go to the lowest index (np.argmax)."""
    if indKeepStart == []:
        indKeep = []
    else:
        indKeep = indKeepStart
    have = len(indKeep)
    if have % 10 == 0 and have < nPoints:
        print("Number of points we have ", have)
    if have < nPoints:
        ctx = _dev.context()
        C = _dev.points(ctx, np.asarray(mcPoints, dtype=float))
        idx = _dev.greedy_var(ctx, kernel._spec(), C, int(nPoints), keep=indKeep, weights=weights)
        for j in idx[have:]:
            indKeep.append(int(j))
    return mcPoints[indKeep, :]


def greedyIVARStep(gaussianProcess, candidates, mcPoints):
    """One step of discrete greedy IVAR for a GP whose training set is already factored
    (gp.addNodesAndComputeCovariance / train): returns (best_index, costs) where
    costs[j] = costFunctionGP_IVAR(gp, n+1, space, mcPoints=mcPoints).evaluate(vstack(gp.pts, candidates[j]))."""
    gp = gaussianProcess
    if not gp._has_factor():
        raise NotImplementedError("greedyIVARStep needs the dense Cholesky factor (GP built with FITC=... has none)")
    sess = _dist.session()
    if sess is not None and sess.use_eval(len(candidates)):      # candidates sharded, first-minimum merge, costs gathered
        return sess.greedy_ivar_step(gp.kernel._spec(), gp._L, gp._X, candidates, mcPoints, float(gp.noise))
    ctx = _dev.context()
    return _dev.greedy_ivar_step(ctx, gp.kernel._spec(), gp._L, gp._X, _dev.points(ctx, candidates),
                                 _dev.points(ctx, mcPoints), float(gp.noise))


def performGreedyIVARExperimentalDesign(gaussianProcess, candidates, mcPoints, nPoints, returnCosts=False):
    """`nPoints` picks of discrete greedy IVAR for a GP whose training set is already factored: at every pick the candidate whose
    addition minimises costFunctionGP_IVAR.evaluate (experimentalDesign.py:79-117) is appended to the design -- the loop
    `for k: best, _ = greedyIVARStep(gp, candidates, mcPoints); gp.addNodesAndComputeCovariance(vstack(gp.pts, candidates[best]))`
    (the reference has no such function) -- WITHOUT refitting: gpx_greedy_ivar keeps L^-1 K(X, C) and
    cov(Z, C | design) resident and conditions them on each pick by a rank-one update.  The GP itself is not modified.
    Returns candidates[indices] (as the reference's greedy designs do) or (indices, costs) with returnCosts=True."""
    gp = gaussianProcess
    if not gp._has_factor():
        raise NotImplementedError("performGreedyIVARExperimentalDesign needs the dense Cholesky factor (GP built with FITC=... has none)")
    candidates = np.asarray(candidates, dtype=float)
    sess = _dist.session()
    if sess is not None and sess.use_eval(len(candidates)):
        idx, costs = sess.greedy_ivar(gp.kernel._spec(), gp._L, gp._X, candidates, mcPoints, float(gp.noise), int(nPoints))
    else:
        ctx = _dev.context()
        n, m, nmc = gp.pts.shape[0], len(candidates), len(mcPoints)
        resident = 8.0 * (m + 128) * (n + nmc + int(nPoints) + 512)       # W_C, cov(Z, C), the picks' rows (+ padding)
        if not hasattr(ctx, "_hbm_bytes"):
            ctx._hbm_bytes = ctx.info()["hbm_bytes"]
        if resident + 16.0 * (n + 128) * (nmc + 128) + 8.0 * n * n > 0.8 * ctx._hbm_bytes:
            # the resident state does not fit beside the factor: the refit loop it replaces (chunked over the candidates)
            idx, costs, X = [], [], np.array(gp.pts, dtype=float, copy=True)
            g2 = copy.copy(gp)
            for _ in range(int(nPoints)):
                g2.addNodesAndComputeCovariance(X)
                best, c = greedyIVARStep(g2, candidates, mcPoints)
                idx.append(int(best))
                costs.append(float(c[best]))
                X = np.vstack((X, candidates[best:best + 1]))
            idx, costs = np.array(idx, dtype=np.int64), np.array(costs)
        else:
            idx, costs = _dev.greedy_ivar(ctx, gp.kernel._spec(), gp._L, gp._X, _dev.points(ctx, candidates),
                                          _dev.points(ctx, np.asarray(mcPoints, dtype=float)), float(gp.noise), int(nPoints))
    if returnCosts:
        return idx, costs
    return candidates[list(idx), :]
