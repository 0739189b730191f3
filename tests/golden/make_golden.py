#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the REFERENCE (goroda/GPEXP).

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 tests/golden/make_golden.py

Writes ``tests/golden/gpexp_golden.npz`` (inputs + expected outputs, data only) and
``tests/golden/gpexp_golden.json`` (index of cases with hyper-parameters).  Every
case records the literal inputs so that the build's oracle and the HIP path can be
run on identical data without the reference present.

Reference entry points exercised (file:line relative to /root/reference):
  gpExp/kernels.py:49-65,100-123,72-91,183-228,250-293   Kernel.evaluate / evaluateF
  gpExp/gp_kernel_utilities.py:34-68                      calculateCovarianceMatrix
  gpExp/gp.py:76-101,103-154,156-181,213-259,373-440      GP.train/evaluate/addNodes/evaluateVariance/loglike
  gpExp/experimentalDesign.py:60-117,223-285,753-845      IVAR, MI, greedy-variance, greedy-MI
  gpExp/gp.py:182-210,401-426; gp_kernel_utilities.py:70-194   FITC sparse GP, Nystrom eigen-basis
"""
import json
import os
import sys

import numpy as np

REF = os.environ.get("GPEXP_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REF, "gpExp")):
    print("reference not present at %s - nothing to do" % REF)
    sys.exit(0)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from gpExp.kernels import (KernelSquaredExponential, KernelIsoMatern,  # noqa: E402
                           KernelMehlerND, KernelMehler1D)
from gpExp.gp import GP  # noqa: E402
from gpExp.gp_kernel_utilities import calculateCovarianceMatrix  # noqa: E402
from gpExp.approximation import Space  # noqa: E402
from gpExp import experimentalDesign as ED  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
arrays = {}
index = {}


def put(case, name, val):
    arrays["%s/%s" % (case, name)] = np.asarray(val)


def make_kernel(spec):
    kind = spec["kind"]
    if kind == "se":
        return KernelSquaredExponential(list(spec["cl"]), spec["signalSize"], spec["d"])
    if kind == "matern32":
        return KernelIsoMatern(spec["rho"], spec["signalSize"], spec["d"])
    if kind == "mehler":
        return KernelMehlerND(list(spec["t"]), spec["d"])
    if kind == "mehler1d":
        return KernelMehler1D(spec["t"][0], 1)
    raise ValueError(kind)


def gp_case(case, spec, X, y, Z, noise, cov_m=8):
    """K, loglike, coeff, posterior mean / var / small covariance, evaluateVariance."""
    k = make_kernel(spec)
    index[case] = dict(type="gp", kernel=spec,
                       noise=(noise if isinstance(noise, float) else "array"))
    put(case, "X", X)
    put(case, "y", y)
    put(case, "Z", Z)
    if not isinstance(noise, float):
        put(case, "noise", noise)
    put(case, "K", calculateCovarianceMatrix(k, X, noise))
    g = GP(k, noise)
    put(case, "loglike", g.computeLogLike(X, y))
    g.train(X, y)
    put(case, "coeff", g.coeff)
    m, v = g.evaluate(Z, compvar=1)
    put(case, "mean", m)
    put(case, "absvar", v)
    put(case, "var", g.evaluateVariance(Z))
    m2, c = g.evaluate(Z[:cov_m], compvar=2)
    put(case, "cov", c)
    put(case, "precision", g.precisionMatrix)
    put(case, "condK", np.linalg.cond(g.covarianceMatrix))


X5 = np.array([[.1, .2], [-.3, .4], [.5, -.6], [.7, .8], [-.9, -.1]])
y5 = np.array([1, -.5, .25, 0, 2.0])

# --- KAT1: demo.py inputs (demo.py:52-72) ------------------------------------------------
x1 = np.array([-0.8, 0.2, 0.3, -0.1]).reshape((4, 1))
gp_case("kat1_demo", dict(kind="se", cl=[0.3], signalSize=1.0, d=1), x1,
        np.sin(2.0 * np.pi * x1)[:, 0], np.array([[-0.5], [0.0], [0.75]]), 0.0, cov_m=3)

# --- KAT2: Matern 3/2 -------------------------------------------------------------------
rng = np.random.default_rng(102)
gp_case("kat2_matern32", dict(kind="matern32", rho=0.7, signalSize=1.5, d=2), X5, y5,
        rng.uniform(-1, 1, (7, 2)), 1e-2, cov_m=4)

# --- KAT3: Mehler ND (nugget 0.0 in K; GP with small noise) -------------------------------
rng = np.random.default_rng(103)
gp_case("kat3_mehler", dict(kind="mehler", t=[0.5, 0.3], d=2), X5, y5,
        rng.uniform(-1, 1, (6, 2)), 1e-3, cov_m=4)
put("kat3_mehler", "K_nugget0", calculateCovarianceMatrix(make_kernel(index["kat3_mehler"]["kernel"]), X5, 0.0))

# --- mid-size random GP cases (well conditioned: SURVEY 7 "pinv != Cholesky") ------------
def rand_gp(case, spec, n, m, noise, seed, per_point_noise=False):
    rng = np.random.default_rng(seed)
    d = spec["d"]
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(noise) * rng.standard_normal(n)
    Z = rng.uniform(-1, 1, (m, d))
    nz = noise
    if per_point_noise:
        nz = noise * (1.0 + rng.uniform(0, 1, n))
    gp_case(case, spec, X, y, Z, nz)


rand_gp("se_iso_d3_n96", dict(kind="se", cl=[0.2], signalSize=1.0, d=3), 96, 40, 0.05, 201)
rand_gp("se_ard_d8_n130", dict(kind="se", cl=[0.4 + 0.05 * k for k in range(8)], signalSize=1.0, d=8),
        130, 33, 0.1, 202)
rand_gp("matern32_d8_n200", dict(kind="matern32", rho=0.5, signalSize=1.0, d=8), 200, 50, 0.1, 203)
rand_gp("mehler_d3_n64", dict(kind="mehler", t=[0.5, 0.3, 0.7], d=3), 64, 20, 0.05, 204)
rand_gp("se_ard_d2_n77_ppnoise", dict(kind="se", cl=[0.4, 0.9], signalSize=2.0, d=2), 77, 25, 0.02, 205,
        per_point_noise=True)
rand_gp("se_iso_d3_n300", dict(kind="se", cl=[0.2], signalSize=1.0, d=3), 300, 64, 0.05, 206)

# --- Kernel.evaluate shape semantics (kernels.py:49-65): (n,d) vs (1,d), (n,d) vs (n,d) ----
rng = np.random.default_rng(301)
A = rng.uniform(-1, 1, (9, 2))
B = rng.uniform(-1, 1, (9, 2))
for nm, spec in [("se", dict(kind="se", cl=[0.4, 0.9], signalSize=2.0, d=2)),
                 ("matern32", dict(kind="matern32", rho=0.7, signalSize=1.5, d=2)),
                 ("mehler", dict(kind="mehler", t=[0.5, 0.3], d=2))]:
    case = "evaluate_" + nm
    k = make_kernel(spec)
    index[case] = dict(type="evaluate", kernel=spec)
    put(case, "A", A)
    put(case, "B", B)
    put(case, "paired", k.evaluate(A, B))
    put(case, "n_vs_1", k.evaluate(A, B[:1]))
    put(case, "1_vs_n", k.evaluate(A[:1], B))

# --- KAT4: IVAR on a grid ---------------------------------------------------------------
spec4 = dict(kind="se", cl=[0.4, 0.9], signalSize=2.0, d=2)
ka = make_kernel(spec4)
g1 = np.linspace(-1, 1, 9)
grid = np.array([[a, b] for a in g1 for b in g1])
space = Space(2, lambda size: np.random.rand(size[0], size[1]) * 2 - 1, lambda p: 0.25 * np.ones(len(p)))
cf = ED.costFunctionGP_IVAR(GP(ka, 1e-3), 5, space, mcPoints=grid)
index["kat4_ivar"] = dict(type="ivar", kernel=spec4, noise=1e-3)
put("kat4_ivar", "X", X5)
put("kat4_ivar", "mc", grid)
put("kat4_ivar", "ivar", cf.evaluate(X5))

# --- KAT5: greedy variance + composed greedy IVAR (generic-position inputs) -------------
rng = np.random.default_rng(777)
C = rng.uniform(-1, 1, (64, 2))
Zm = rng.uniform(-1, 1, (128, 2))
X0 = rng.uniform(-1, 1, (5, 2))
index["kat5_greedy"] = dict(type="greedy", kernel=spec4, noise=1e-3)
put("kat5_greedy", "C", C)
put("kat5_greedy", "Z", Zm)
put("kat5_greedy", "X0", X0)
keep = [0]
pts = ED.performGreedyVarExperimentalDesign(ka, C, 8, 2, indKeepStart=keep)
put("kat5_greedy", "gvar_idx", np.array(keep, dtype=np.int64))
put("kat5_greedy", "gvar_pts", pts)
# from an empty start (first pick = argmax of prior variance, experimentalDesign.py:816-821)
keep0 = [7]
ED.performGreedyVarExperimentalDesign(ka, C, 10, 2, indKeepStart=keep0)
put("kat5_greedy", "gvar_idx_from7", np.array(keep0, dtype=np.int64))
# weighted variant (experimentalDesign.py:819-820,839-840)
w = 0.5 + rng.uniform(0, 1, 64)
keepw = [3, 11]
ED.performGreedyVarExperimentalDesign(ka, C, 9, 2, weights=w, indKeepStart=keepw)
put("kat5_greedy", "weights", w)
put("kat5_greedy", "gvar_idx_weighted", np.array(keepw, dtype=np.int64))
# record the per-step variance vector of the first run for gap diagnostics
def gvar_trace(kernel, Cc, keep_idx):
    covMat = calculateCovarianceMatrix(kernel, Cc[keep_idx, :])
    inv = np.linalg.pinv(covMat)
    kv = np.zeros((len(keep_idx), len(Cc)))
    for ii, ix in enumerate(keep_idx):
        kv[ii, :] = kernel.evaluate(Cc, Cc[ix:ix + 1])
    out = np.zeros(len(Cc))
    for ii in range(len(Cc)):
        out[ii] = kernel.evaluate(Cc[ii:ii + 1], Cc[ii:ii + 1])[0] - kv[:, ii] @ (inv @ kv[:, ii])
    return out
put("kat5_greedy", "gvar_var_after4", gvar_trace(ka, C, list(np.array(keep[:4]))))

# greedy IVAR composed from costFunctionGP_IVAR (SURVEY 8c note)
Xc = X0.copy()
sel = []
costs = []
allcosts = []
for step in range(4):
    cfk = ED.costFunctionGP_IVAR(GP(ka, 1e-3), len(Xc) + 1, space, mcPoints=Zm)
    vals = np.array([cfk.evaluate(np.vstack((Xc, C[j:j + 1]))) for j in range(len(C))])
    j = int(np.argmin(vals))
    sel.append(j)
    costs.append(vals[j])
    allcosts.append(vals)
    Xc = np.vstack((Xc, C[j:j + 1]))
put("kat5_greedy", "givar_idx", np.array(sel, dtype=np.int64))
put("kat5_greedy", "givar_cost", np.array(costs))
put("kat5_greedy", "givar_allcosts", np.array(allcosts))

# --- KAT6: MI greedy ----------------------------------------------------------------------
rng = np.random.default_rng(778)
Cm = rng.uniform(-1, 1, (40, 2))
gmi = GP(ka, 1e-3)
cm = ED.costFunctionGP_MI(gmi, 6, space, nmc=40, mcpoints=Cm)
ptsmi = ED.performGreedyMIExperimentalDesign(cm, 6)
idx = [int(np.where((Cm == p).all(1))[0][0]) for p in ptsmi]
index["kat6_mi"] = dict(type="mi", kernel=spec4, noise=1e-3)
put("kat6_mi", "C", Cm)
put("kat6_mi", "mi_idx", np.array(idx, dtype=np.int64))
put("kat6_mi", "mi_pts", ptsmi)
# single evaluations (experimentalDesign.py:249-285)
put("kat6_mi", "eval_5_given_0_14", cm.evaluate(5, [0, 14]))
put("kat6_mi", "eval_all_given_0", np.array([cm.evaluate(j, [0]) for j in range(1, 40)]))
# start != 0
ptsmi2 = ED.performGreedyMIExperimentalDesign(cm, 5, start=9)
put("kat6_mi", "mi_idx_start9", np.array([int(np.where((Cm == p).all(1))[0][0]) for p in ptsmi2], dtype=np.int64))

# --- finite-difference reference for the (unrunnable) log-like gradient ---
# loglikeParams(returnDeriv=0) IS runnable; central differences of it pin the analytic gradient
rng = np.random.default_rng(401)
Xg = rng.uniform(-1, 1, (40, 3))
yg = np.sin(2 * np.pi * Xg.sum(1) / 3) + 0.1 * rng.standard_normal(40)
specg = dict(kind="se", cl=[0.5, 0.7, 0.9], signalSize=1.3, d=3)
noise_g = 0.05
index["lml_fd"] = dict(type="lml_fd", kernel=specg, noise=noise_g)
put("lml_fd", "X", Xg)
put("lml_fd", "y", yg)
gg = GP(make_kernel(specg), noise_g)
put("lml_fd", "loglike", gg.loglikeParams(Xg, yg))
keys = list(gg.kernel.hyperParam.keys()) + ["noise"]
fd = []
for key in keys:
    base = dict(gg.kernel.hyperParam)
    base["noise"] = noise_g
    h = 1e-6 * max(1.0, abs(base[key]))
    vals = []
    for sgn in (+1, -1):
        p = dict(base)
        p[key] = base[key] + sgn * h
        g2 = GP(make_kernel(specg), noise_g)
        g2.updateKernelParams(p)
        vals.append(g2.loglikeParams(Xg, yg))
    fd.append((vals[0] - vals[1]) / (2 * h))
index["lml_fd"]["keys"] = keys
put("lml_fd", "fd_grad_raw", np.array(fd))  # d loglike / d theta_k (noise entry: d/d noise, unscaled)

np.savez_compressed(os.path.join(OUT, "gpexp_golden.npz"), **arrays)
with open(os.path.join(OUT, "gpexp_golden.json"), "w") as f:
    json.dump(index, f, indent=1, sort_keys=True)
print("wrote %d arrays, %d cases" % (len(arrays), len(index)))
for c in ("kat1_demo", "kat2_matern32", "kat4_ivar", "kat5_greedy", "kat6_mi"):
    ks = [k for k in arrays if k.startswith(c + "/") and arrays[k].size <= 8 and "idx" in k or k.endswith("loglike") and k.startswith(c)]
    for k in ks:
        print(k, arrays[k])

# --- appended: demo.py flow (config C1) and variance-derivative vectors -----------------------------------
# Same sequence of API calls as the reference's demo.py:52-151 (1-D SE GP, log-like, hyper-parameter fit,
# train/evaluate, IVAR design from a greedy-variance start + SLSQP), with the MC points passed explicitly so
# the fixture does not depend on the global NumPy RNG stream.
def demo_flow():
    import io
    import contextlib
    case = "demo_flow"
    np.random.seed(0)
    kernel = KernelSquaredExponential([0.3], 1.0, 1)
    gpT = GP(kernel, 0.0)
    xTrain = np.array([-0.8, 0.2, 0.3, -0.1]).reshape((4, 1))
    yTrain = np.sin(2.0 * np.pi * xTrain)[:, 0]
    put(case, "xTrain", xTrain)
    put(case, "yTrain", yTrain)
    put(case, "loglike0", gpT.computeLogLike(xTrain, yTrain))
    params, optval = gpT.findOptParamsLogLike(xTrain, yTrain)
    put(case, "opt_keys_order", np.array([0]))
    put(case, "opt_cl0", params["cl0"])
    put(case, "opt_signalSize", params["signalSize"])
    put(case, "opt_noise", params["noise"])
    put(case, "opt_value", optval)
    gpT.train(xTrain, yTrain)
    xDemo = np.linspace(-1, 1, 1000).reshape((1000, 1))
    m, var = gpT.evaluate(xDemo, compvar=1)
    put(case, "mean1", m)
    put(case, "var1", var)
    mc = np.random.rand(10000, 1) * 2.0 - 1.0
    put(case, "mc", mc)
    sampler = lambda size: np.random.rand(size[0], size[1]) * 2.0 - 1.0
    distrib = lambda points: (np.abs(points) < 1.0) * 0.5
    space1 = Space(1, sampler, distrib)
    cf = ED.costFunctionGP_IVAR(gpT, 8, space1, mcPoints=mc)
    exp = ED.ExperimentalDesignDerivative(cf, 8, 1)
    lb = np.concatenate((xTrain.flatten(), -np.ones(4)))
    ub = np.concatenate((xTrain.flatten(), np.ones(4)))
    with contextlib.redirect_stdout(io.StringIO()):
        newPts = exp.beginWithVarGreedy(nodesKeep=xTrain, lbounds=lb, rbounds=ub)
    put(case, "design", newPts)
    put(case, "design_cost", cf.evaluate(newPts))
    # greedy start that SLSQP began from
    keep = [0, 1, 2, 3]
    with contextlib.redirect_stdout(io.StringIO()):
        start = ED.performGreedyVarExperimentalDesign(copy_kernel(gpT.kernel), np.concatenate((xTrain, mc), axis=0),
                                                      8, 1, indKeepStart=keep)
    put(case, "greedy_start_idx", np.array(keep, dtype=np.int64))
    put(case, "greedy_start_cost", cf.evaluate(start))
    put(case, "greedy_start_grad", cf.derivative(start))
    index[case] = dict(type="demo")


def copy_kernel(k):
    import copy
    return copy.copy(k)


def varderiv_case():
    case = "varderiv"
    rng = np.random.default_rng(501)
    X = rng.uniform(-1, 1, (7, 2))
    Z = rng.uniform(-1, 1, (9, 2))
    spec = dict(kind="se", cl=[0.4, 0.6], signalSize=1.3, d=2)
    g = GP(make_kernel(spec), 1e-2)
    g.addNodesAndComputeCovariance(X)
    index[case] = dict(type="varderiv", kernel=spec, noise=1e-2)
    put(case, "X", X)
    put(case, "Z", Z)
    put(case, "dvar_dpts", g.evaluateVarianceDerivative(Z))
    put(case, "dvar_dnew", g.evaluateVarianceDerivWRTnewpt(Z))
    put(case, "kernel_derivative", g.kernel.derivative(X, Z[:1]))


def fitc_case():
    """FITC sparse GP (gp.py:182-210, 401-426; gp_kernel_utilities.py:70-104) and the Nystrom eigen-basis
    (gp_kernel_utilities.py:107-194).  A fresh GP per operation: the reference compares `self.fitcnodes == None`, which
    raises for an ndarray, so only the first FITC operation of an instance runs (numpy 2.x)."""
    from gpExp.gp_kernel_utilities import calculateCovarianceMatrixFITC, calculateKernelBasisFunctionsMC
    case = "fitc"
    rng = np.random.default_rng(777001)
    n, d, m = 60, 2, 11
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(2.0 * X[:, 0]) + 0.5 * X[:, 1] + 0.05 * rng.standard_normal(n)
    Z = rng.uniform(-1, 1, (m, d))
    spec = dict(kind="se", cl=[0.5, 0.8], signalSize=1.2, d=2)
    noise, frac, seed = 0.05, 0.25, 3
    index[case] = dict(type="fitc", kernel=spec, noise=noise, fitc=frac, seed=seed)
    put(case, "X", X); put(case, "y", y); put(case, "Z", Z)
    np.random.seed(seed)
    g = GP(make_kernel(spec), noise, FITC=frac)
    g.train(X, y)
    put(case, "fitcnodes", g.fitcnodes)
    put(case, "cov", g.covarianceMatrix)
    put(case, "prec", g.precisionMatrix)
    put(case, "coeff", g.coeff)
    mean, var = g.evaluate(Z, compvar=1)
    put(case, "mean", mean); put(case, "var", var)
    put(case, "var_signed", g.evaluateVariance(Z, parallel=0))
    np.random.seed(seed)
    g2 = GP(make_kernel(spec), noise, FITC=frac)
    put(case, "loglike", g2.computeLogLike(X, y))
    assert np.array_equal(g2.fitcnodes, g.fitcnodes)
    cov, prec, sn = calculateCovarianceMatrixFITC(make_kernel(spec), X, noise, g.fitcnodes.copy(), returnCov=True)
    put(case, "util_cov", cov); put(case, "util_prec", prec)
    mc = rng.uniform(-1, 1, (80, d))
    put(case, "nys_mc", mc)
    ev, evec = calculateKernelBasisFunctionsMC(make_kernel(spec), 6, mc)
    put(case, "nys_eigv", ev); put(case, "nys_eigve", evec)


def c2_case():
    """BASELINE config C2 at full size (N=4096, d=3, iso-SE l=0.2, s=1, noise=0.05, seed 4096): the one
    config the reference itself can run (~25 s here).  Inputs are regenerated from the seed by the tests; only compact
    outputs are stored."""
    case = "c2_full"
    N, d, M = 4096, 3, 4096
    rng = np.random.default_rng(4096)
    X = rng.uniform(-1, 1, (N, d))
    noise = 0.05
    y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(noise) * rng.standard_normal(N)
    Z = rng.uniform(-1, 1, (M, d))
    spec = dict(kind="se", cl=[0.2], signalSize=1.0, d=3)
    g = GP(make_kernel(spec), noise)
    g.train(X, y)
    mean, var = g.evaluate(Z[:256], compvar=1)
    index[case] = dict(type="c2", kernel=spec, noise=noise, N=N, M=M, seed=4096)
    put(case, "loglike", g.computeLogLike(X, y))
    put(case, "coeff", g.coeff)
    put(case, "mean256", mean)
    put(case, "var256", var)
    put(case, "ytalpha", float(y @ g.coeff))
    put(case, "cond_proxy", np.linalg.cond(g.covarianceMatrix))


def hetero_ivar_case():
    """costFunctionGP_IVAR.evaluate with a heteroscedastic space.noiseFunc (experimentalDesign.py:107-117): the design
    points' own noise enters K as a per-point nugget."""
    case = "ivar_noisefunc"
    rng = np.random.default_rng(909)
    d, n, nmc = 2, 23, 150
    X = rng.uniform(-1, 1, (n, d))
    mc = rng.uniform(-1, 1, (nmc, d))
    spec = dict(kind="se", cl=[0.45, 0.7], signalSize=1.4, d=2)
    nf = lambda p: 0.01 + 0.05 * (p[:, 0] ** 2 + 0.5 * p[:, 1] ** 2)   # noqa: E731
    space = Space(d, lambda size: np.random.rand(size[0], size[1]) * 2 - 1, lambda p: 0.25 * np.ones(len(p)), noise=nf)
    cf = ED.costFunctionGP_IVAR(GP(make_kernel(spec), 1e-3), n, space, mcPoints=mc)
    index[case] = dict(type="ivar_noisefunc", kernel=spec, noise=1e-3, noisefunc="0.01 + 0.05*(x0^2 + 0.5*x1^2)")
    put(case, "X", X); put(case, "mc", mc)
    put(case, "pointnoise", nf(X))
    put(case, "ivar", cf.evaluate(X))


demo_flow()
varderiv_case()
fitc_case()
c2_case()
hetero_ivar_case()
np.savez_compressed(os.path.join(OUT, "gpexp_golden.npz"), **arrays)
with open(os.path.join(OUT, "gpexp_golden.json"), "w") as f:
    json.dump(index, f, indent=1, sort_keys=True)
print("appended demo_flow + varderiv: %d arrays, %d cases" % (len(arrays), len(index)))
print("demo opt:", arrays["demo_flow/opt_cl0"], arrays["demo_flow/opt_signalSize"], arrays["demo_flow/opt_noise"],
      arrays["demo_flow/opt_value"], "design", np.sort(arrays["demo_flow/design"][:, 0]), arrays["demo_flow/design_cost"])
