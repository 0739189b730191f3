"""CPU oracle for the GPEXP GP-inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module restates, in plain NumPy, the *algorithms of the reference* (goroda/GPEXP,
mounted read-only at /root/reference) for the path named by BASELINE.json `north_star`:
kernel-matrix assembly -> pseudo-inverse "fit" -> posterior mean / variance ->
log-marginal likelihood -> IVAR / greedy-variance / MI design costs.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and only as the checker / the timed CPU baseline.  Nothing under `gpexp_amd/` imports it; the
product path fails loudly when the HIP library is missing.

Parity pin: the reference holds no tests and no golden vectors of its own (SURVEY.md 8c:
"parity unpinned" by the reference).  This oracle is pinned instead against outputs of the
reference itself, generated in the build container by `tests/golden/make_golden.py`
(imports /root/reference) and committed as `tests/golden/gpexp_golden.npz`;
`tests/test_oracle_golden.py` checks every function below against those vectors.
Sub-paths with NO runnable reference (follow the source as text): Matern nu=5/2
(kernels.py:85-91 leaves `out` unbound) and `loglikeParams(returnDeriv=1)`
(kernels.py:140-141 indexes with a float) -- those two are "parity unpinned"; the gradient is
cross-checked against central differences of the runnable `loglikeParams(returnDeriv=0)`.

Kernel specs are plain dicts so that fixtures stay data:
    {"kind": "se",       "cl": [l_0..l_{d-1}] or [l], "signalSize": s, "d": d}
    {"kind": "matern32", "rho": r, "signalSize": s, "d": d}
    {"kind": "matern52", "rho": r, "signalSize": s, "d": d}       (unpinned)
    {"kind": "mehler",   "t": [t_0..t_{d-1}], "d": d}
All arithmetic is IEEE fp64.
"""
import numpy as np

__all__ = [
    "kernel_eval", "kernel_diag", "cov_matrix", "cross_matrix", "fit", "posterior",
    "loglike", "loglike_grad", "ivar", "greedy_var", "greedy_ivar", "mi_evaluate", "greedy_mi",
    "kernel_derivative", "variance_deriv_wrt_newpt", "variance_derivative", "ivar_grad",
]


# --------------------------------------------------------------------------------------------
# L0  kernels  (reference gpExp/kernels.py)
# --------------------------------------------------------------------------------------------
def _cl(spec):
    """Length-1 correlationLength is tiled to d (kernels.py:106-107)."""
    cl = np.asarray(spec["cl"], dtype=float)
    if cl.size == 1:
        cl = np.tile(cl, spec["d"])
    return cl


def kernel_eval(spec, x1, x2):
    """Paired evaluation k(x1[i], x2[i]) for equally shaped (n,d) inputs, or (n,d) vs (1,d).

    Follows Kernel.evaluate's tiling rule (kernels.py:49-65) and the evaluateF bodies:
    squared exponential kernels.py:115-123, Matern nu=3/2 kernels.py:81-91,
    Mehler N-D = product of 1-D Mehler kernels kernels.py:200-228 + 264-293.
    """
    x1 = np.asarray(x1, dtype=float)
    x2 = np.asarray(x2, dtype=float)
    assert x1.ndim == 2 and x2.ndim == 2
    d = spec["d"]
    assert x1.shape[1] == d and x2.shape[1] == d
    n1, n2 = x1.shape[0], x2.shape[0]
    if n1 > n2:
        x2 = np.tile(x2, (n1, 1))
    elif n1 < n2:
        x1 = np.tile(x1, (n2, 1))
    assert x1.shape == x2.shape
    kind = spec["kind"]
    if kind == "se":
        cl = _cl(spec)
        w = cl ** -2.0
        return spec["signalSize"] * np.exp(-0.5 * np.sum((x1 - x2) ** 2.0 * w[None, :], axis=1))
    if kind == "matern32":
        r = np.sqrt(np.sum((x1 - x2) ** 2.0, axis=1))
        t = np.sqrt(3) * r / spec["rho"]
        return spec["signalSize"] * (1.0 + t) * np.exp(-t)
    if kind == "matern52":
        # NOT in the reference (kernels.py:85-91 only handles nu=3/2): standard closed form
        r = np.sqrt(np.sum((x1 - x2) ** 2.0, axis=1))
        t = np.sqrt(5) * r / spec["rho"]
        return spec["signalSize"] * (1.0 + t + t * t / 3.0) * np.exp(-t)
    if kind in ("mehler", "mehler1d"):   # KernelMehler1D (kernels.py:264-293) = the d = 1 factor of KernelMehlerND
        out = np.ones(x1.shape[0])
        for k, t in enumerate(spec["t"]):
            a = x1[:, k]
            b = x2[:, k]
            one = (1.0 - t ** 2.0) ** (-1.0 / 2.0) * np.exp(
                -(a ** 2.0 * t ** 2.0 - 2.0 * t * a * b + b ** 2.0 * t ** 2.0) / (2.0 * (1.0 - t ** 2.0)))
            out = out * one
        return out
    raise ValueError("unknown kernel kind %r" % (kind,))


def kernel_diag(spec, z):
    """k(z_j, z_j) for every row (the `kernel.evaluate(newpt, newpt)` of gp.py:140, 251)."""
    return kernel_eval(spec, z, z)


# --------------------------------------------------------------------------------------------
# L1  covariance assembly  (reference gpExp/gp_kernel_utilities.py:34-68)
# --------------------------------------------------------------------------------------------
def _nugget_vec(nugget, n):
    """float -> scalar*I; ndarray (n,) -> per point; anything else is an error in the reference
    (UnboundLocalError at gp_kernel_utilities.py:67); mirrored as TypeError."""
    if isinstance(nugget, float):
        return nugget * np.ones(n)
    if isinstance(nugget, np.ndarray):
        return nugget[:]
    raise TypeError("nugget must be a float or an ndarray (reference: gp_kernel_utilities.py:62-67)")


def cov_matrix(spec, points, nugget=0.0, row_loop=True):
    """K_ij = k(x_i, x_j) + diag(nugget): one kernel row per training point, then + np.diag.

    row_loop=True walks the rows exactly as gp_kernel_utilities.py:56-60 does (that loop IS the
    reference's cost; bench.py's cpu_baseline times this mode); row_loop=False evaluates the same
    arithmetic per element in one broadcast for speed in tests.
    """
    points = np.asarray(points, dtype=float)
    n, d = points.shape
    if row_loop:
        K = np.zeros((n, n))
        for j in range(n):
            K[j, :] = kernel_eval(spec, points, points[j:j + 1, :])
    else:
        a = np.repeat(points, n, axis=0)
        b = np.tile(points, (n, 1))
        K = kernel_eval(spec, b, a).reshape(n, n)
    return K + np.diag(_nugget_vec(nugget, n))


def cross_matrix(spec, newpt, pts):
    """kernelvals[m, j] = k(newpt_m, pts_j), built column by column (gp.py:132-135, 246-249)."""
    newpt = np.asarray(newpt, dtype=float)
    pts = np.asarray(pts, dtype=float)
    out = np.zeros((newpt.shape[0], pts.shape[0]))
    for j in range(pts.shape[0]):
        out[:, j] = kernel_eval(spec, newpt, pts[j:j + 1, :])
    return out


# --------------------------------------------------------------------------------------------
# L2  GP model  (reference gpExp/gp.py)
# --------------------------------------------------------------------------------------------
def fit(spec, X, y, noise):
    """GP.train (gp.py:76-101) -> dict(K, P, coeff): K via cov_matrix with nugget=noise (a variance,
    gp.py:68,178), P = pinv(K) (gp.py:181), coeff = P y (gp.py:101; zero prior mean gp.py:73)."""
    K = cov_matrix(spec, X, noise)
    P = np.linalg.pinv(K)
    out = dict(K=K, P=P, X=np.array(X, dtype=float, copy=True))
    if y is not None:
        y = np.asarray(y, dtype=float)
        assert y.ndim == 1, "evaluations must be an (N,) array for training GP"
        out["coeff"] = P @ y
    return out


def posterior(spec, model, Z, compvar=1):
    """GP.evaluate / GP.evaluateVariance (gp.py:103-154, 213-259).

    Returns (mean, var) with the SIGNED variance of evaluateVariance (gp.py:253-256); GP.evaluate
    (compvar=1) returns abs(var) (gp.py:145).  compvar=2 returns the (M,M) posterior covariance
    (gp.py:146-152).  The variance is a per-point k_z^T (P k_z) loop, as the reference does it.
    """
    kv = cross_matrix(spec, Z, model["X"])
    mean = kv @ model["coeff"] if "coeff" in model else None
    if compvar == 2:
        kzz = np.zeros((len(Z), len(Z)))
        for j in range(len(Z)):
            kzz[:, j] = kernel_eval(spec, Z, Z[j:j + 1, :])
        return mean, kzz - kv @ (model["P"] @ kv.T)
    prior = kernel_diag(spec, Z)
    var = np.zeros(len(Z))
    for j in range(len(Z)):
        var[j] = prior[j] - kv[j, :] @ (model["P"] @ kv[j, :].T)
    return mean, var


def loglike(spec, X, y, noise):
    """GP.computeLogLike -> loglikeParams(returnDeriv=0) (gp.py:373-440):
    -1/2 y^T pinv(K) y - 1/2 slogdet(K)[1] - N/2 log(2 pi)."""
    K = cov_matrix(spec, X, noise)
    P = np.linalg.pinv(K)
    _, logdet = np.linalg.slogdet(K)
    a = P @ y
    return -0.5 * (y @ a) - 0.5 * logdet - len(y) / 2.0 * np.log(2.0 * np.pi)


def hyp_keys(spec):
    """Key order of kernel.hyperParam + ['noise'] (gp.py:442; SE keys kernels.py:108-111, Matern keys kernels.py:74-79)."""
    if spec["kind"] == "se":
        return ["cl%d" % i for i in range(spec["d"])] + ["signalSize", "noise"]
    if spec["kind"] in ("matern32", "matern52"):
        return ["rho", "signalSize", "noise"]
    raise NotImplementedError("no hyper-parameter derivatives for this kernel")


def loglike_grad(spec, X, y, noise):
    """loglikeParams(returnDeriv=1) (gp.py:444-466) with the SE hyper-parameter derivatives of
    kernels.py:125-144:  dK/d signalSize = exp(.),  dK/d cl_k = K * D_k^2 / cl_k^3,  dK/d noise = I,
    out[key] = 1/2 tr((a a^T - P) dK_key); the 'noise' entry is then multiplied by 2*noise
    (gp.py:463-464).  UNPINNED: the reference's own code raises IndexError (kernels.py:140-141);
    tests cross-check this against central differences of the runnable loglike().  Matern kernels (round 6): an extension
    with no reference counterpart, pinned the same way -- by central differences of loglike(), which IS pinned for nu = 3/2.
    Returns (value, {key: derivative}) like the reference.
    """
    X = np.asarray(X, dtype=float)
    n, d = X.shape
    K0 = cov_matrix(spec, X, 0.0, row_loop=False)
    K = K0 + np.diag(_nugget_vec(noise, n))
    P = np.linalg.pinv(K)
    _, logdet = np.linalg.slogdet(K)
    a = P @ y
    val = -0.5 * (y @ a) - 0.5 * logdet - n / 2.0 * np.log(2.0 * np.pi)
    T = np.outer(a, a) - P
    out = {}
    if spec["kind"] == "se":
        cl = _cl(spec)
        for k in range(d):
            D2 = (X[:, k][:, None] - X[:, k][None, :]) ** 2.0
            out["cl%d" % k] = 0.5 * np.trace(T @ (K0 * D2 / cl[k] ** 3.0))
    else:
        # EXTENSION (round 6): the reference's Matern has no derivativeWrtHypParams at all (kernels.py:93-97 raises).  Closed
        # forms with t = sqrt(2 nu) r / rho:  nu = 3/2: dk/d rho = s t^2 e^-t / rho;  nu = 5/2: dk/d rho = s t^2 (1 + t) e^-t / (3 rho).
        assert spec["kind"] in ("matern32", "matern52")
        r = np.sqrt(np.maximum(((X[:, None, :] - X[None, :, :]) ** 2.0).sum(-1), 0.0))
        rho, sg = spec["rho"], spec["signalSize"]
        if spec["kind"] == "matern32":
            t = np.sqrt(3.0) * r / rho
            dK = sg * t * t * np.exp(-t) / rho
        else:
            t = np.sqrt(5.0) * r / rho
            dK = sg * t * t * (1.0 + t) * np.exp(-t) / (3.0 * rho)
        out["rho"] = 0.5 * np.trace(T @ dK)
    out["signalSize"] = 0.5 * np.trace(T @ (K0 / spec["signalSize"]))
    out["noise"] = 0.5 * np.trace(T) * noise * 2.0
    return val, out


# --------------------------------------------------------------------------------------------
# f1  derivatives w.r.t. point locations  (reference gpExp/kernels.py, gpExp/gp.py)
# --------------------------------------------------------------------------------------------
def kernel_derivative(spec, x1, x2):
    """Kernel.derivative(x1, x2): out[j, i] = dK(x1[j], x2) / d x1[j, i], x2 a single (1, d) point.

    Squared exponential (kernels.py:146-181): -signalSize * (x1 - x2) / cl^2 * evaluate(x1, x2) -- the kernel value
    already carries signalSize, so it enters twice (:177), kept as is.  1-D Mehler (kernels.py:295-324):
    -1/2 (2 x1 t^2 - 2 t x2) / (1 - t^2) * evaluate(x1, x2).  Other kernels have no derivative in the reference.
    A 1-D "mehler" spec is the reference's KernelMehler1D."""
    x1 = np.asarray(x1, dtype=float)
    x2 = np.asarray(x2, dtype=float)
    assert x2.ndim == 2 and x1.ndim == 2 and x2.shape == (1, spec["d"]) and x1.shape[1] == spec["d"]
    n = x1.shape[0]
    r = kernel_eval(spec, x1, x2)
    if spec["kind"] == "se":
        cl = _cl(spec)
        return (-spec["signalSize"] * 0.5 * 2 * (x1 - np.tile(x2, (n, 1))) / np.tile(cl ** 2.0, (n, 1))
                * np.tile(np.reshape(r, (n, 1)), (1, spec["d"])))
    if spec["kind"] in ("mehler", "mehler1d") and spec["d"] == 1:
        t = spec["t"][0]
        return -0.5 * (2.0 * x1 * t ** 2.0 - 2.0 * t * np.tile(x2, (n, 1))) / (1.0 - t ** 2.0) * np.reshape(r, (n, 1))
    raise NotImplementedError("the reference defines derivative() for SE and 1-D Mehler only")


def variance_deriv_wrt_newpt(spec, model, Z):
    """GP.evaluateVarianceDerivWRTnewpt (gp.py:261-280): d var(z_i) / d z_i, flattened point-major."""
    X, P = model["X"], model["P"]
    Z = np.asarray(Z, dtype=float)
    derivs = np.zeros((Z.shape[0], Z.shape[1], len(X)))
    evals = np.zeros((Z.shape[0], len(X)))
    for ii in range(len(X)):
        p = X[ii:ii + 1, :]
        derivs[:, :, ii] = kernel_derivative(spec, Z, p)
        evals[:, ii] = kernel_eval(spec, p, Z)
    es = P @ evals.T
    out = np.zeros(Z.shape)
    for ii in range(len(Z)):
        out[ii, :] = -2.0 * derivs[ii, :, :] @ es[:, ii]
    return out.reshape(np.prod(Z.shape))


def variance_derivative(spec, model, Z, noise_func=None):
    """GP.evaluateVarianceDerivative (gp.py:282-341): out[k*d + l, j] = d var(z_j) / d X[k, l] for the model fitted on
    X = model["X"] with precision model["P"]; `noise_func` = callable with .deriv (demo2.py:45-58), its terms at
    gp.py:314-320 (including the whole-set norm test of :318)."""
    X, P = model["X"], model["P"]
    Z = np.asarray(Z, dtype=float)
    n, d = X.shape
    dcov = np.zeros((n, n, d))
    tot = np.zeros((len(Z), n))
    dtot = []
    for zz in range(n):
        p = X[zz:zz + 1, :]
        ind = np.array([np.linalg.norm(pp - p) < 1e-10 for pp in X])
        dcov[zz, :, :] = kernel_derivative(spec, X, p)
        tot[:, zz] = kernel_eval(spec, p, Z)
        dtot.append(-kernel_derivative(spec, Z, p))
        if noise_func is not None:
            dcov[zz, :, :] += np.tile(ind.reshape((n, 1)), d) * noise_func.deriv(X)
            if np.linalg.norm(p - Z) < 1e-10:
                tot[:, zz] += noise_func(p)
                dtot[-1] -= noise_func.deriv(p)
    e = tot @ P
    out1 = np.zeros((n * d, len(Z)))
    out2 = np.zeros((n * d, len(Z)))
    for jj in range(n):
        for kk in range(d):
            out1[jj * d + kk, :] = 2.0 * e[:, jj] * dtot[jj][:, kk]
            dS = np.zeros((n, n))
            dS[jj, :] = dcov[:, jj, kk]
            dS[:, jj] = dcov[:, jj, kk]
            out2[jj * d + kk, :] = -np.sum((e @ dS) * e, axis=1)
    return -(out1 + out2)


def ivar_grad(spec, design, mc, noise, noise_func=None):
    """costFunctionGP_IVAR.derivative, version 1 (experimentalDesign.py:168-179): refit on the design points (with the
    per-point noise noise_func(design) when given), mean over the MC points of variance_derivative."""
    nug = noise if noise_func is None else noise_func(design)
    model = fit(spec, design, None, nug)
    return np.sum(variance_derivative(spec, model, mc, noise_func), axis=1) / float(len(mc))


# --------------------------------------------------------------------------------------------
# L3  experimental-design cost evaluators  (reference gpExp/experimentalDesign.py)
# --------------------------------------------------------------------------------------------
def ivar(spec, design, mc, noise):
    """costFunctionGP_IVAR.evaluate, version 1 (experimentalDesign.py:104-117): refit on the design
    points (no y), posterior variance at every MC point, abs(mean)."""
    model = fit(spec, design, None, noise)
    _, var = posterior(spec, model, mc, compvar=1)
    return np.abs(1.0 / float(len(mc)) * np.sum(var))


def greedy_var(spec, cand, n_points, weights=None, keep_start=()):
    """performGreedyVarExperimentalDesign (experimentalDesign.py:787-845): greedy maximum posterior
    variance among `cand`; nugget 0.0 (:825); pinv (:826); optional weights (:819-820, :839-840);
    np.argmax = first maximum.  Returns the index list (the reference returns cand[idx] and
    appends to the caller's list)."""
    keep = list(keep_start)
    cand = np.asarray(cand, dtype=float)
    while len(keep) < n_points:
        if len(keep) == 0:
            k = kernel_diag(spec, cand)
        else:
            P = np.linalg.pinv(cov_matrix(spec, cand[keep, :], 0.0))
            kv = np.zeros((len(keep), len(cand)))
            for ii, ix in enumerate(keep):
                kv[ii, :] = kernel_eval(spec, cand, cand[ix:ix + 1, :])
            prior = kernel_diag(spec, cand)
            k = np.zeros(len(cand))
            for ii in range(len(cand)):
                k[ii] = prior[ii] - kv[:, ii].T @ (P @ kv[:, ii])
        if weights is not None:
            k = k * weights
        keep.append(int(np.argmax(k)))
    return keep


def greedy_ivar(spec, X0, cand, mc, noise, n_steps):
    """Discrete greedy IVAR as composed in SURVEY.md 8c (the reference has no such function; its
    oracle is costFunctionGP_IVAR.evaluate(vstack(X, cand[j])) for every j, arg-min, append):
    returns (indices, costs, all_costs[n_steps, M])."""
    X = np.array(X0, dtype=float, copy=True)
    idx, costs, allc = [], [], []
    for _ in range(n_steps):
        vals = np.array([ivar(spec, np.vstack((X, cand[j:j + 1])), mc, noise) for j in range(len(cand))])
        j = int(np.argmin(vals))
        idx.append(j)
        costs.append(vals[j])
        allc.append(vals)
        X = np.vstack((X, cand[j:j + 1]))
    return idx, np.array(costs), np.array(allc)


def mi_evaluate(spec, cand, noise, index, added):
    """costFunctionGP_MI.evaluate(index, indexAdded) (experimentalDesign.py:249-285):
    var(c|A) / var(c|Abar), Abar = all \\ A \\ {c}; each conditional via a fresh pinv with nugget=noise."""
    cand = np.asarray(cand, dtype=float)
    pt = cand[index:index + 1, :]
    var = kernel_eval(spec, pt, pt)
    A = cand[list(added), :].reshape((len(added), cand.shape[1]))
    kA = kernel_eval(spec, A, pt)
    num = var - kA.T @ (np.linalg.pinv(cov_matrix(spec, A, noise)) @ kA)
    left = np.setdiff1d(np.setdiff1d(np.arange(len(cand)), added), [index])
    B = cand[left, :]
    kB = kernel_eval(spec, B, pt)
    den = var - kB.T @ (np.linalg.pinv(cov_matrix(spec, B, noise)) @ kB)
    return (num / den)[0]


def greedy_mi(spec, cand, noise, n_points, start=0):
    """performGreedyMIExperimentalDesign (experimentalDesign.py:753-785): seeded with [start]; each step
    evaluates every remaining candidate (ascending index order, np.setdiff1d) and takes np.argmax."""
    keep = [start]
    options = np.setdiff1d(np.arange(len(cand)), keep)
    ratios = []
    while len(keep) < n_points:
        out = np.array([mi_evaluate(spec, cand, noise, int(j), keep) for j in options])
        new = int(options[np.argmax(out)])
        ratios.append(out.max())
        keep.append(new)
        options = np.setdiff1d(options, new)
    return keep, np.array(ratios)


# ---- FITC sparse approximation and Nystrom eigen-basis -------------------------------------------
def fitc_matrices(spec, X, noise, snodes):
    """Covariance Q + G and its Woodbury precision for inducing points `snodes`, as the reference builds them
    (gp.py:189-206 == gp_kernel_utilities.py:81-96): Quu and K both carry the nugget; G = diag(K - Q);
    P = G^-1 - G^-1 Kfu (Quu + Kuf G^-1 Kfu)^-1 Kuf G^-1 with G^-1 = 1 / (g + 1e-12)."""
    X = np.asarray(X, dtype=float)
    snodes = np.asarray(snodes, dtype=float)
    Quu = cov_matrix(spec, snodes, noise, row_loop=False)
    Kuf = cross_matrix(spec, snodes, X)                      # (nu, N): kernelvals[jj, :] = k(snodes_jj, nodes)
    Q = Kuf.T @ (np.linalg.pinv(Quu) @ Kuf)
    K = cov_matrix(spec, X, noise, row_loop=False)
    g = np.diag(K - Q)
    ginv = 1.0 / (g + 1e-12)
    inner = np.linalg.inv(Quu + (Kuf * ginv) @ Kuf.T)
    prec = np.diag(ginv) - (ginv[:, None] * Kuf.T) @ inner @ (Kuf * ginv)
    return Q + np.diag(g), prec


def fitc_fit(spec, X, y, noise, snodes):
    """GP.train with FITC (gp.py:76-101): coeff = P y (zero prior mean)."""
    cov, prec = fitc_matrices(spec, X, noise, snodes)
    return dict(X=np.asarray(X, dtype=float), cov=cov, prec=prec, P=prec, coeff=prec @ np.asarray(y, dtype=float))


def fitc_loglike(spec, X, y, noise, snodes):
    """loglikeParams, FITC branch (gp.py:401-440): slogdet of Q + G, quadratic form with the Woodbury precision."""
    cov, prec = fitc_matrices(spec, X, noise, snodes)
    y = np.asarray(y, dtype=float)
    return -0.5 * y @ (prec @ y) - 0.5 * np.linalg.slogdet(cov)[1] - len(y) / 2.0 * np.log(2.0 * np.pi)


def nystrom_basis(spec, num_basis, mc):
    """calculateKernelBasisFunctionsMC (gp_kernel_utilities.py:147-194): leading eigen-pairs of K(mc, mc), descending,
    eigenvalues / nMC and eigenvectors * sqrt(nMC).  (Dense eigh here; the reference runs ARPACK on a matrix-free
    operator -- same eigen-pairs, eigenvectors defined up to sign.)"""
    mc = np.asarray(mc, dtype=float)
    n = mc.shape[0]
    k = int(min(num_basis, n))
    w, v = np.linalg.eigh(cov_matrix(spec, mc, 0.0, row_loop=False))
    w, v = w[::-1][:k], v[:, ::-1][:, :k]
    return w / float(n), v * np.sqrt(float(n))
