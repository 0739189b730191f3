#!/usr/bin/env python3
"""Round-6 value pins for the FULL-SIZE configurations -- LAPACK, NOT the reference.

    python3 tests/golden/make_golden_r6.py [c4_matern52 c4_matern32 c3 c5_lite]

The reference cannot run these sizes (pinv + slogdet of a 32768-order matrix: hours and ~35 GB of SVD workspace; the
N = 8192 reference run of make_golden_r4.py took tens of minutes), so the values below come from the SAME closed forms the
reference evaluates (gp.py:373-440 log-marginal, gp.py:213-259 variance, experimentalDesign.py:104-117 IVAR, gp.py:444-466
gradient) computed by an independent route in the build container: NumPy assembly from coordinate DIFFERENCES (no Gram
trick), LAPACK dpotrf / dtrtrs / dpotri through SciPy (at N = 32768 block by block: chol_blocked).  Written to ``tests/golden/gpexp_golden_r6.npz`` / ``.json``; every
array is labelled "LAPACK, not reference" in the index.  Inputs are regenerated from the seed by the tests (same generator,
same order of draws as bench.py's workload()).

  c4_matern52 / c4_matern32   N = 32768, d = 8, rho = 0.5, signalSize = 1, noise = 0.1, seed 32768 (bench.py's workload):
        log-marginal likelihood, log det K, y^T alpha, IVAR over the first 4096 of the M = 32768 MC points, the first 256
        individual posterior variances.
  c3    N = 16384, d = 8 ARD squared exponential l_k = 0.4 + 0.05 k, noise = 0.1, seed 16384, 65536 candidates, nMC = 4096:
        IVAR of the start design and the greedy-IVAR step's cost (IVAR after adding the candidate) of 256 fixed candidates.
  c5_full   N = 65536, d = 10 (the full BASELINE config 5): log-marginal, log det, y^T alpha, 256 entries of alpha, 256 variances.
  c5_lite   N = 16384, d = 10 ARD squared exponential l_k = 0.5 + 0.03 k, noise = 0.1, seed 65536: log-marginal and its 12
        derivatives (10 length scales, signalSize, noise) from a dense dpotri inverse.
"""
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

OUT = os.path.dirname(os.path.abspath(__file__))
NPZ = os.path.join(OUT, "gpexp_golden_r6.npz")
JSN = os.path.join(OUT, "gpexp_golden_r6.json")
LABEL = "LAPACK (NumPy difference assembly + dpotrf/dtrtrs/dpotri), not reference"


def c4_inputs(n=32768, d=8, m=32768, seed=32768, noise=0.1):
    """bench.py workload(): the tests regenerate exactly this."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(noise) * rng.standard_normal(n)
    Z = rng.uniform(-1, 1, (m, d))
    return X, y, Z


def c3_inputs(N=16384, d=8, M=65536, nmc=4096, seed=16384):
    """tests/test_gpu_scale.py::test_c3_*: same generator, same order of draws."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (N, d))
    C, Z = rng.uniform(-1, 1, (M, d)), rng.uniform(-1, 1, (nmc, d))
    return X, C, Z


def c5_lite_inputs(N=16384, d=10, seed=65536, noise=0.1):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (N, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(noise) * rng.standard_normal(N)
    return X, y


def sqdist(A, B, w=None, out=None, rows=2048):
    """sum_k w_k (a_ik - b_jk)^2 from coordinate differences, row chunks (no 3-D temporary)."""
    na, nb = len(A), len(B)
    R = np.empty((na, nb)) if out is None else out
    w = np.ones(A.shape[1]) if w is None else w
    for i0 in range(0, na, rows):
        i1 = min(na, i0 + rows)
        acc = R[i0:i1]
        acc[:] = 0.0
        for k in range(A.shape[1]):
            df = A[i0:i1, k][:, None] - B[None, :, k]
            df *= df
            if w[k] != 1.0:
                df *= w[k]
            acc += df
    return R


def kern(kind, R, rho=0.5, s=1.0):
    """In place on the (weighted) squared distances."""
    if kind == "se":
        R *= -0.5
        np.exp(R, out=R)
    else:
        c = 5.0 if kind == "matern52" else 3.0
        np.sqrt(R, out=R)
        R *= np.sqrt(c) / rho
        t = R
        e = np.exp(-t)
        if kind == "matern52":
            p = 1.0 + t + t * t / 3.0
        else:
            p = 1.0 + t
        np.multiply(p, e, out=R)
    if s != 1.0:
        R *= s
    return R


NB_HOST = 4096


def chol_blocked(K, nb=NB_HOST):
    """Lower Cholesky factor in place, right-looking over nb-order blocks: LAPACK dpotrf on the diagonal blocks, dtrsm for the
    block column, dgemm for the trailing blocks on / below the diagonal.  (One dpotrf call on the whole 32768-order matrix
    segfaults inside this image's scipy-openblas; every call here stays at block size.)  Blocks above the diagonal keep K."""
    N = len(K)
    for j0 in range(0, N, nb):
        j1 = min(N, j0 + nb)
        K[j0:j1, j0:j1] = sl.cholesky(K[j0:j1, j0:j1], lower=True, check_finite=False)
        L11 = K[j0:j1, j0:j1]
        for i0 in range(j1, N, nb):
            i1 = min(N, i0 + nb)
            K[i0:i1, j0:j1] = sl.solve_triangular(L11, K[i0:i1, j0:j1].T, lower=True, check_finite=False).T
        for i0 in range(j1, N, nb):
            i1 = min(N, i0 + nb)
            K[i0:i1, j1:i1] -= K[i0:i1, j0:j1] @ K[j1:i1, j0:j1].T
    return K


def fwd_blocked(L, B, nb=NB_HOST):
    """W = L^-1 B (B: N x m or N), block forward substitution (only blocks on / below the diagonal of L are read)."""
    N = len(L)
    W = np.empty_like(B)
    for i0 in range(0, N, nb):
        i1 = min(N, i0 + nb)
        Bi = B[i0:i1] - (L[i0:i1, :i0] @ W[:i0] if i0 else 0.0)
        W[i0:i1] = sl.solve_triangular(L[i0:i1, i0:i1], Bi, lower=True, check_finite=False)
    return W


def bwd_blocked(L, u, nb=NB_HOST):
    """x = L^-T u for a vector."""
    N = len(L)
    x = np.empty_like(u)
    starts = list(range(0, N, nb))
    for i0 in reversed(starts):
        i1 = min(N, i0 + nb)
        r = u[i0:i1] - (L[i1:, i0:i1].T @ x[i1:] if i1 < N else 0.0)
        x[i0:i1] = sl.solve_triangular(L[i0:i1, i0:i1], r, lower=True, trans="T", check_finite=False)
    return x


def c4_case(kind):
    case = "c4_" + kind
    N, d, noise, nsub, nvar = 32768, 8, 0.1, 4096, 256
    X, y, Z = c4_inputs()
    t0 = time.time()
    K = kern(kind, sqdist(X, X))
    K[np.diag_indices(N)] += noise
    print("%s: assembly %.0f s" % (case, time.time() - t0), flush=True)
    L = chol_blocked(K)
    print("%s: factor %.0f s" % (case, time.time() - t0), flush=True)
    logdet = 2.0 * float(np.sum(np.log(np.diag(L))))
    u = fwd_blocked(L, y)
    yta = float(u @ u)
    alpha = bwd_blocked(L, u)
    ll = -0.5 * yta - 0.5 * logdet - N / 2.0 * np.log(2 * np.pi)
    W = fwd_blocked(L, kern(kind, sqdist(X, Z[:nsub])))
    var = 1.0 - np.sum(W * W, axis=0)
    del W
    mean = kern(kind, sqdist(Z[:nvar], X)) @ alpha
    print("%s: done %.0f s  loglike %.15g  ivar %.15g" % (case, time.time() - t0, ll, abs(var.mean())), flush=True)
    return {case + "/loglike": ll, case + "/logdet": logdet, case + "/yTalpha": yta, case + "/ivar4096": abs(float(var.mean())),
            case + "/var256": var[:nvar].copy(), case + "/mean256": mean, case + "/alpha_head": alpha[:256].copy()}


def c3_case():
    case = "c3"
    N, d, noise, ncand = 16384, 8, 0.1, 256
    X, C, Z = c3_inputs()
    w = (0.4 + 0.05 * np.arange(d)) ** -2.0
    t0 = time.time()
    K = kern("se", sqdist(X, X, w))
    K[np.diag_indices(N)] += noise
    L = sl.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    cand = np.arange(ncand) * (len(C) // ncand) + 7             # 256 fixed candidates spread over the 65536
    Wz = sl.solve_triangular(L, kern("se", sqdist(X, Z, w)), lower=True, check_finite=False, overwrite_b=True)
    Wc = sl.solve_triangular(L, kern("se", sqdist(X, C[cand], w)), lower=True, check_finite=False, overwrite_b=True)
    varz = 1.0 - np.sum(Wz * Wz, axis=0)
    iv0 = float(varz.mean())
    cov = kern("se", sqdist(Z, C[cand], w)) - Wz.T @ Wc          # posterior covariance cov(z, c | design)
    varc = 1.0 - np.sum(Wc * Wc, axis=0) + noise                 # the refit's pivot: prior + nugget - |w_c|^2
    costs = np.abs(iv0 - np.mean(cov * cov, axis=0) / varc)      # IVAR after adding candidate c (experimentalDesign.py:104-117)
    print("c3: done %.0f s  ivar0 %.15g  min cost %.15g" % (time.time() - t0, iv0, costs.min()), flush=True)
    return {case + "/ivar0": abs(iv0), case + "/cand_index": cand.astype(np.int64), case + "/cand_cost": costs}


def c5_lite_case():
    case = "c5_lite"
    N, d, noise, s = 16384, 10, 0.1, 1.0
    X, y = c5_lite_inputs()
    cl = 0.5 + 0.03 * np.arange(d)
    t0 = time.time()
    K0 = kern("se", sqdist(X, X, cl ** -2.0))
    K = K0.copy()
    K[np.diag_indices(N)] += noise
    L = sl.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    logdet = 2.0 * float(np.sum(np.log(np.diag(L))))
    alpha = sl.cho_solve((L, True), y, check_finite=False)
    ll = -0.5 * float(y @ alpha) - 0.5 * logdet - N / 2.0 * np.log(2 * np.pi)
    P, info = sl.lapack.dpotri(L, lower=1, overwrite_c=1)        # lower triangle of K^-1
    assert info == 0
    print("c5_lite: dpotri %.0f s" % (time.time() - t0), flush=True)
    trP = float(np.trace(P))
    P = np.tril(P) + np.tril(P, -1).T                            # symmetrise
    T = np.outer(alpha, alpha)
    T -= P
    del P
    T *= K0                                                      # T o K0 : every length-scale trace reuses it
    g = np.zeros(d + 2)
    for k in range(d):                                           # dK/dl_k = K0 o D_k^2 / l_k^3 (kernels.py:125-144)
        D = X[:, k][:, None] - X[None, :, k]
        D *= D
        g[k] = 0.5 * float(np.sum(T * D)) / cl[k] ** 3
    g[d] = 0.5 * float(np.sum(T)) / s                            # dK/d signalSize = K0 / s
    g[d + 1] = 0.5 * (float(alpha @ alpha) - trP)                # d/d noise (raw): 1/2 tr((alpha alpha^T - K^-1) I)
    print("c5_lite: done %.0f s  loglike %.15g  grad %s" % (time.time() - t0, ll, g), flush=True)
    return {case + "/loglike": ll, case + "/logdet": logdet, case + "/grad": g}


def c5_full_case():
    """BASELINE config 5 at its FULL size (N = 65536, d = 10 ARD-SE l_k = 0.5 + 0.03 k, noise = 0.1, seed 65536: the inputs of
    tests/test_gpu_scale.py::test_c5_full_size_fit_on_one_gpu): log-marginal, log det, y^T alpha, 256 entries of alpha and the
    posterior variance at the first 256 training points.  K is 34.4 GB in host memory; the factorisation is chol_blocked."""
    case = "c5_full"
    N, d, noise, nvar = 65536, 10, 0.1, 256
    rng = np.random.default_rng(65536)
    X = rng.uniform(-1, 1, (N, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(noise) * rng.standard_normal(N)
    cl = 0.5 + 0.03 * np.arange(d)
    t0 = time.time()
    K = np.empty((N, N))
    for i0 in range(0, N, 4096):                     # row chunks: in place, no second N x N array
        blk = K[i0:i0 + 4096]
        sqdist(X[i0:i0 + 4096], X, cl ** -2.0, out=blk)
        blk *= -0.5
        np.exp(blk, out=blk)
    K[np.diag_indices(N)] += noise
    print("%s: assembly %.0f s" % (case, time.time() - t0), flush=True)
    L = chol_blocked(K)
    print("%s: factor %.0f s" % (case, time.time() - t0), flush=True)
    logdet = 2.0 * float(np.sum(np.log(np.diag(L))))
    u = fwd_blocked(L, y)
    yta = float(u @ u)
    alpha = bwd_blocked(L, u)
    ll = -0.5 * yta - 0.5 * logdet - N / 2.0 * np.log(2 * np.pi)
    Kz = kern("se", sqdist(X, X[:nvar], cl ** -2.0))
    W = fwd_blocked(L, Kz)
    var = 1.0 - np.sum(W * W, axis=0)
    print("%s: done %.0f s  loglike %.15g  logdet %.15g" % (case, time.time() - t0, ll, logdet), flush=True)
    return {case + "/loglike": ll, case + "/logdet": logdet, case + "/yTalpha": yta, case + "/alpha_head": alpha[:256].copy(),
            case + "/var256_at_training_points": var}


def main():
    want = sys.argv[1:] or ["c4_matern52", "c4_matern32", "c3", "c5_lite"]
    arrays = dict(np.load(NPZ)) if os.path.exists(NPZ) else {}
    for name in want:
        if name.startswith("c4_"):
            arrays.update(c4_case(name[3:]))
        elif name == "c3":
            arrays.update(c3_case())
        elif name == "c5_lite":
            arrays.update(c5_lite_case())
        elif name == "c5_full":
            arrays.update(c5_full_case())
        np.savez(NPZ, **{k: np.asarray(v) for k, v in arrays.items()})
    index = {k: dict(shape=list(np.shape(v)), source=LABEL) for k, v in sorted(arrays.items())}
    with open(JSN, "w") as f:
        json.dump(dict(generator="tests/golden/make_golden_r6.py", label=LABEL, arrays=index,
                       numpy=np.__version__, scipy=__import__("scipy").__version__), f, indent=1)
    print("wrote %s (%d arrays)" % (NPZ, len(arrays)))


if __name__ == "__main__":
    main()
