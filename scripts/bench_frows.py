#!/usr/bin/env python3
"""Performance evidence for the SURVEY.md 8 f-rows and the design kernels: one JSON line per
entry point with its size, best-of-3 time, the algorithmic work and the achieved rate against the roof that bounds it.
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel table (profiles/r03_frows_kernel_stats.csv).

    python scripts/bench_frows.py [--quick] [--only f1,design,mi,fitc,refit,f3]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpexp_amd import device as dev  # noqa: E402

PEAK_TF, PEAK_GBS = 78.6, 8000.0
ctx = dev.context()
quick = "--quick" in sys.argv
only = set(sys.argv[sys.argv.index("--only") + 1].split(",")) if "--only" in sys.argv else None


def want(section):
    return only is None or section in only


def best(f, reps=3):
    ts = []
    out = None
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        out = f()
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    return out, min(ts)


def report(name, replaces, size, t, flops=None, nbytes=None, note=""):
    line = {"entry": name, "replaces": replaces, "size": size, "ms": 1e3 * t}
    if flops is not None:
        line.update(bound="mfma", algorithmic_flop=flops, achieved_tflops=flops / t / 1e12, frac=flops / t / 1e12 / PEAK_TF)
    if nbytes is not None:
        line.update(bound="hbm", algorithmic_bytes=nbytes, achieved_gbs=nbytes / t / 1e9, frac=nbytes / t / 1e9 / PEAK_GBS)
    if note:
        line["note"] = note
    print(json.dumps(line), flush=True)


rng = np.random.default_rng(8192)
d = 8
sp = dev.KernelSpec(dev.K_SE, d, list(0.4 + 0.05 * np.arange(d)) + [1.0])
# ---- f1: gradients of the posterior variance w.r.t. point locations (gp.py:261-341, experimentalDesign.py:168-179)
if want("f1"):
    N, M = (2048, 8192) if quick else (8192, 32768)
    Xh, Zh = rng.uniform(-1, 1, (N, d)), rng.uniform(-1, 1, (M, d))
    X, Z = dev.points(ctx, Xh), dev.points(ctx, Zh)
    L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
    _, t = best(lambda: dev.ivar_grad(ctx, sp, L, X, Z))
    report("gpx_ivar_grad", "costFunctionGP_IVAR.derivative v1 (experimentalDesign.py:168-179)", dict(N=N, M=M, d=d), t,
           flops=3.0 * N * N * M, note="ONE count, in SURVEY 8d's units (a triangular solve of an N x M block = N^2 M): beta = K^-1 K(X,Z) "
           "is two solves (2 N^2 M), S = beta beta^T as a lower SYRK is N^2 M more; tr(dK/dx S) needs S")
    _, t = best(lambda: dev.var_grad_newpt(ctx, sp, L, X, Z))
    report("gpx_var_grad_newpt", "GP.evaluateVarianceDerivWRTnewpt (gp.py:261-280)", dict(N=N, M=M, d=d), t, flops=2.0 * N * N * M)
    # one optimiser iteration of a continuous design (SLSQP, experimentalDesign.py:471-489): cost, then gradient, same design
    from gpExp.kernels import KernelSquaredExponential
    from gpExp.gp import GP
    from gpExp.approximation import Space
    from gpExp.experimentalDesign import costFunctionGP_IVAR
    space = Space(d, lambda size: rng.uniform(-1, 1, size), lambda p: np.ones(len(p)))
    def iteration(reuse):
        g = GP(KernelSquaredExponential(list(0.4 + 0.05 * np.arange(d)), 1.0, d), 0.1)
        g.reuseFactor = reuse
        cf = costFunctionGP_IVAR(g, N, space, mcPoints=Zh)
        cf.evaluate(Xh); cf.derivative(Xh)                      # warm: pools, block inverses
        Xq = Xh * 0.999                                         # a new design in EVERY point: one full fit, one full forward solve
        def both():
            cf.evaluate(Xq)
            return cf.derivative(Xq)
        return best(both, reps=1)[1]
    def batch_move(reuse, nb_=512):
        g = GP(KernelSquaredExponential(list(0.4 + 0.05 * np.arange(d)), 1.0, d), 0.1)
        g.reuseFactor = reuse
        cf = costFunctionGP_IVAR(g, N, space, mcPoints=Zh)
        cf.evaluate(Xh)
        ts = []
        for rep in range(3):
            Xq = Xh.copy(); Xq[-nb_:] = np.random.default_rng(rep).uniform(-1, 1, (nb_, d))
            ts.append(best(lambda: cf.evaluate(Xq), reps=1)[1])
        return min(ts)
    def batch_gradient(pinned, nb_=512):
        g = GP(KernelSquaredExponential(list(0.4 + 0.05 * np.arange(d)), 1.0, d), 0.1)
        cf = costFunctionGP_IVAR(g, N, space, mcPoints=Zh)
        cf.pinnedPoints = pinned
        cf.evaluate(Xh); cf.derivative(Xh)
        Xq = Xh.copy(); Xq[-nb_:] = np.random.default_rng(7).uniform(-1, 1, (nb_, d))
        cf.evaluate(Xq)
        return best(lambda: cf.derivative(Xq), reps=2)[1]
    t_gall, t_gfree = batch_gradient(0), batch_gradient(N - 512)
    report("design gradient of the last 512 points", "costFunctionGP_IVAR.derivative inside the batch loop (experimentalDesign.py:"
           "694-751: earlier batches pinned by equal bounds, :719-724)", dict(N=N, M=M, free=512, d=d), t_gfree,
           flops=2.0 * 512 * N * M + 512.0 * 512 * M + 512.0 * N * N,
           note="beta_T = L_TT^-T W_T, S_T = (beta_T W^T) L^-1 from the kept solve, row kernel for the free points: %.1f ms for "
                "all points (kept forward solve), %.1f ms for the free ones" % (1e3 * t_gall, 1e3 * t_gfree))
    t_full, t_inc = batch_move(False), batch_move(True)
    report("design cost after moving the last 512 points", "costFunctionGP_IVAR.evaluate inside the batch loop "
           "(experimentalDesign.py:694-751: earlier batches pinned)", dict(N=N, M=M, moved=512, d=d), t_inc,
           flops=2.0 * 512 * (N - 512) * M + 512.0 * 512 * M + 512.0 * N * N,
           note="refit of the moved rows (rows N^2) + W2 = L22^-1 (K(X2, Z) - L21 W1) on the kept solve; %.1f ms when every evaluation "
                "refits and solves from scratch (reuseFactor = False), %.1f ms incrementally" % (1e3 * t_full, 1e3 * t_inc))
    t_sep, t_shared = iteration(False), iteration(True)
    report("design iteration: cost + gradient", "costFunctionGP_IVAR.evaluate + .derivative at one design (experimentalDesign.py:100-117, "
           "168-179, 471-489)", dict(N=N, M=M, d=d), t_shared, flops=3.0 * N * N * M + N ** 3 / 3.0,
           note="count = what one iteration needs: ONE fit (N^3/3), the forward solve once (N^2 M), backward solve and SYRK "
                "(2 N^2 M); %.1f ms when the gradient call refits and solves forward again (reuseFactor = False: two fits, "
                "4 N^2 M), %.1f ms with the kept factor and the kept forward solve" % (1e3 * t_sep, 1e3 * t_shared))
    Nv, Mv = (1024, 1024) if quick else (2048, 4096)
    Xv, Zv = dev.points(ctx, Xh[:Nv]), dev.points(ctx, Zh[:Mv])
    Lv = dev.potrf(ctx, dev.kfill(ctx, sp, Xv, nugget=0.1))
    _, t = best(lambda: dev.var_grad(ctx, sp, Lv, Xv, Zv), reps=2)
    report("gpx_var_grad", "GP.evaluateVarianceDerivative (gp.py:282-341)", dict(N=Nv, M=Mv, d=d), t, flops=2.0 * (d + 1) * Nv * Nv * Mv,
           nbytes=8.0 * Nv * d * Mv,
           note="the (N d) x M result goes to the HOST by contract (that is what the reference returns): %.2f GB over PCIe dominate; "
                "one N x N matrix dK_l is refilled per coordinate (not d of them)" % (8.0 * Nv * d * Mv / 1e9))
    del L, Lv
# ---- a13-a15 / 8c: design kernels at C3 / C5 sizes
if want("design"):
    N3, M3, nmc = (4096, 16384, 1024) if quick else (16384, 65536, 4096)
    rng = np.random.default_rng(16384)
    X3h, C3h, Z3h = rng.uniform(-1, 1, (N3, d)), rng.uniform(-1, 1, (M3, d)), rng.uniform(-1, 1, (nmc, d))
    X3, C3, Z3 = dev.points(ctx, X3h), dev.points(ctx, C3h), dev.points(ctx, Z3h)
    K3 = dev.potrf(ctx, dev.kfill(ctx, sp, X3, nugget=0.1))
    _, t = best(lambda: dev.greedy_ivar_step(ctx, sp, K3, X3, C3, Z3, 0.1))
    report("gpx_greedy_ivar_step", "costFunctionGP_IVAR.evaluate per candidate (experimentalDesign.py:79-117; SURVEY 8c composition)",
           dict(N=N3, candidates=M3, nMC=nmc, d=d), t, flops=2.0 * nmc * M3 * N3 + 1.0 * N3 * N3 * (M3 + nmc),
           note="rank-one scoring of every candidate: two N x (M + nMC) triangular solves + the nMC x M x N product")
    _, t = best(lambda: dev.greedy_var(ctx, sp, C3, 16))
    report("gpx_greedy_var", "performGreedyVarExperimentalDesign (experimentalDesign.py:787-845)", dict(candidates=M3, picks=16, d=d), t,
           nbytes=8.0 * M3 * 16 * 16 / 2 + 8.0 * M3 * d * 16, note="incremental Cholesky rows; latency-bound (one launch chain per pick)")
    del K3
if want("mi"):
    d5 = 10
    sp5 = dev.KernelSpec(dev.K_SE, d5, list(0.5 + 0.03 * np.arange(d5)) + [1.0])
    Mm = 2048 if quick else 8192
    Cm = dev.points(ctx, rng.uniform(-1, 1, (Mm, d5)))
    _, t = best(lambda: dev.mi_greedy(ctx, sp5, Cm, 0.1, 8, 0))
    report("gpx_mi_greedy", "costFunctionGP_MI + performGreedyMIExperimentalDesign (experimentalDesign.py:223-285, 753-785)",
           dict(candidates=Mm, picks=8, d=d5), t, flops=Mm ** 3 / 3.0 + 2.0 * Mm ** 3 / 3.0,
           note="SURVEY 8d count: potrf M^3/3 + potri 2 M^3/3 = M^3 (round 4 charged a dense M^3 product on top that the code no "
                "longer executes); then 7 rank-one down-dates (HBM: 16 M^2 bytes each)")
# ---- f4: FITC
if want("fitc"):
    Nf = 8192 if quick else 32768
    nu = Nf // 8
    Xf = rng.uniform(-1, 1, (Nf, d))
    yf = np.sin(2 * np.pi * Xf.sum(1) / d) + np.sqrt(0.1) * rng.standard_normal(Nf)
    Xfd, Sd = dev.points(ctx, Xf), dev.points(ctx, Xf[rng.permutation(Nf)[:nu]].copy())
    m, t = best(lambda: dev.FitcModel(ctx, sp, Xfd, Sd, 0.1), reps=2)
    report("gpx_fitc_fit", "FITC branch of addNodesAndComputeCovariance (gp.py:182-210; gp_kernel_utilities.py:70-104)",
           dict(N=Nf, nu=nu, d=d), t, flops=2.0 * nu ** 3 / 3.0 + 2.0 * Nf * nu * nu,
           note="chol(Quu), Kuf, G = diag(K - Q), chol(Quu + Kuf G^-1 Kfu): two nu-order factorisations + two nu x nu x N products")
    _, t = best(lambda: m.solve(yf))
    report("gpx_fitc_solve", "GP.train with the Woodbury precision (gp.py:100-101)", dict(N=Nf, nu=nu), t, nbytes=2.0 * 8.0 * Nf * nu)
# ---- f2: refit of the changed rows
if want("refit"):
    Nr = 4096 if quick else 16384
    Xr = rng.uniform(-1, 1, (Nr, d))
    Xrd = dev.points(ctx, Xr)
    Lr = dev.potrf(ctx, dev.kfill(ctx, sp, Xrd, nugget=0.1))
    keep = Nr - 512
    Xr2 = Xr.copy()
    Xr2[keep:] = rng.uniform(-1, 1, (Nr - keep, d))
    Xr2d = dev.points(ctx, Xr2)
    _, t = best(lambda: dev.refit_rows(ctx, sp, Xr2d, 0.1, Lr, keep))
    report("gpx_refit_rows", "ExperimentalDesignGreedyWithDerivatives batch loop (experimentalDesign.py:694-751)",
           dict(N=Nr, changed_rows=Nr - keep), t, flops=1.0 * (Nr - keep) * Nr * Nr,
           note="the leading %d rows of the factor are reused; only the last %d rows are re-assembled and re-solved" % (keep, Nr - keep))
# ---- f3: the hyper-parameter loop (findOptParamsLogLike / chooseParams, gp.py:498-639)
if want("f3"):
    from gpExp.kernels import KernelSquaredExponential, KernelIsoMatern
    from gpExp.gp import GP

    def f3_case(label, kern, N3, d3, maxiter):
        r3 = np.random.default_rng(N3)
        Xo = r3.uniform(-1, 1, (N3, d3))
        yo = np.sin(2 * np.pi * Xo.sum(1) / d3) + np.sqrt(0.1) * r3.standard_normal(N3)
        out = {}
        for analytic in (False, True):
            g = GP(kern(), 0.1)
            counts = dict(fits=0, grads=0)
            inner = g.loglikeParams

            def counted(pts, evals, returnDeriv=0, noiseIn=None):
                counts["fits"] += 1
                counts["grads"] += int(returnDeriv == 1)
                return inner(pts, evals, returnDeriv=returnDeriv, noiseIn=noiseIn)

            g.loglikeParams = counted
            g.loglikeParams(Xo, yo)                        # warm: pools, kernel attributes
            counts.update(fits=0, grads=0)
            ctx.sync()
            t0 = time.perf_counter()
            params, opt = g.findOptParamsLogLike(Xo, yo, maxiter=maxiter, analyticGradient=analytic)
            ctx.sync()
            out[analytic] = (time.perf_counter() - t0, dict(counts), float(opt))
        (tn, cn, on), (ta, ca, oa) = out[False], out[True]
        nhyp = len(kern().hyperParam) + 1
        # an L-BFGS-B iterate with numerical gradients = 1 + nhyp likelihood evaluations (approx_grad: forward differences)
        report("findOptParamsLogLike (%s)" % label, "GP.findOptParamsLogLike / chooseParams (gp.py:498-639)",
               dict(N=N3, d=d3, hyper_parameters=nhyp, maxfun=maxiter), ta / max(ca["grads"], 1),
               flops=N3 ** 3 / 3.0 + 2.0 * N3 ** 3 / 3.0,
               note="ms per likelihood + gradient evaluation with analyticGradient=True (one factorisation N^3/3 + the trace "
                    "through L^-1, 2 N^3/3): %d evaluations in %.2f s, optimum %.6g; numerical gradients as in the reference "
                    "(maxfun = %d likelihood evaluations = %d factorisations, about %d per L-BFGS iterate): %.2f s, %.1f ms per "
                    "factorisation, optimum %.6g" % (ca["grads"], ta, -oa, maxiter, cn["fits"], nhyp + 1, tn, 1e3 * tn / max(cn["fits"], 1), -on))

    if quick:
        f3_case("C2-lite: N=1024 d=3 iso-SE", lambda: KernelSquaredExponential([0.5], 1.0, 3), 1024, 3, 20)
    else:
        f3_case("C2: N=4096 d=3 iso-SE", lambda: KernelSquaredExponential([0.5], 1.0, 3), 4096, 3, 40)
        f3_case("N=16384 d=8 Matern-5/2", lambda: KernelIsoMatern(0.7, 1.0, 8, nu=2.5), 16384, 8, 24)
        f3_case("N=16384 d=8 ARD-SE", lambda: KernelSquaredExponential(list(0.5 + 0.05 * np.arange(8)), 1.0, 8), 16384, 8, 24)
