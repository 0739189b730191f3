#!/usr/bin/env python3
"""bench.py -- GP-fit + IVAR-eval on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (SURVEY.md 8d, config C4 -- the configuration the metric is quoted on; it fits one GPU):
    N = 32768 training points, d = 8, Matern nu=5/2 (rho=0.5, signalSize=1, noise=0.1), seed 32768,
    X ~ U[-1,1]^(N x d), y = sin(2 pi sum(x)/d) + sqrt(noise) N(0,1), M = 32768 MC points ~ U[-1,1]^(M x d).
One step = the whole hot path on that batch, inputs already resident in HBM:
    K = kfill(X) + noise I  ->  L = potrf(K)  ->  alpha = potrs(L, y)  ->  logdet  ->  log marginal likelihood
    ->  IVAR = mean_z [k(z,z) - |L^-1 k(X,z)|^2] over the M points.
value = (N + M) points / step time (whole job, all ranks); ms_per_step is the GP-fit + IVAR-eval wall time.
With --gpus N > 1 the same total problem is split over the ranks (strong scaling): see gpexp_amd/dist.py.  Each launched rank
process then only SUPERVISES a child that does the work (it never touches the GPU): a first-contact preflight that hangs on the
2-D layout is retried once on the 1-D layout instead of ending the run (supervise()).

Extra objects on the JSON line: "roofline" for the dominant kernel (the fp64 MFMA GEMM behind SYRK/TRSM; HIP
events recorded on the launch stream by the library around every launch inside the timed region) and
"cpu_baseline" (the oracle's reference-exact algorithm -- row-loop assembly + pinv + slogdet + per-point variance
loop -- timed on the host cores on a bounded sample of the same workload, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
os.environ.setdefault("NCCL_DEBUG", "WARN")                 # RCCL says why when a collective fails (stderr; stdout stays one JSON line)
if int(os.environ.get("WORLD_SIZE", "1")) == 1:
    # the multi_gpu_replay object (N = 1 only, after the timed region) reads time stamps off the distributed loop's events; the
    # single-GPU timed region records none of them
    os.environ.setdefault("GPX_EVENT_TIMING", "1")
os.environ.setdefault("GPX_DIST_ATTACH", "0")               # bench.py drives the distributed runners itself: the class API of this
#                                                             process must not attach to the process group on its own (dist.session)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # 256 CU x 2.4 GHz x 128 flop/clk/CU (MI355X fp64 matrix = vector peak)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E


def hdot(a, b):
    """sum a_i b_i without BLAS (gpexp_amd._lib.hdot: np.dot wakes OpenBLAS' 64 threads, whose spinning stalls the next step's kernel
    launches in a CPU-quota'd container)."""
    return float(np.sum(np.multiply(a, b)))


def workload(n, d, m, seed):
    rng = np.random.default_rng(seed)
    noise = 0.1
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + np.sqrt(noise) * rng.standard_normal(n)
    Z = rng.uniform(-1, 1, (m, d))
    return X, y, Z, noise


def _cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas, threads = "unknown", os.cpu_count() or 1
    try:
        import threadpoolctl
        pools = [p for p in threadpoolctl.threadpool_info() if p.get("user_api") == "blas"]
        if pools:
            blas = "%s %s" % (pools[0].get("internal_api", "?"), pools[0].get("version", "?"))
            threads = max(p.get("num_threads", 1) for p in pools)
    except Exception:
        pass
    return model, blas, int(threads)


def _ref_exact(n, m, d, seed, kind="matern52"):
    """One pass of the REFERENCE ALGORITHM (the oracle's restatement) at N=n: row-loop assembly
    (gp_kernel_utilities.py:56-60) + pinv (gp.py:181) + slogdet (gp.py:434) + per-point variance loop (gp.py:246-256)."""
    from oracle import gpexp_oracle as orc
    X, y, Z, noise = workload(n, d, m, seed)
    spec = dict(kind=kind, rho=0.5, signalSize=1.0, d=d)
    t0 = time.perf_counter()
    K = orc.cov_matrix(spec, X, noise, row_loop=True)
    t_fill = time.perf_counter() - t0
    P = np.linalg.pinv(K)
    _, logdet = np.linalg.slogdet(K)
    alpha = P @ y
    ll = -0.5 * y @ alpha - 0.5 * logdet - n / 2.0 * np.log(2 * np.pi)
    t_fit = time.perf_counter() - t0
    _, var = orc.posterior(spec, dict(K=K, P=P, X=X), Z)
    t_all = time.perf_counter() - t0
    return dict(N=n, M=m, fill_s=t_fill, fit_s=t_fit, ivar_s=t_all - t_fit, total_s=t_all, loglike=float(ll),
                ivar=float(abs(var.mean())))


def _fair_chol(n, m, d, seed, kind="matern52"):
    """The "fair CPU" line of SURVEY.md 8d: the algorithm the GPU runs (vectorised fill + LAPACK Cholesky + triangular
    solves) on the host cores."""
    import scipy.linalg as sl
    from oracle import gpexp_oracle as orc
    X, y, Z, noise = workload(n, d, m, seed)
    spec = dict(kind=kind, rho=0.5, signalSize=1.0, d=d)

    def kern(r2):
        if kind == "matern52":
            t = np.sqrt(5.0 * r2) / 0.5
            return (1.0 + t + t * t / 3.0) * np.exp(-t)
        t = np.sqrt(3.0 * r2) / 0.5          # Matern-3/2, kernels.py:85-89
        return (1.0 + t) * np.exp(-t)

    t0 = time.perf_counter()
    r2 = np.maximum(((X * X).sum(1)[:, None] + (X * X).sum(1)[None, :] - 2.0 * X @ X.T), 0.0)
    K = kern(r2)
    K[np.diag_indices(n)] += noise
    c = sl.cho_factor(K, lower=True, overwrite_a=True, check_finite=False)
    alpha = sl.cho_solve(c, y, check_finite=False)
    ll = -0.5 * y @ alpha - np.sum(np.log(np.diag(c[0]))) - n / 2.0 * np.log(2 * np.pi)
    t_fit = time.perf_counter() - t0
    kz = orc.cross_matrix(spec, Z, X).T if m <= 4096 else None
    if kz is None:
        r2 = np.maximum(((X * X).sum(1)[:, None] + (Z * Z).sum(1)[None, :] - 2.0 * X @ Z.T), 0.0)
        kz = kern(r2)
    W = sl.solve_triangular(c[0], kz, lower=True, check_finite=False, overwrite_b=True)
    iv = abs(np.mean(1.0 - np.sum(W * W, axis=0)))
    t_all = time.perf_counter() - t0
    return dict(N=n, M=m, fit_s=t_fit, ivar_s=t_all - t_fit, total_s=t_all, loglike=float(ll), ivar=float(iv))


T_START = time.perf_counter()   # the default run keeps to a wall-clock budget: see cpu_baseline


def cpu_baseline(d, full=False, kind="matern52", budget_s=360.0):
    """SURVEY.md 8d protocol: the reference algorithm at N = 2048, 4096 and -- when the run's wall-clock budget allows (it
    predicts the N = 8192 run from the N = 4096 one, x 8.5; 163 s on 64 threads) -- 8192 (M = 512 evaluation points), a
    least-squares fit of t = c N^3 through the measured fits, and the reference's time at the BENCH configuration (N = M = 32768)
    from that fit.  `value` is that like-for-like figure -- EXTRAPOLATED, labelled so: the reference's pinv needs ~3 h at
    N = 32768 --, `sample_value` the measured points/s at the largest measured N.  `--cpu-baseline full` forces N = 8192 and adds
    the fair-CPU Cholesky line at the full N = 32768."""
    model, blas, threads = _cpu_info()
    runs = []
    for n in (2048, 4096, 8192):   # a line per size on stderr: the protocol runs for minutes, and a silent job looks hung
        if n == 8192 and not full:
            predicted = 8.5 * runs[-1]["total_s"] + 15.0   # + the fair-CPU line below
            if (time.perf_counter() - T_START) + predicted > budget_s:
                print("bench.py: cpu_baseline: N=8192 skipped (predicted %.0f s, %.0f s of %.0f used)"
                      % (predicted, time.perf_counter() - T_START, budget_s), file=sys.stderr, flush=True)
                break
        print("bench.py: cpu_baseline: reference algorithm at N=%d ..." % n, file=sys.stderr, flush=True)
        runs.append(_ref_exact(n, 512, d, seed=n, kind=kind))
        print("bench.py: cpu_baseline: N=%d took %.1f s" % (n, runs[-1]["total_s"]), file=sys.stderr, flush=True)
    sizes = [r["N"] for r in runs]
    c3 = float(np.sum([r["fit_s"] * r["N"] ** 3 for r in runs]) / np.sum([float(r["N"]) ** 6 for r in runs]))
    civ = float(np.mean([r["ivar_s"] / (r["N"] ** 2 * r["M"]) for r in runs]))
    big = runs[-1]
    print("bench.py: cpu_baseline: fair-CPU Cholesky at N=%d ..." % (32768 if full else 8192), file=sys.stderr, flush=True)
    fair = _fair_chol(32768 if full else 8192, 32768 if full else 2048, d, seed=32768 if full else 8192, kind=kind)
    NB, MB = 32768, 32768
    fit_x, ivar_x = c3 * float(NB) ** 3, civ * float(NB) ** 2 * MB
    return dict(value=(NB + MB) / (fit_x + ivar_x), unit="points/s", cores=threads, kind="port",
                sample="reference algorithm (oracle: row-loop fill + pinv + slogdet + per-point variance loop) measured at N=%s, "
                       "M=512, d=%d %s; value = points/s at the bench configuration N=M=32768 EXTRAPOLATED from t_fit = c N^3 "
                       "(%d sizes) and t_ivar ~ N^2 M: %.0f s + %.0f s; sample_value = measured at N=%d (%.1f s)"
                       % (sizes, d, kind, len(sizes), fit_x, ivar_x, big["N"], big["total_s"]),
                sample_value=(big["N"] + big["M"]) / big["total_s"], value_is_extrapolated=True,
                cpu_model=model, blas=blas, blas_threads=threads, host_cores=os.cpu_count(), runs=runs,
                fit="t_fit = c N^3, c = %.3e s (least squares over the measured N = %s)" % (c3, sizes),
                extrapolated={"N": NB, "M": MB, "fit_s": fit_x, "ivar_s": ivar_x,
                              "note": "EXTRAPOLATED from the fit, not measured (SURVEY.md 8d)"},
                fair_cpu_chol=dict(fair, note="vectorised fill + LAPACK potrf/potrs + TRSM, the algorithm the GPU runs"),
                seconds=float(sum(r["total_s"] for r in runs) + fair["total_s"]))


def bench_c5(args, ctx, dev, world, rank, stdout_fd):
    """BASELINE config C5: N = 65536, d = 10, ARD-SE l_k = 0.5 + 0.03 k, s = 1, noise = 0.1, seed 65536.
    One step = assembly + factorisation + alpha + log marginal likelihood + its gradient w.r.t. the 10 length scales, signalSize
    and noise (gp.py:444-466) + greedy MI design, 8 picks over M = 8192 candidates (BASELINE fixes no M for MI).  One GPU: the
    single-GPU path with the gradient traces accumulated slab by slab (no N x N inverse); N GPUs: DistFitGrad2D (2-D
    block-cyclic fit, traces and MI scoring sharded).  Not the headline metric: one JSON line for the record."""
    N = 65536 if args.n is None else args.n
    d = 10 if args.d is None else args.d
    M = 8192 if args.m is None else args.m
    rng = np.random.default_rng(N)
    noise = 0.1
    Xh = rng.uniform(-1, 1, (N, d))
    yh = np.sin(2 * np.pi * Xh.sum(1) / d) + np.sqrt(noise) * rng.standard_normal(N)
    Ch = rng.uniform(-1, 1, (M, d))
    spec = dev.KernelSpec(dev.K_SE, d, [0.5 + 0.03 * k for k in range(d)] + [1.0])
    times = {}
    if world > 1 or os.environ.get("GPX_FORCE_DIST") == "1":
        from gpexp_amd import dist
        fc_timer, fc = first_contact_watchdog(rank, world) if world > 1 else (None, {})
        comm = dist.init_from_env(ctx)
        fc["phase"] = "the runner's constructor (ncclCommSplit)"
        runner = dist.DistFitGrad2D(ctx, comm, spec, Xh, yh, noise, nb=None, cand=Ch, nsel=8)
        fc["phase"] = "the first all-gather behind the constructor"
        comm.barrier()
        if fc_timer is not None:
            fc_timer.cancel()
        layout = "%d ranks, 2-D block-cyclic %dx%d grid; gradient traces and MI scoring sharded" % (world, runner.geo.Pr, runner.geo.Pc)

        def step():
            out = runner.step()
            times.update(runner.times)
            return out
        barrier, reduce_max = comm.barrier, comm.max_float
    else:
        nslab = int(os.environ.get("GPX_C5_SLABS", "16"))   # measured 1 / 2 / 4 / 8 / 16 / 32 slabs: 7.8 / 5.7 / 4.4 / 3.7 / 3.5 / 3.5 s (fewer
        #                                                      flops with more slabs -- 2 N^3 down to 2 N^3 / 3 -- thinner products)
        linv_form = os.environ.get("GPX_LML_GRAD_FORM", "linv") == "linv" and dev.lml_grad_linv_fits(ctx, N)
        layout = "1 GPU (gradient traces: %s)" % ("L^-1 once, lower K^-1 = L^-T L^-1 as one triangular-operand product over it"
                                                  if linv_form else "%d row slabs of K^-1, two triangular solves each" % nslab)
        X = dev.points(ctx, Xh)
        Cp = dev.points(ctx, Ch)
        K = dev.DeviceMatrix.zeros(ctx, N, N)
        bounds = dev.lml_grad_slab_bounds(N, nslab)

        def step():
            t0 = time.perf_counter()
            dev.kfill_into(ctx, spec, X, K, nugget=noise)
            dev.potrf(ctx, K)
            alpha = dev.potrs(ctx, K, yh)
            ll = -0.5 * hdot(yh, alpha) - 0.5 * dev.logdet(ctx, K) - N / 2.0 * np.log(2 * np.pi)
            t1 = time.perf_counter()
            if linv_form:        # one explicit L^-1, lower K^-1 over it (gpx_lml_grad_linv: 2 N^2 + N^2/4 doubles of scratch)
                sums = dev.lml_grad_linv(ctx, spec, K, X, alpha)
            else:
                sums = np.zeros(d + 2)
                for r0, r1 in zip(bounds[:-1], bounds[1:]):
                    if r1 > r0:
                        sums += dev.lml_grad_slab(ctx, spec, K, X, alpha, r0, r1)
            grad = dev.lml_grad_from_sums(spec, sums)
            t2 = time.perf_counter()
            picks, _ = dev.mi_greedy(ctx, spec, Cp, noise, 8)
            t3 = time.perf_counter()
            times.update({"fit_ms": 1e3 * (t1 - t0), "lml_grad_ms": 1e3 * (t2 - t1), "mi_ms": 1e3 * (t3 - t2)})
            return ll, grad, picks

        def barrier():
            pass

        def reduce_max(v):
            return v
    import threading
    limit_s = float(os.environ.get("GPX_BENCH_WATCHDOG_S", "1500"))
    progress = {"at": "warm-up"}

    def _expired():
        print("bench.py: watchdog: rank %d of %d still in %s of --config c5 after %.0f s (%s) -- aborting"
              % (rank, world, progress["at"], limit_s, layout), file=sys.stderr, flush=True)
        os._exit(124)

    watchdog = threading.Timer(limit_s, _expired)
    watchdog.daemon = True
    watchdog.start()
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.sync()
    progress["at"] = "the timed steps"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ll, grad, picks = step()
    ctx.sync()
    barrier()
    dt = reduce_max(time.perf_counter() - t0)
    watchdog.cancel()
    if rank == 0:
        fit_flops = float(N) ** 3 / 3.0
        grad_flops = 2.0 * float(N) ** 3 / 3.0
        line = {"metric": "GP-fit + log-marginal gradient + MI design wall-time at N=%d d=%d (ms_per_step)" % (N, d),
                "value": N / (dt / args.steps), "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": "C5: N=%d d=%d ARD-SE (l_k=0.5+0.03k, s=1, noise=0.1): kfill+potrf+potrs+logdet+lml_grad "
                                       "(12 traces)+greedy MI (8 picks, M=%d candidates)" % (N, d, M),
                           "N": N, "d": d, "M_candidates": M, "kernel": "se-ard", "seed": N, "parallelism": layout},
                "phases_ms_last_step": times,
                "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel (factorisation N^3/3 + inverse slabs 2N^3/3)",
                             "achieved": (fit_flops + grad_flops) * args.steps / dt / 1e12, "peak": PEAK_FP64_MFMA_TFLOPS,
                             "unit": "TFLOP/s", "frac": (fit_flops + grad_flops) * args.steps / dt / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                             "traffic": None,
                             "note": "whole-step figure: algorithmic flops N^3 (fit + the two triangular solves of the inverse slabs) "
                                     "over the wall time, which also contains the HBM-bound trace kernel and the MI design"},
                "results": {"loglike": ll, "grad": [float(g) for g in grad], "mi_picks": [int(p) for p in picks]},
                "device": ctx.info()["name"]}
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)
    barrier()
    ctx.close()


def dist_preflight(ctx, comm, dist, dev, spec, d, nb=256, two_d=True, n=2048, m=512):
    """One small distributed step (N = 2048, M = 512, the bench's kernel) against the single-GPU path on the same inputs:
    log-likelihood and IVAR to 1e-10, and the factor block by block (2-D layout: every rank checks the blocks of the
    distributed factor it owns).  Returns a dict with `ok`; on a mismatch `first_bad_block` = (I, J) of the first nb-block of L
    that differs by more than 1e-10."""
    Xh, yh, Zh, noise = workload(n, d, m, seed=12345)
    t0 = time.perf_counter()
    if two_d and debug_inject() == "hang2d":      # (debug switch) a collective that never completes
        time.sleep(1e6)
    if two_d:
        run = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, noise, nb=nb)
    else:
        run = dist.DistFitIvar(ctx, comm, spec, Xh, yh, Zh, noise, nb=nb)
    ll, iv = run.step()
    X = dev.points(ctx, Xh)
    K1 = dev.potrf(ctx, dev.kfill(ctx, spec, X, nugget=noise))
    a1 = dev.potrs(ctx, K1, yh)
    ll1 = -0.5 * hdot(yh, a1) - 0.5 * dev.logdet(ctx, K1) - n / 2.0 * np.log(2 * np.pi)
    iv1 = abs(dev.ivar(ctx, spec, K1, X, dev.points(ctx, Zh)))
    out = {"N": n, "M": m, "nb": nb, "layout": "2d" if two_d else "1d", "loglike": ll, "ivar": iv,
           "rel_err_loglike": abs(ll - ll1) / abs(ll1) if (two_d or comm.rank == 0) else 0.0,
           "rel_err_ivar": abs(iv - iv1) / abs(iv1)}
    ok = out["rel_err_loglike"] <= 1e-10 and out["rel_err_ivar"] <= 1e-10
    if two_d and debug_inject() == "fail2d":      # (debug switch) exercise the fall-back to the 1-D layout
        ok = False
        out["forced_failure"] = True
    L1 = K1.to_host(tri=1)
    scale = float(np.max(np.abs(L1)))
    if two_d:
        # this rank's blocks of the DISTRIBUTED factor (the streamed evaluation keeps no replica): block (I, J) of L at local
        # block (I // Pr, J // Pc)
        geo, Al = run.geo, run.A.to_host()
        worst, first_bad = 0.0, None
        for J in range(geo.pc, geo.nblk, geo.Pc):
            for I in range(geo.pr, geo.nblk, geo.Pr):
                if I < J:
                    continue
                h, wj = min(geo.height(I), n - I * nb), min(geo.height(J), n - J * nb)
                if h <= 0 or wj <= 0:
                    continue
                loc = Al[(I // geo.Pr) * nb:(I // geo.Pr) * nb + h, (J // geo.Pc) * nb:(J // geo.Pc) * nb + wj]
                ref = L1[I * nb:I * nb + h, J * nb:J * nb + wj]
                if I == J:
                    loc = np.tril(loc)
                e = float(np.max(np.abs(loc - ref)))
                worst = max(worst, e) if e == e else float("inf")
                if not (e <= 1e-10 * scale) and first_bad is None:
                    first_bad = [I, J]
        out["max_err_L"] = worst / scale
        if first_bad is not None:
            ok = False
            out["first_bad_block"] = first_bad
    else:
        diff = np.abs(run.K.to_host(tri=1) - L1)
        out["max_err_L"] = float(diff.max() / scale)
        if not (diff.max() <= 1e-10 * scale):
            ok = False
            bad = np.argwhere(~(diff <= 1e-10 * scale))
            i, j = min(((int(a) // nb, int(b) // nb) for a, b in bad), key=lambda t: (t[1], t[0]))
            out["first_bad_block"] = [i, j]
    out["ok"] = bool(ok)
    out["seconds"] = time.perf_counter() - t0
    del run
    return out


def multi_gpu_replay(ctx, dev, spec, Xh, yh, Zh, noise, K, X, fit_ms, step_ms, grid=(2, 4)):
    """After the timed region, --gpus 1 only: the 8-GPU form of the step, measured on THIS GPU by PACED
    single-rank replays (scripts/dist_replay.py paced_grid).  One process plays a rank of the 2 x 4 grid -- that rank's exact
    recorded program of the 2-D block-cyclic factorisation (gpexp_amd.dist.dist2_potrf_enqueue), every receive a device copy of
    the same bytes out of the factor the timed steps left in K (scripts/replay_comm.py).  A replay in which every foreign panel
    arrives at once keeps the rank busy all the time and hides what bounds a real run: step k+1's panel solve needs step k's
    panel, so the factorisation time is the SUM over the steps of the holder column's latency (near update of its column, panel
    solve, hand-over).  So every process column is replayed in turn with each foreign panel held back by the latency measured
    for ITS holder column, and the sweep is iterated to a fixed point.  `paced_step_ms_max` is then the step time of the grid.
    xGMI IS NOT IN IT: transfers cost a device copy, so the figure is a lower bound on a real node's time."""
    scripts = os.path.join(ROOT, "scripts")
    if scripts not in sys.path:
        sys.path.insert(0, scripts)
    import dist_replay
    t0 = time.perf_counter()
    from gpexp_amd import dist as _d
    nb = _d.default_nb(len(Xh), grid[0] * grid[1], False)        # what the product's runners choose at this size and world
    nb_streamed = _d.default_nb(len(Xh), grid[0] * grid[1], True)
    dev.kfill_into(ctx, spec, X, K, nugget=noise)      # (the isolated fill launches above left an unfactored matrix in K)
    dev.potrf(ctx, K)
    ctx.sync()
    fit = dist_replay.paced_grid(ctx, spec, Xh, yh, Zh[:1024], noise, K, X, grid, nb=nb, streamed=False, iters=6, steps=2)
    both = dist_replay.paced_grid(ctx, spec, Xh, yh, Zh, noise, K, X, grid, nb=nb_streamed, streamed=True, iters=8, steps=2)
    if "error" in fit or "error" in both:
        return {"error": fit.get("error") or both.get("error")}
    # the OTHER schedule of the step: factorisation, then each rank evaluates its M / 8 slice against its replica (the product
    # default below N = 98304): the slice's solve measured alone + alpha / log det from the replica by the single-GPU sweeps
    W = grid[0] * grid[1]
    Zs = dev.points(ctx, Zh[:max(len(Zh) // W, 1)])
    y_dev, a_dev = dev.padded_vector(ctx, yh), dev.padded_vector(ctx, np.zeros(len(yh)))
    ts = []
    for _ in range(3):
        ctx.sync()
        t1 = time.perf_counter()
        dev.posterior(ctx, spec, K, X, None, Zs, want_mean=False)
        dev.potrs_dev(ctx, K, y_dev, a_dev)
        dev.logdet(ctx, K)
        ctx.sync()
        ts.append(1e3 * (time.perf_counter() - t1))
    slice_ms = min(ts)

    def brief(r):
        return {"panels_per_trailing_update": r["agg"], "replayed_ranks": r["replayed_ranks"],
                "unpaced_rank_busy_ms": {k: v for k, v in r["iterations"][0]["rank_step_ms"].items()},
                "chain_ms": r["chain_ms"], "chain_ms_by_process_column": r["chain_ms_by_process_column"],
                "chain_ms_per_iteration": [h["chain_ms"] for h in r["iterations"]],
                "paced_step_ms": r["paced_step_ms"], "paced_step_ms_max": r["paced_step_ms_max"],
                "diagonal_chain_paced": r.get("diagonal_chain_paced"),
                # what the stand-in device copies of the receives cost the slowest rank's communication stream (in the paced step)
                "standin_copy_ms_max": max([v for v in (r.get("foreign_excess_ms") or {}).values() if v is not None] or [None],
                                           key=lambda v: -1.0 if v is None else v),
                "holder_latency_ms_first_mid_last": r["holder_latency_ms_first_mid_last"],
                "bytes_received_per_step": r["bytes_received_per_fit"], "variance_check_rel": r["variance_check_rel"]}
    out = {"grid": "%dx%d" % grid, "nb": nb, "nb_streamed": nb_streamed,
           "method": "paced single-rank replays on one GPU, every rank of the grid in turn, foreign panels held back by the "
                     "measured latency of their holder column, iterated; see scripts/dist_replay.py paced_grid",
           "fit_only": brief(fit), "fit_ivar_streamed": brief(both),
           "ivar_slice_after_fit_ms": slice_ms,
           "fit_then_ivar_ms": fit["paced_step_ms_max"] + slice_ms,
           "single_gpu_fit_ms": fit_ms, "single_gpu_step_ms": step_ms,
           "ratio_fit": (fit_ms / fit["paced_step_ms_max"]) if fit_ms else None,
           "ratio_step_fit_then_ivar": (step_ms / (fit["paced_step_ms_max"] + slice_ms)) if step_ms else None,
           "ratio_step_streamed": (step_ms / both["paced_step_ms_max"]) if step_ms else None,
           "schedule_note": "fit_then_ivar = the product default at this size (evaluation after the fit against the rank's replica: "
                            "paced factorisation + the slice's solve, alpha and log det measured alone); fit_ivar_streamed = the "
                            "evaluation streamed underneath the factorisation against a window of the factor (default from N = 98304)",
           "xgmi": "NOT INCLUDED -- receives are device copies of the same bytes, sends cost nothing; the ratios are single-GPU time / "
                   "paced step time of the grid: upper bounds on what 8 GPUs can reach, not a measured scaling figure"}
    # Round 5: the class API's distributed-factor mode on the same grid -- rank 0 replayed with NO replica of the factor
    # (block-cyclic local matrix + ring of packed buffers only), its M / 8 slice of the evaluation solved against a window while
    # the panels are re-streamed (DistFitIvar2D.cyclic_posterior; receives = stand-in device copies), checked against this GPU's
    # own posterior; and what a rank holds for the factor in either mode
    try:
        os.environ["GPX_REPLAY_CYCLIC"] = "1"
        r = dist_replay.replay_rank(ctx, spec, Xh, yh, Zh, noise, K, X, grid, 0, nb=nb, streamed=False, steps=1, profile=False)
        df = r["distributed_factor"]
        out["distributed_factor"] = dict(
            df, rank=0, unpaced_fit_busy_ms=r["ms_per_step"],
            resident_factor_bytes_replica_mode=df["resident_factor_bytes"] - df["window_bytes"] + df["replica_would_be_bytes"],
            note="resident_factor_bytes = local share of the matrix + ring of packed panel buffers + early buffers + window; the "
                 "replica mode (default below N = 65536) adds one N x N replica per rank (round 4: two, the GP held a clone)")
    except Exception as e:      # the measurement above stands without it
        out["distributed_factor"] = {"error": repr(e)}
    finally:
        os.environ.pop("GPX_REPLAY_CYCLIC", None)
    out["seconds"] = time.perf_counter() - t0
    return out


PREFLIGHT_HANG = 125     # exit code of a rank whose preflight step never finished (its own watchdog)


def first_contact_watchdog(rank, world):
    """Armed in the child BEFORE its first RCCL call: ncclCommInitRank + the barrier all-gather, the two blocking
    ncclCommSplit of the 2-D runner's constructor and the layout agreement are the likeliest places for a first-contact hang,
    and they all come before the preflight step.  On expiry the process leaves with PREFLIGHT_HANG, so supervise() starts the
    second attempt on the 1-D layout instead of the run sitting until the launcher's limit with nothing on stderr.  The caller
    updates `state["phase"]` as it goes and cancels the timer behind the post-preflight all-gather."""
    import threading
    limit = float(os.environ.get("GPX_BENCH_PREFLIGHT_WATCHDOG_S", "240"))
    state = {"phase": "ncclCommInitRank / first barrier", "layout": os.environ.get("GPX_DIST_LAYOUT", "2d")}

    def expired():
        print("bench.py: watchdog: rank %d of %d: FIRST CONTACT did not finish within %.0f s (stuck in: %s; layout %s) -- a "
              "collective of the distributed path never completed; aborting (try GPX_DIST_LAYOUT=1d)"
              % (rank, world, limit, state["phase"], state["layout"]), file=sys.stderr, flush=True)
        os._exit(PREFLIGHT_HANG)

    t = threading.Timer(limit, expired)
    t.daemon = True
    t.start()
    return t, state


def debug_inject():
    """ONE debug switch for the failure injections the first-contact tests need (GPX_BENCH_INJECT=hang2d | fail2d); nothing
    else in this file changes behaviour for tests."""
    return os.environ.get("GPX_BENCH_INJECT", "")


def supervise(world, rank):
    """N > 1: the work runs in a CHILD process of this one.  This process never touches the GPU, so it may start programs; it
    exists for one case -- RCCL with more than one rank has never run on the build's hardware, and a collective that never
    completes can only be left by ending the process.  When the child ends at the PREFLIGHT watchdog on the 2-D layout (every
    rank's child does: they hang in the same collective, and a rank that got through waits for the others in the all-gather
    behind it), every rank's supervisor starts a second child on the simpler 1-D block-column layout (one ncclBroadcast per
    panel) with a fresh rendezvous, instead of reporting nothing.  Any other outcome is passed through."""
    import subprocess
    key = "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
    attempts = ["2d", "1d"] if os.environ.get("GPX_DIST_LAYOUT", "2d") != "1d" else ["1d"]
    code = 1
    for i, layout in enumerate(attempts):
        env = dict(os.environ, GPX_BENCH_CHILD="1", GPX_RDV_KEY="%s_try%d" % (key, i), GPX_DIST_LAYOUT=layout)
        if i > 0 and env.get("MASTER_PORT", "").isdigit():
            # (only the host-staged TEST communicator rendezvouses through torch.distributed: a store of its own, on the next
            # port, instead of the launcher's, which still holds the first attempt's keys)
            env["MASTER_PORT"] = str(int(env["MASTER_PORT"]) + 1)
            env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
        code = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env).returncode
        if code != PREFLIGHT_HANG or i + 1 == len(attempts):
            break
        print("bench.py: supervisor of rank %d: the %s preflight never finished -- second attempt on the 1-D block-column layout"
              % (rank, layout), file=sys.stderr, flush=True)
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--train-points", dest="n", type=int, default=None, help="default: 32768 (c4) / 65536 (c5)")
    ap.add_argument("--dim", dest="d", type=int, default=None, help="default: 8 (c4) / 10 (c5)")
    ap.add_argument("--mc-points", dest="m", type=int, default=None, help="default: 32768 (c4); c5: MI candidates, 8192")
    ap.add_argument("--kernel", choices=["matern52", "matern32"], default="matern52",
                    help="matern52 = BASELINE config C4; matern32 = the only Matern the reference itself can evaluate "
                         "(kernels.py:85-89), same workload")
    ap.add_argument("--config", choices=["c4", "c5"], default="c4",
                    help="c4 = the headline workload (default); c5 = BASELINE config 5: N=65536, d=10 ARD-SE, fit + log-marginal "
                         "gradient + greedy MI design (8 picks over 8192 candidates), 1..N GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-replay", action="store_true", help="skip the multi_gpu_replay object (rank 0 of a 2x4 grid replayed on this GPU)")
    ap.add_argument("--cpu-baseline", choices=["bounded", "full"], default="bounded",
                    help="full: SURVEY.md 8d's whole protocol (adds N=8192 and the fair-CPU Cholesky at N=32768; minutes)")
    args = ap.parse_args()

    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("GPX_BENCH_CHILD") != "1"
            and os.environ.get("GPX_BENCH_SUPERVISE", "1") == "1"):
        sys.exit(supervise(int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0"))))

    # The contract is ONE JSON line on stdout, and native libraries write there too (RCCL's version banner at init and at
    # ncclCommSplit, gloo's connection lines in the host-staged rehearsal): fd 1 points at stderr until the line is printed.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    from gpexp_amd import device as dev
    # GPX_FORCE_DEVICE: rehearsal of the multi-rank flow on a box with fewer GPUs than ranks (with GPX_COMM=host)
    ctx = dev.Context(int(os.environ.get("GPX_FORCE_DEVICE", local_rank)))
    dev._ctx = ctx
    info = ctx.info()

    if args.config == "c5":
        bench_c5(args, ctx, dev, world, rank, stdout_fd)
        return
    N, d, M = (32768 if args.n is None else args.n), (8 if args.d is None else args.d), (32768 if args.m is None else args.m)
    Xh, yh, Zh, noise = workload(N, d, M, seed=N)
    spec = dev.KernelSpec(dev.K_MATERN52 if args.kernel == "matern52" else dev.K_MATERN32, d, [0.5, 1.0])
    preflight = None

    if world > 1 or os.environ.get("GPX_FORCE_DIST") == "1":  # GPX_FORCE_DIST: rehearse the RCCL runner on one GPU
        from gpexp_amd import dist
        fc_timer, fc = (first_contact_watchdog(rank, world) if (world > 1 or os.environ.get("GPX_BENCH_PREFLIGHT") == "1")
                        else (None, {}))
        comm = dist.init_from_env(ctx)
        fc["phase"] = "the runner's constructor (ncclCommSplit)"
        # default: north_star's 2-D block-cyclic layout (Pr x Pc grid, gpexp_amd/dist.py); GPX_DIST_LAYOUT=1d selects the
        # round-1 block-column layout (every rank holds the full matrix, one ncclBroadcast per panel)
        want_2d = os.environ.get("GPX_DIST_LAYOUT", "2d") != "1d"
        nb1d = int(os.environ.get("GPX_DIST_NB", "512"))       # block-column width of the 1-D fallback layout
        nb = dist.default_nb(N, world, False) if want_2d else nb1d
        runner, err = None, ""
        if want_2d:
            try:
                runner = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, noise, nb=nb)
            except Exception as exc:   # ncclCommSplit missing / failed, out of memory, ...
                err = "%s: %s" % (type(exc).__name__, exc)
        # The layout is agreed on COLLECTIVELY: a rank that could not build the 2-D runner (while others could) would
        # otherwise issue the 1-D path's collectives against its peers' 2-D ones and hang until the watchdog.
        fc["phase"] = "the layout agreement all-gather"
        ok_all = comm.allgather(np.array([1.0 if (runner is not None or not want_2d) else 0.0]))[:, 0]
        if want_2d and ok_all.min() < 1.0:
            print("bench.py: rank %d: 2-D layout unavailable on rank(s) %s%s; ALL ranks fall back to the 1-D block-column layout"
                  % (rank, [i for i, v in enumerate(ok_all) if v < 1.0], (" (here: %s)" % err) if err else ""),
                  file=sys.stderr, flush=True)
            runner = None
            want_2d = False
        if runner is None:
            runner = dist.DistFitIvar(ctx, comm, spec, Xh, yh, Zh, noise, nb=nb1d)
            layout = "1-D block-cyclic columns" + ("" if os.environ.get("GPX_DIST_LAYOUT") == "1d" else " (2-D setup failed)")
        else:
            layout = "2-D block-cyclic %dx%d grid, nb=%d, %d panels per trailing update" % (runner.geo.Pr, runner.geo.Pc, nb, runner.agg)
        # First contact (RCCL with more than one rank has never run on the build's hardware): one SMALL step of the same
        # runner class, checked on every rank against the single-GPU path before anything is timed.  A mismatch names the
        # first block of the replicated factor that differs; the bench then stops instead of timing garbage.
        if world > 1 or os.environ.get("GPX_BENCH_PREFLIGHT") == "1":
            # (the first-contact watchdog armed before ncclCommInitRank is still running: a collective that never completes
            # here costs minutes, not the timed region's whole allowance, and the message says it was the preflight)
            fc["phase"] = "the PREFLIGHT step (N = 2048)"
            fc["layout"] = "2d" if want_2d else "1d"
            preflight = dist_preflight(ctx, comm, dist, dev, spec, d, nb=min(nb, 256), two_d=want_2d)
            print("bench.py: preflight rank %d: %s" % (rank, json.dumps(preflight)), file=sys.stderr, flush=True)
            bad = comm.allgather(np.array([0.0 if preflight["ok"] else 1.0]))[:, 0]
            if bad.max() > 0 and want_2d:
                # a WRONG (not hanging) 2-D step: all ranks agree (the all-gather above) to try the simpler 1-D block-column
                # layout -- one ncclBroadcast per panel, every rank holds the matrix -- rather than to report nothing
                print("bench.py: rank %d: 2-D preflight FAILED on rank(s) %s; ALL ranks fall back to the 1-D block-column layout"
                      % (rank, [i for i, v in enumerate(bad) if v > 0]), file=sys.stderr, flush=True)
                first = preflight
                del runner
                want_2d = False
                runner = dist.DistFitIvar(ctx, comm, spec, Xh, yh, Zh, noise, nb=nb1d)
                layout = "1-D block-cyclic columns (the 2-D preflight failed)"
                preflight = dist_preflight(ctx, comm, dist, dev, spec, d, nb=min(nb, 256), two_d=False)
                preflight["failed_2d_preflight"] = first
                print("bench.py: preflight (1-D) rank %d: %s" % (rank, json.dumps(preflight)), file=sys.stderr, flush=True)
                bad = comm.allgather(np.array([0.0 if preflight["ok"] else 1.0]))[:, 0]
            fc_timer.cancel()
            if bad.max() > 0:
                print("bench.py: preflight FAILED on rank(s) %s -- not timing a wrong result" % [i for i, v in enumerate(bad) if v > 0],
                      file=sys.stderr, flush=True)
                comm.barrier()
                os._exit(3)
        step = runner.step
        barrier = comm.barrier
        reduce_max = comm.max_float
    else:
        layout = "1 GPU"
        X = dev.points(ctx, Xh)
        Z = dev.points(ctx, Zh)
        K = dev.DeviceMatrix.zeros(ctx, N, N)   # allocated once; refilled in place every step
        y_dev = dev.padded_vector(ctx, yh)
        alpha_dev = dev.padded_vector(ctx, np.zeros(N))

        def step():
            dev.kfill_into(ctx, spec, X, K, nugget=noise)
            dev.potrf(ctx, K)
            # alpha = K^-1 y: a chain of 64 dependent launches -- on the high-priority side stream, underneath the IVAR
            # GEMMs (both only read L); the results are collected after the device-wide sync
            ctx.stream(1)
            dev.potrs_dev(ctx, K, y_dev, alpha_dev)
            ctx.stream(0)
            iv = abs(dev.ivar(ctx, spec, K, X, Z))
            logdet = dev.logdet(ctx, K)
            ctx.sync()
            alpha = alpha_dev.to_host()[:N, 0]
            ll = -0.5 * hdot(yh, alpha) - 0.5 * logdet - N / 2.0 * np.log(2 * np.pi)   # (no BLAS: _lib.hdot)
            return ll, iv

        def barrier():
            pass

        def reduce_max(v):
            return v

    def sync():
        # device-wide synchronisation of everything this process enqueued.  All work of the hot path is issued on the
        # library's own HIP stream, so this is the equivalent of torch.cuda.synchronize(); torch itself is kept out
        # of the process because a second HIP/HSA runtime next to the system RCCL breaks ncclCommInitRank.
        ctx.sync()

    # A collective that never completes (the multi-rank RCCL path has only been rehearsed through a host-staged communicator)
    # would otherwise hang until the launcher's own limit with nothing on stderr: say where, and leave with an error so that
    # torch.distributed.run tears the other ranks down.
    import threading
    limit_s = float(os.environ.get("GPX_BENCH_WATCHDOG_S", "1500"))
    progress = {"at": "warm-up"}

    def _expired():
        print("bench.py: watchdog: rank %d of %d still in %s after %.0f s (layout: %s) -- aborting"
              % (rank, world, progress["at"], limit_s, layout), file=sys.stderr, flush=True)
        os._exit(124)

    watchdog = threading.Timer(limit_s, _expired)
    watchdog.daemon = True
    watchdog.start()

    for _ in range(args.warmup):
        out = step()
    barrier()
    sync()
    progress["at"] = "the timed steps"
    # The timed region carries NO profiler events: the library's per-class HIP-event spans cost 13-15 ms per step at C4
    # (707 -> 694 ms in one call; ~1900 launches, the look-ahead factorisation alternates kernel classes on three streams).
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    barrier()
    dt = time.perf_counter() - t0
    dt = reduce_max(dt)
    ll, iv = out
    # ... they are recorded in an identical repeat right behind it (same steps, same inputs, same results): per-class spans,
    # launch counts and algorithmic flops for the roofline object and the rocprofv3 cross-check
    progress["at"] = "the instrumented repeat"
    if world > 1 or os.environ.get("GPX_FORCE_DIST") == "1":
        runner.force_interpret = True     # profiler events cannot live inside a captured graph: issue this repeat row by row
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out2 = step()
    sync()
    barrier()
    dt_instr = reduce_max(time.perf_counter() - t0)
    prof = ctx.profile_get()
    ctx.profile(False)
    if world == 1:
        assert out2 == out, "the instrumented repeat must reproduce the timed steps bit for bit"
    else:
        # several ranks: the sums inside ncclReduce / ncclAllReduce are not promised to associate the same way twice; a
        # mismatch beyond round-off is still an error, but it must not kill a timing run over the last bit
        assert all(abs(a - b) <= 1e-12 * abs(b) for a, b in zip(out2, out)), (out2, out)
    phases = {k: v["ms"] / args.steps for k, v in prof.items() if v["launches"]}
    watchdog.cancel()

    # GP-fit and IVAR-eval separately (N / t_fit and M / t_IVAR), outside the timed region: inside it the
    # alpha sweeps run underneath the IVAR GEMMs, so the two phases overlap and do not add up to ms_per_step
    fit_ms = ivar_ms = None
    if world == 1 and os.environ.get("GPX_FORCE_DIST") != "1":
        def fit_only():
            dev.kfill_into(ctx, spec, X, K, nugget=noise)
            dev.potrf(ctx, K)
            dev.potrs_dev(ctx, K, y_dev, alpha_dev)
            dev.logdet(ctx, K)

        def ivar_only():
            dev.ivar(ctx, spec, K, X, Z)

        res = []
        for fn in (fit_only, ivar_only):
            fn()
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            sync()
            res.append(1e3 * (time.perf_counter() - t1) / args.steps)
        fit_ms, ivar_ms = res

        # The two assembly kernels by themselves: one launch between two host synchronisations, host clock.  That is the
        # kernel's duration + ~0.03 ms of launch and completion latency (scripts/probe_event_vs_kernel.py: 1.825 ms against
        # 1.797 ms in rocprofv3's trace of the same launch).  The HIP-event spans of the passes above are NOT used for these two:
        # behind a host sync they open up to 0.5 ms before the kernel starts (scripts/probe_event_vs_kernel2.py: spans of
        # 1.6 -> 2.1 ms over twelve fits whose fill kernels all took 1.57-1.61 ms), and back-to-back launches slow the
        # rectangular fill itself down (1.65 -> 1.95 ms, rocprofv3), which is not how it occurs in the path.
        KX = K if M == N else dev.DeviceMatrix.zeros(ctx, N, M)
        for cls, fill, nbytes in (("kfill", lambda: dev.kfill_into(ctx, spec, X, K, nugget=noise), 8.0 * N * N + 8.0 * N * d),
                                  ("kcross", lambda: dev.kfill_into(ctx, spec, X, KX, Z=Z), 8.0 * N * M + 8.0 * (N + M) * d)):
            ts = []
            for _ in range(5):
                sync()
                time.sleep(0.003)  # a fill right behind another fill runs slower (the rectangular one by up to 20 %,
                t1 = time.perf_counter()  # rocprofv3); in the path it follows the factorisation / a host sync
                fill()
                sync()
                ts.append(1e3 * (time.perf_counter() - t1))
            prof[cls] = {"launches": 1, "ms": sorted(ts)[len(ts) // 2], "flops": 0.0, "bytes": nbytes}
        del KX

    if rank == 0:
        ms = 1e3 * dt / args.steps
        g = prof["gemm"]
        kf = prof["kfill"]
        kc = prof["kcross"]

        def hbm(p):
            gbs = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
            return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                    "avg_launch_ms": p["ms"] / p["launches"] if p["launches"] else 0.0,
                    "algorithmic_bytes_per_launch": p["bytes"] / p["launches"] if p["launches"] else 0.0}
        # The look-ahead factorisation runs GEMMs on three streams at once, so per-class event spans overlap and their sum
        # can exceed the wall time: the roofline divides the class's algorithmic flops by the WALL time of the timed region
        # (which also contains the ~1 % of assembly / reduction kernels) -- overlap cannot inflate it.
        # ALGORITHMIC flops of a step: N^3/3 (factorisation) + N^2 M (evaluation solve), SURVEY.md 8d.  The launched count of
        # the GEMM class (g["flops"]: adds the block-inverse builds and the padding of the triangular products, ~0.6 %) is
        # reported beside it, never used for `achieved`.
        algo_flops = float(N) ** 3 / 3.0 + float(N) ** 2 * float(M)
        ach = algo_flops * args.steps / dt / 1e12 if dt > 0 else 0.0
        traffic, traffic_src = None, None
        try:  # PMC counters cannot be read from inside the process: take the committed rocprofv3 --pmc passes of this
            # same command (profiles/), per launch like `achieved`; null when the profile is for another config
            pmc = [f for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
                   if os.path.exists(os.path.join(ROOT, "profiles", f))][0]
            with open(os.path.join(ROOT, "profiles", pmc)) as f:
                pj = json.load(f)
            if world == 1 and (N, d, M) == (32768, 8, 32768) and args.kernel == "matern52":
                traffic = (pj["fetch_bytes_per_step_corrected"] + pj["write_bytes_per_step"]) / pj["launches_per_step"]
                traffic_src = pj["source"]
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "GP-fit+IVAR-eval points/s at N=%d d=%d (wall-time in ms_per_step); value = (N+M)/t_step, SURVEY 8d's "
                      "N/t_fit and M/t_IVAR are points_per_s_fit / points_per_s_ivar right behind it" % (N, d),
            "value": (N + M) / (dt / args.steps),
            "unit": "points/s",
            "points_per_s_fit": (N / (fit_ms * 1e-3)) if fit_ms else None,
            "points_per_s_ivar": (M / (ivar_ms * 1e-3)) if ivar_ms else None,
            "fit_ms": fit_ms, "ivar_ms": ivar_ms,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C4: N=%d d=%d Matern-%s (rho=0.5,s=1,noise=0.1) kfill+potrf+potrs+logdet+"
                                   "IVAR over M=%d MC points" % (N, d, "5/2" if args.kernel == "matern52" else "3/2", M),
                       "N": N, "d": d, "M": M, "kernel": args.kernel, "seed": N,
                       "parallelism": layout if world == 1 else "%d ranks, %s" % (world, layout)},
            "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel (SYRK/TRSM updates)", "achieved": ach,
                         "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (mean over the step's launches)",
                         "traffic_source": traffic_src,
                         "traffic_source_run": "builder" if traffic is not None else None,
                         "algorithmic_flop_per_step": algo_flops,
                         "algorithmic_flop_per_launch": (algo_flops * args.steps / g["launches"]) if g["launches"] else 0.0,
                         "launched_flop_per_step": g["flops"] / args.steps,
                         "avg_launch_ms": (g["ms"] / g["launches"]) if g["launches"] else 0.0,
                         "launches_per_step": g["launches"] / args.steps,
                         "event_ms_per_step_summed_over_streams": g["ms"] / args.steps,
                         "ms_per_step_instrumented": 1e3 * dt_instr / args.steps,
                         "note": "achieved = algorithmic flops (N^3/3 + N^2 M) / wall time of the timed region (no profiler events inside it); the "
                                 "per-launch figures are HIP-event spans from an identical instrumented repeat of the same "
                                 "steps (bit-identical results, asserted); their sum counts time on concurrent streams twice "
                                 "(look-ahead) and is reported for the rocprofv3 cross-check only"},
            "roofline_kfill": dict(hbm(kf), kernel="kfill_kernel<SYM> (symmetric N x N assembly, mirror-written)",
                                   traffic=None),
            "roofline_kcross": dict(hbm(kc), kernel="kfill_rectn_kernel (rectangular N x M cross matrix, every element "
                                                   "computed)", traffic=None),
            "phase_note": "fit_ms (kfill + potrf + potrs + logdet) and ivar_ms are timed separately after the timed region "
                          "(inside it the alpha sweeps run underneath the IVAR GEMMs); roofline_kfill / roofline_kcross are "
                          "one isolated launch each between two host syncs, host clock, median of 5 (kernel duration + "
                          "~0.03 ms)",
            "phases_ms_per_step": phases,
            "phases_note": "HIP-event spans per kernel class in the instrumented repeat; trsv and reduce run on a side stream UNDERNEATH the IVAR GEMMs, "
                           "so their spans include waiting and the classes do not add up to ms_per_step",
            "results": {"loglike": ll, "ivar": iv},
            "device": info["name"],
        }
        if preflight is not None:
            line["preflight"] = preflight
        if world > 1 or os.environ.get("GPX_FORCE_DIST") == "1":
            line["host_issue_ms_per_step"] = dict(getattr(runner, "host_ms", {}) or {})
            line["comm_note"] = ("phases_ms_per_step.comm = HIP-event spans around the collectives on the communication stream "
                                 "(includes waiting for the peers); gemm / leaf = the compute strands")
        if world == 1 and not args.no_replay and os.environ.get("GPX_FORCE_DIST") != "1" and N >= 8192:
            line["multi_gpu_replay"] = multi_gpu_replay(ctx, dev, spec, Xh, yh, Zh, noise, K, X, fit_ms, ms)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(d, full=(args.cpu_baseline == "full"), kind=args.kernel)
            ref = os.path.join(ROOT, "profiles", "r02_bench_n1_cpu_full.json")
            if os.path.exists(ref):   # the WHOLE SURVEY 8d protocol takes ~5 minutes of CPU: a builder-run record, quoted here
                try:
                    with open(ref) as f:
                        full = json.load(f)["cpu_baseline"]
                    line["cpu_baseline"]["full_protocol_ref"] = {
                        "provenance": "profiles/r02_bench_n1_cpu_full.json: `bench.py --cpu-baseline full`, run by the builder on "
                                      "a gpurun MI355X box in round 2 (not re-measured by this run)",
                        "cpu_model": full.get("cpu_model"), "blas_threads": full.get("blas_threads"),
                        "reference_algorithm_runs": [{k: r[k] for k in ("N", "M", "total_s")} for r in full.get("runs", [])],
                        "fit": full.get("fit"), "extrapolated": full.get("extrapolated"),
                        "fair_cpu_chol_full_size": {k: full["fair_cpu_chol"][k] for k in ("N", "M", "fit_s", "ivar_s", "total_s")}}
                except (OSError, KeyError, ValueError):
                    pass
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)
    barrier()
    ctx.close()


if __name__ == "__main__":
    main()
