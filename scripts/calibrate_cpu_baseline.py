#!/usr/bin/env python3
"""SURVEY.md 8d calibration gate: "in this container the build's ref-exact mode must reproduce the reference's timings within
+-20 % before its GPU-box timings are trusted".

Runs in the BUILD container only (it imports the live reference from /root/reference, which does not travel to the GPU box):
times, on the same inputs and the same cores,
    the reference itself   gpExp.gp.GP.train (row-loop assembly + numpy.linalg.pinv, gp.py:76-101, 156-181) and
                           GP.evaluateVariance (per-point loop, gp.py:213-259)
    the oracle's port      bench.py's `_ref_exact` pieces: oracle.cov_matrix(row_loop=True) + numpy.linalg.pinv, and
                           oracle.posterior (the restated variance loop)
with the one Matern the reference can evaluate (nu = 3/2, kernels.py:85-89), and writes profiles/r03_cpu_calibration.json with
the ratios and the verdict of the gate.  bench.py's cpu_baseline leg times exactly the oracle pieces measured here.

    python scripts/calibrate_cpu_baseline.py [--sizes 2048,4096] [--m 256]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2048,4096")
    ap.add_argument("--m", type=int, default=256)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_cpu_calibration.json"))
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("calibrate_cpu_baseline.py needs the live reference at %s (build container only)" % REF)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    os.environ.setdefault("MPLBACKEND", "Agg")
    from gpExp.kernels import KernelIsoMatern          # the live reference
    from gpExp.gp import GP as RefGP
    from oracle import gpexp_oracle as orc
    import bench

    d, noise = 8, 0.1
    rows = []
    # warm-up of both sides (BLAS thread pool, first-call imports): untimed
    Xw, yw, Zw, _ = bench.workload(256, d, 16, seed=1)
    gw = RefGP(KernelIsoMatern(0.5, 1.0, d), noise)
    gw.train(Xw, yw)
    gw.evaluateVariance(Zw, parallel=0)
    Kw = orc.cov_matrix(dict(kind="matern32", rho=0.5, signalSize=1.0, d=d), Xw, noise, row_loop=True)
    np.linalg.pinv(Kw)
    for n in [int(v) for v in args.sizes.split(",")]:
        X, y, Z, _ = bench.workload(n, d, args.m, seed=n)
        spec = dict(kind="matern32", rho=0.5, signalSize=1.0, d=d)
        # --- reference
        k = KernelIsoMatern(0.5, 1.0, d)
        g = RefGP(k, noise)
        t0 = time.perf_counter()
        g.train(X, y)
        t_ref_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        v_ref = g.evaluateVariance(Z, parallel=0)
        t_ref_var = time.perf_counter() - t0
        # --- oracle (what bench.py's cpu_baseline times)
        t0 = time.perf_counter()
        K = orc.cov_matrix(spec, X, noise, row_loop=True)
        P = np.linalg.pinv(K)
        coeff = P @ y
        t_orc_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, v_orc = orc.posterior(spec, dict(K=K, P=P, X=X), Z)
        t_orc_var = time.perf_counter() - t0
        rows.append(dict(N=n, M=args.m, ref_fit_s=t_ref_fit, oracle_fit_s=t_orc_fit, fit_ratio=t_orc_fit / t_ref_fit,
                         ref_var_s=t_ref_var, oracle_var_s=t_orc_var, var_ratio=t_orc_var / t_ref_var,
                         total_ratio=(t_orc_fit + t_orc_var) / (t_ref_fit + t_ref_var),
                         max_abs_diff_coeff=float(np.max(np.abs(coeff - g.coeff))),
                         max_abs_diff_var=float(np.max(np.abs(np.ravel(v_orc) - np.ravel(v_ref))))))
        print(json.dumps(rows[-1]), flush=True)
    ok = all(0.8 <= r["total_ratio"] <= 1.2 for r in rows)
    out = dict(gate="oracle ref-exact time within +-20 % of the live reference (SURVEY.md 8d)", passed=bool(ok),
               kernel="matern32 (the one Matern the reference evaluates)", d=d, noise=noise, cores=os.cpu_count(), runs=rows,
               where="build container (8 CPUs); the reference does not travel to the GPU box")
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("calibration gate:", "PASSED" if ok else "FAILED", "->", args.out)


if __name__ == "__main__":
    main()
