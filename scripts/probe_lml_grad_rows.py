"""One slab of the sharded log-marginal gradient in its two forms -- row slab of K^-1 by two solves (gpx_lml_grad_slab) and
rows of L^-1 by one solve + one SYRK (gpx_lml_grad_rows) -- per slab of a partition of equal work, with the GEMM class's
profile (launches, ms, TF/s on the launched flops).  Usage: probe_lml_grad_rows.py [N] [parts]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpexp_amd import device as dev
ctx = dev.context()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = 10
rng = np.random.default_rng(N)
Xh = rng.uniform(-1, 1, (N, d)); y = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(N)
sp = dev.KernelSpec(dev.K_SE, d, [0.5 + 0.03 * k for k in range(d)] + [1.0])
X = dev.points(ctx, Xh)
L = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.1))
alpha = dev.potrs(ctx, L, y)
nsub = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for form, bounds, piece in (("slab", dev.lml_grad_slab_bounds, dev.lml_grad_slab), ("rows", dev.lml_grad_rows_bounds, dev.lml_grad_rows),
                            ("rows x%d" % nsub, dev.lml_grad_rows_bounds, lambda *a: dev.lml_grad_rows(*a, nsub=nsub))):
    b = bounds(N, parts)
    tot = np.zeros(d + 2); tsum = 0.0
    for i in range(parts):
        if b[i + 1] <= b[i]:
            continue
        piece(ctx, sp, L, X, alpha, b[i], b[i + 1])          # warm (pool blocks, block inverses)
        ctx.sync(); ctx.profile(True); ctx.profile_reset(); t0 = time.perf_counter()
        s = piece(ctx, sp, L, X, alpha, b[i], b[i + 1]); ctx.sync(); t = time.perf_counter() - t0
        p = ctx.profile_get(); ctx.profile(False)
        g = p["gemm"]
        work = (2.0 * (N - b[i]) ** 2 if form == "slab" else 2.0 * b[i + 1] ** 2) * (b[i + 1] - b[i])
        if form.startswith("rows x"):       # the staircase of the sub-slabs: what the range's own flops are
            work = 2.0 * (b[i + 1] ** 3 - b[i] ** 3) / 3.0 * (1.0 + 0.5 / nsub)
        print("%s [%6d, %6d): %7.1f ms = %5.1f TF/s on the slab's 2 s r^2 flops | gemm class: %3d launches %7.1f ms, launched %.2e flop | reduce %.1f ms"
              % (form, b[i], b[i + 1], 1e3 * t, work / t / 1e12, g["launches"], g["ms"], g["flops"], p.get("reduce", {}).get("ms", 0.0)), flush=True)
        tot += s; tsum += t
    print("%s total %.1f ms = %.1f TF/s on 2 N^3 / 3; gradient %s" % (form, 1e3 * tsum, 2.0 * N ** 3 / 3 / tsum / 1e12,
                                                                     np.array2string(dev.lml_grad_from_sums(sp, tot)[:3], precision=10)), flush=True)
