"""FITC at scale: N training points, a tenth of them inducing; timing of fit / solve / posterior."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rng = np.random.default_rng(N)
d = 8
Xh = rng.uniform(-1, 1, (N, d))
y = np.sin(2 * np.pi * Xh.sum(1) / d) + np.sqrt(0.1) * rng.standard_normal(N)
S = Xh[rng.permutation(N)[:N // 10]].copy()
Zh = rng.uniform(-1, 1, (8192, d))
sp = dev.KernelSpec(dev.K_SE, d, list(0.4 + 0.05 * np.arange(d)) + [1.0])
X, Sd, Z = dev.points(ctx, Xh), dev.points(ctx, S), dev.points(ctx, Zh)
for it in range(2):
    ctx.sync(); t0 = time.perf_counter()
    m = dev.FitcModel(ctx, sp, X, Sd, 0.1); ctx.sync(); t1 = time.perf_counter()
    coeff, quad = m.solve(y); t2 = time.perf_counter()
    ld = m.logdet(); t3 = time.perf_counter()
    mean, var = m.posterior(coeff, Z); t4 = time.perf_counter()
print("FITC N=%d nu=%d: fit %.1f ms, solve %.1f ms, logdet %.2f ms, posterior(8192) %.1f ms; loglike %.6f; mean in [%.3f, %.3f], var in [%.4g, %.4g]"
      % (N, len(S), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3),
         -0.5 * quad - 0.5 * ld - N / 2 * np.log(2 * np.pi), mean.min(), mean.max(), var.min(), var.max()))
# training-point residual check at a few points: the FITC mean at training inputs tracks y up to the noise level
mt, _ = m.posterior(coeff, dev.points(ctx, Xh[:2048]))
print("rms(mean - y) at training points: %.3f (noise std %.3f)" % (float(np.sqrt(np.mean((mt - y[:2048]) ** 2))), np.sqrt(0.1)))
