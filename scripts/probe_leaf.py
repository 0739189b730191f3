"""The 128 x 128 Cholesky leaf alone: phase breakdown from the kernel's own time stamps (gpx_dbg_leaf_stamps; shader clocks)
for round 5's diagonal step (fast = 1) and the general one (fast = 0), each on a fresh SPD block, checked against LAPACK."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev

ctx = dev.context()
rng = np.random.default_rng(0)
A = rng.standard_normal((128, 128))
A = A @ A.T + 128 * np.eye(128)
Lref = np.linalg.cholesky(A)
mhz = ctx.info()["clock_mhz"] if isinstance(ctx.info(), dict) and "clock_mhz" in ctx.info() else 2400
for fast in (1, 0):
    best = None
    for it in range(6):
        K = dev.DeviceMatrix.from_host(ctx, A)
        st = (C.c_int64 * 30)()
        dev.check(ctx.lib.gpx_dbg_leaf_stamps(ctx.h, K.h, fast, st))
        s = np.array(list(st), dtype=np.int64)
        if best is None or s[29] - s[28] < best[29] - best[28]:
            best = s
        if it == 0:
            Kh = K.to_host()[:128, :128]
            L = np.tril(Kh)
            print("fast=%d: max |L - L_lapack| / max|L| = %.2e, upper part written as zeros: %s"
                  % (fast, np.abs(L - Lref).max() / np.abs(Lref).max(), bool(np.all(np.triu(Kh, 16) == 0))))
    s = best
    wall_us = (s[29] - s[28]) / 100.0                 # whole kernel: start of wave 0 -> end of wave 1 (100 MHz clock)
    mhz = float(os.environ.get("GPX_SHADER_MHZ", "2393"))   # shader clock the leaf holds (measured in round 5: 2393)
    us = lambda c: c / mhz
    diag = [s[3 + 3 * p] - s[2 + 3 * p] for p in range(8)]
    rest = [s[4 + 3 * p] - s[3 + 3 * p] for p in range(8)]
    print("fast=%d: kernel %.1f us (wall clock, incl. the inverse's last row and the drain of the stores); wave 0: %.1f us | "
          "block (0,0) in LDS after %.2f | diag steps %s (sum %.1f) | scale + column update %s (sum %.1f)"
          % (fast, wall_us, us(s[27] - s[0]), us(s[1] - s[0]), " ".join("%.2f" % us(x) for x in diag), us(sum(diag)),
             " ".join("%.2f" % us(x) for x in rest), us(sum(rest))), flush=True)
