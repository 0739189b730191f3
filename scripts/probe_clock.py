"""Sample rocm-smi clocks/power while a long series of GEMMs (or the MFMA microbenchmark) runs: is the GEMM clock-bound?"""
import sys, os, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpexp_amd import device as dev

samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l)]
            samples.append((time.time(), " | ".join(keep)))
        except Exception as e:  # noqa
            samples.append((time.time(), "ERR %r" % (e,)))
        time.sleep(0.3)


ctx = dev.context()
n = 16384
rng = np.random.default_rng(1)
mode = sys.argv[1] if len(sys.argv) > 1 else "random"
blk = rng.standard_normal((1024, n)) if mode == "random" else np.zeros((1024, n))
A = dev.DeviceMatrix.from_host(ctx, np.tile(blk, (n // 1024, 1)))
B = dev.DeviceMatrix.from_host(ctx, np.tile(blk[::-1], (n // 1024, 1)))
Cm = dev.DeviceMatrix.zeros(ctx, n, n)
th = threading.Thread(target=poll)
th.start()
time.sleep(1.0)
t0 = time.time()
ctx.profile(True); ctx.profile_reset()
for i in range(40):
    dev.dbg_gemm(ctx, A, B, Cm, 1, 0, 0)
ctx.sync()
p = ctx.profile_get()["gemm"]; ctx.profile(False)
t1 = time.time()
time.sleep(1.0)
stop = True
th.join()
print("mode=%s: 40 x NT gemm %d^3: %.1f ms each, %.2f TF/s" % (mode, n, p["ms"] / 40, p["flops"] / p["ms"] / 1e9))
for ts, s in samples:
    print("%6.2f %s %s" % (ts - t0, "RUN " if t0 <= ts <= t1 else "idle", s))
