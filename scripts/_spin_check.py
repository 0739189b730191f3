import sys, time, os
sys.path.insert(0, "/root/repo")
from gpexp_amd import device as dev
ctx = dev.context()
for us in (1000, 5000):
    ctx.sync(); t0 = time.perf_counter(); ctx.lib.gpx_dbg_spin_us(ctx.h, us); ctx.sync(); print(us, "us ->", 1e3 * (time.perf_counter() - t0), "ms")
ctx.sync(); t0 = time.perf_counter(); ctx.lib.gpx_dbg_spin(ctx.h, 5); ctx.sync(); print("5 ms ->", 1e3 * (time.perf_counter() - t0), "ms")
