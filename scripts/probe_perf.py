"""Ad-hoc timing probe (not the bench): phases at a given N with per-class kernel timing."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpexp_amd import device as dev

def run(N, d, M, kind="matern52"):
    ctx = dev.context()
    rng = np.random.default_rng(N)
    Xh = rng.uniform(-1, 1, (N, d)); y = np.sin(2*np.pi*Xh.sum(1)/d) + np.sqrt(0.1)*rng.standard_normal(N)
    Zh = rng.uniform(-1, 1, (M, d))
    if kind == "se":
        sp = dev.KernelSpec(0, d, list(0.4+0.05*np.arange(d)) + [1.0])
    else:
        sp = dev.KernelSpec(2, d, [0.5, 1.0])
    X = dev.points(ctx, Xh); Z = dev.points(ctx, Zh)
    K = dev.DeviceMatrix.zeros(ctx, N, N)
    for it in range(2):
        ctx.profile(True); ctx.profile_reset()
        t0 = time.perf_counter(); dev.kfill_into(ctx, sp, X, K, nugget=0.1); ctx.sync(); t1 = time.perf_counter()
        dev.potrf(ctx, K); t2 = time.perf_counter()
        alpha = dev.potrs(ctx, K, y); t3 = time.perf_counter()
        ld = dev.logdet(ctx, K); t4 = time.perf_counter()
        iv = dev.ivar(ctx, sp, K, X, Z); t5 = time.perf_counter()
        prof = ctx.profile_get(); ctx.profile(False)
        print(f"N={N} d={d} M={M} it={it}: kfill {1e3*(t1-t0):.2f} ms  potrf {1e3*(t2-t1):.1f} ms ({N**3/3/(t2-t1)/1e12:.1f} TF)  potrs {1e3*(t3-t2):.1f}  logdet {1e3*(t4-t3):.2f}  ivar {1e3*(t5-t4):.1f} ms ({N*N*M/(t5-t4)/1e12:.1f} TF)  total {1e3*(t5-t0):.1f} ms  ivar={iv:.6g} logdet={ld:.6g}", flush=True)
        for k, v in prof.items():
            if v["launches"]:
                extra = f" {v['flops']/v['ms']/1e9:.2f} TF/s" if v["flops"] and v["ms"] else ""
                extra += f" {v['bytes']/v['ms']/1e6:.1f} GB/s" if v["bytes"] and v["ms"] else ""
                print(f"    {k:7s} launches={v['launches']:6d} ms={v['ms']:.2f}{extra}", flush=True)

if __name__ == "__main__":
    for a in sys.argv[1:]:
        N, d, M = map(int, a.split(","))
        run(N, d, M)
