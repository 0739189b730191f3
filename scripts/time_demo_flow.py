"""BASELINE config C1 (the reference's demo flow: 4 training points, 1000 test points, 10 000 MC points, hyper-parameter fit,
greedy start + SLSQP design of 4 more points) timed phase by phase.  `--impl gpx` runs this repository's gpExp (GPU box),
`--impl ref` the reference itself (build container only: PYTHONPATH=/root/reference; it cannot travel to the GPU box).
Inputs: the committed fixture tests/golden (demo_flow), so both run the same numbers."""
import argparse, contextlib, io, json, os, sys, time, warnings
ap = argparse.ArgumentParser()
ap.add_argument("--impl", choices=["gpx", "ref"], default="gpx")
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import Golden          # (puts the repository root on sys.path)
if args.impl == "ref":
    sys.dont_write_bytecode = True
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    sys.path.insert(0, "/root/reference")
else:
    sys.path.insert(0, ROOT)
from gpExp.kernels import KernelSquaredExponential
from gpExp.experimentalDesign import costFunctionGP_IVAR, ExperimentalDesignDerivative, performGreedyVarExperimentalDesign
from gpExp.gp import GP
from gpExp.approximation import Space
golden = Golden(); c = "demo_flow"
T = {}
def timed(name, fn):
    t0 = time.perf_counter(); r = fn(); T[name] = time.perf_counter() - t0; return r
xTrain, yTrain, mc = golden(c, "xTrain"), golden(c, "yTrain"), golden(c, "mc")
gpT = GP(KernelSquaredExponential([0.3], 1.0, 1), 0.0)
with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    timed("first_loglike (incl. device / library start-up)", lambda: gpT.computeLogLike(xTrain, yTrain))
    timed("findOptParamsLogLike", lambda: gpT.findOptParamsLogLike(xTrain, yTrain))
    gpT.updateKernelParams({"cl0": float(golden(c, "opt_cl0")), "signalSize": float(golden(c, "opt_signalSize")), "noise": float(golden(c, "opt_noise"))})
    timed("train", lambda: gpT.train(xTrain, yTrain))
    timed("evaluate 1000 points, compvar=1", lambda: gpT.evaluate(np.linspace(-1, 1, 1000).reshape((1000, 1)), compvar=1))
    space = Space(1, lambda size: np.random.rand(size[0], size[1]) * 2.0 - 1.0, lambda p: (np.abs(p) < 1.0) * 0.5)
    cf = costFunctionGP_IVAR(gpT, 8, space, mcPoints=mc)
    start = timed("greedy variance start (8 of 10 004)", lambda: performGreedyVarExperimentalDesign(gpT.kernel, np.concatenate((xTrain, mc), axis=0), 8, 1, indKeepStart=[0, 1, 2, 3]))
    timed("IVAR cost, one evaluation (10 000 MC points)", lambda: cf.evaluate(start))
    timed("IVAR gradient, one evaluation", lambda: cf.derivative(start))
    exp = ExperimentalDesignDerivative(cf, 8, 1)
    lb = np.concatenate((xTrain.flatten(), -np.ones(4))); ub = np.concatenate((xTrain.flatten(), np.ones(4)))
    design = timed("SLSQP design of 4 new points (beginWithVarGreedy)", lambda: exp.beginWithVarGreedy(nodesKeep=xTrain, lbounds=lb, rbounds=ub))
    cost = cf.evaluate(design)
T["total"] = sum(T.values())
print(json.dumps({"impl": args.impl, "design_cost": float(cost), "seconds": {k: round(v, 4) for k, v in T.items()}}))
