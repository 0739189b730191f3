import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, warnings
from conftest import Golden
from gpExp.kernels import KernelSquaredExponential
from gpExp.experimentalDesign import costFunctionGP_IVAR, ExperimentalDesignDerivative, performGreedyVarExperimentalDesign
from gpExp.gp import GP
from gpExp.approximation import Space
golden = Golden(); c = "demo_flow"
def rel(a,b): return np.max(np.abs(np.asarray(a)-np.asarray(b)))/np.max(np.abs(b))
gpT = GP(KernelSquaredExponential([0.3], 1.0, 1), 0.0)
xTrain, yTrain = golden(c, "xTrain"), golden(c, "yTrain")
print("ll0", abs(gpT.computeLogLike(xTrain, yTrain)/float(golden(c,"loglike0"))-1))
params, optval = gpT.findOptParamsLogLike(xTrain, yTrain)
print("opt", params, float(golden(c,"opt_cl0")), float(golden(c,"opt_signalSize")), float(golden(c,"opt_noise")), optval, float(golden(c,"opt_value")))
gpT.updateKernelParams({"cl0": float(golden(c, "opt_cl0")), "signalSize": float(golden(c, "opt_signalSize")), "noise": float(golden(c, "opt_noise"))})
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    gpT.train(xTrain, yTrain)
    print("warnings:", [str(x.message)[:80] for x in w])
m, var = gpT.evaluate(np.linspace(-1, 1, 1000).reshape((1000, 1)), compvar=1)
print("mean1", rel(m, golden(c,"mean1")), "var1 abs", np.max(np.abs(var-golden(c,"var1"))))
mc = golden(c, "mc")
space = Space(1, lambda size: np.random.rand(size[0], size[1]) * 2.0 - 1.0, lambda p: (np.abs(p) < 1.0) * 0.5)
cf = costFunctionGP_IVAR(gpT, 8, space, mcPoints=mc)
keep = [0, 1, 2, 3]
with contextlib.redirect_stdout(io.StringIO()):
    start = performGreedyVarExperimentalDesign(gpT.kernel, np.concatenate((xTrain, mc), axis=0), 8, 1, indKeepStart=keep)
print("start cost", abs(cf.evaluate(start)/float(golden(c,"greedy_start_cost"))-1), "grad", rel(cf.derivative(start), golden(c,"greedy_start_grad")))
exp = ExperimentalDesignDerivative(cf, 8, 1)
lb = np.concatenate((xTrain.flatten(), -np.ones(4))); ub = np.concatenate((xTrain.flatten(), np.ones(4)))
with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    design = exp.beginWithVarGreedy(nodesKeep=xTrain, lbounds=lb, rbounds=ub)
    print("design warnings:", len(w), file=sys.stderr)
print("design cost", abs(cf.evaluate(design)/float(golden(c,"design_cost"))-1), "design", np.max(np.abs(np.sort(design[:,0])-np.sort(golden(c,"design")[:,0]))))
