"""Covariance-kernel classes with the reference's interface (gpExp/kernels.py), evaluated on the GPU.

Mirrors `Kernel` (kernels.py:30-70), `KernelIsoMatern` (:72-97), `KernelSquaredExponential` (:100-181),
`KernelMehlerND` (:183-248) and `KernelMehler1D` (:250-324): same constructor arguments, the same
`hyperParam` dictionaries and key names, the same shape rules and assertion messages.  The arithmetic of
`evaluate` runs in libgpx_hip.so (`gpx_kernel_eval`); dense assemblies never go through `evaluate` at all
(see gp_kernel_utilities.calculateCovarianceMatrix).

Differences from the reference, on purpose:
  * Matern nu=5/2 is implemented (the reference leaves `out` unbound for nu != 3/2, kernels.py:85-91);
    any other nu raises NotImplementedError instead of UnboundLocalError.
  * `KernelSquaredExponential.derivativeWrtHypParams` works (the reference indexes with a float,
    kernels.py:140-141, and raises IndexError on current NumPy).
  * A NaN from the Mehler kernel raises FloatingPointError instead of print + exit() (kernels.py:288-292).
"""
import numpy as np

from . import device as _dev


class Kernel(object):
    """Base class: holds `dimension` and the `hyperParam` dict; default kernel is the constant 0."""

    nugget = 0.0
    hyperParam = dict({})

    def __init__(self, hyperParam, dimension, *argc):
        self.dimension = dimension
        self.hyperParam = hyperParam
        super(Kernel, self).__init__()

    def updateHyperParameters(self, hyperParamNew):
        for key in hyperParamNew.keys():
            assert key in self.hyperParam.keys(), (key, " is not a valid hyperParameter")
        self.hyperParam = hyperParamNew

    # -- flat (kind, d, hyp[]) description for the C ABI; subclasses override ------------------------
    def _spec(self):
        return None

    def evaluate(self, x1, x2):
        """k(x1[i], x2[i]) for (n,d) vs (n,d), or one point against n; returns a 1-D (n,) array."""
        assert len(x2.shape) > 1 and len(x1.shape) > 1, "Must supply nd arrays to evaluation function"
        assert x1.shape[1] == self.dimension and x2.shape[1] == self.dimension, \
            (" Incorrect dimension of input points fed to kernel ", x1.shape, x2.shape)
        n1, n2 = x1.shape[0], x2.shape[0]
        assert n1 == n2 or n1 == 1 or n2 == 1, "__evaluate() received non-equal shaped point sets"
        return self.evaluateF(x1, x2)

    def evaluateF(self, x1, x2):
        """Paired evaluation; the base kernel is identically 0 (kernels.py:67-70)."""
        spec = self._spec()
        if spec is None:
            return 0
        n = max(x1.shape[0], x2.shape[0])
        if n == 0:
            return np.zeros((0,))
        return _dev.kernel_eval(_dev.context(), spec, x1, x2)


class KernelIsoMatern(Kernel):
    """Isotropic Matern kernel, nu = 3/2 (reference) or 5/2 (extension); nu is not a hyper-parameter."""

    def __init__(self, rho, signalSize, dimension, nu=3.0 / 2.0):
        self.nu = nu
        super(KernelIsoMatern, self).__init__(dict({'rho': rho, 'signalSize': signalSize}), dimension)

    def _spec(self):
        if np.abs(1.5 - self.nu) < 1e-10:
            kind = _dev.K_MATERN32
        elif np.abs(2.5 - self.nu) < 1e-10:
            kind = _dev.K_MATERN52
        else:
            raise NotImplementedError("KernelIsoMatern supports nu = 3/2 (reference) and nu = 5/2 only")
        return _dev.KernelSpec(kind, self.dimension, [self.hyperParam['rho'], self.hyperParam['signalSize']])

    def derivativeWrtHypParams(self, x1, x2):
        assert x1.shape == x2.shape, "__evaluate() received non-equal shaped point sets"
        raise AttributeError("derivativeWrtHypParams not implemented for KernelIsoMatern")


class KernelSquaredExponential(Kernel):
    """signalSize * exp(-1/2 sum_k (x_k - x'_k)^2 / cl_k^2); a length-1 correlationLength is isotropic."""

    def __init__(self, correlationLength, signalSize, dimension):
        # keys 'cl0'..'cl{d-1}' then 'signalSize', in that order (kernels.py:103-112): the order is API -- it is the
        # order of getHypParamNames() and of the optimiser's parameter vector (gp.py:566-582)
        lengths = list(correlationLength) * dimension if len(correlationLength) == 1 else list(correlationLength)
        params = {'cl%d' % k: lengths[k] for k in range(len(lengths))}
        params['signalSize'] = signalSize
        super(KernelSquaredExponential, self).__init__(params, dimension)

    def _cl(self):
        return np.array([self.hyperParam['cl' + str(ii)] for ii in range(self.dimension)], dtype=float)

    def _spec(self):
        return _dev.KernelSpec(_dev.K_SE, self.dimension, list(self._cl()) + [self.hyperParam['signalSize']])

    def derivativeWrtHypParams(self, x1, x2):
        """{key: dK/dkey} at the current hyper-parameters for paired points (kernels.py:125-144):
        d/d signalSize = exp(.), d/d cl_k = K * (x1_k - x2_k)^2 / cl_k^3."""
        assert x1.shape == x2.shape, "__evaluate() received non-equal shaped point sets"
        cl = self._cl()
        evals = self.evaluateF(x1, x2)
        out = {}
        for key in self.hyperParam.keys():
            if key == 'signalSize':
                out[key] = evals / self.hyperParam['signalSize']
            else:
                direction = int(key[2:])
                out[key] = evals * (x1[:, direction] - x2[:, direction]) ** 2.0 / cl[direction] ** 3.0
        return out

    def derivative(self, x1, x2, version=0):
        """out[j, i] = dK(x1[j], x2) / d x1[j, i] for a single point x2 (1, d) (kernels.py:146-181)."""
        assert len(x2.shape) > 1 and len(x1.shape) > 1, "Must supply nd arrays to evaluation function"
        assert x2.shape[0] == 1 and x2.shape[1] == self.dimension, "x2 not in correct shape"
        assert x1.shape[0] > 0 and x1.shape[1] == self.dimension, "x1 not in correct shape"
        cl = self._cl()
        if version == 0 or version == 1:
            rEvals = self.evaluate(x1, x2)
            # the reference multiplies by signalSize a second time (kernels.py:177); kept for parity
            return -self.hyperParam['signalSize'] * (x1 - x2) / cl[None, :] ** 2.0 * rEvals[:, None]


class KernelMehlerND(Kernel):
    """Product of 1-D Mehler kernels; hyper-parameter keys are the ints 0..d-1 (kernels.py:183-198)."""

    def __init__(self, tIn, dimension):
        hyperParam = dict({})
        self.oneDKern = []
        for ii in range(dimension):
            hyperParam[ii] = tIn[ii]
            self.oneDKern.append(KernelMehler1D(tIn[ii], 1))
        super(KernelMehlerND, self).__init__(hyperParam, dimension)

    def updateHyperParameters(self, params):
        for keys in self.hyperParam.keys():
            self.hyperParam[keys] = params[keys]
        for ii in range(self.dimension):
            self.oneDKern[ii].updateHyperParameters(dict({'t': self.hyperParam[ii]}))

    def _spec(self):
        return _dev.KernelSpec(_dev.K_MEHLER, self.dimension, [self.hyperParam[ii] for ii in range(self.dimension)])

    def evaluateF(self, x1, x2):
        out = super(KernelMehlerND, self).evaluateF(x1, x2)
        if out.size and np.isnan(out[0]):
            raise FloatingPointError("NAN in Mehler kernel (t=%r)" % (self.hyperParam,))
        return out

    def derivative(self, x1, x2):
        raise AttributeError("derivative of KernelMehlerND not yet implemented")


class KernelMehler1D(Kernel):
    """1-D Mehler (Hermite) kernel with parameter 't' (kernels.py:250-293)."""

    def __init__(self, tIn, dimension):
        assert dimension == 1, "Mehler Hermite Kernel is only one dimensional"
        super(KernelMehler1D, self).__init__(dict({'t': tIn}), dimension)

    def _spec(self):
        return _dev.KernelSpec(_dev.K_MEHLER, 1, [self.hyperParam['t']])

    def evaluateF(self, x1, x2):
        assert x1.shape[1] == 1 and x2.shape[1] == 1, "Hermite1d kernel only accepts one dimensional points"
        out = super(KernelMehler1D, self).evaluateF(x1, x2)
        if out.size and np.isnan(out[0]):
            raise FloatingPointError("NAN in kernel hermi1d (t=%r)" % (self.hyperParam['t'],))
        return out

    def derivative(self, x1, x2):
        """out[j, 0] = dK(x1[j], x2)/d x1[j] for a single point x2 (kernels.py:295-324)."""
        assert len(x2.shape) > 1 and len(x1.shape) > 1, "Must supply nd arrays to evaluation function"
        assert x2.shape[0] == 1 and x2.shape[1] == self.dimension, "x2 not in correct shape"
        assert x1.shape[0] > 0 and x1.shape[1] == self.dimension, "x1 not in correct shape"
        t = self.hyperParam['t']
        rEvals = self.evaluate(x1, x2)
        return -0.5 * (2.0 * x1 * t ** 2.0 - 2.0 * t * x2) / (1.0 - t ** 2.0) * rEvals[:, None]
