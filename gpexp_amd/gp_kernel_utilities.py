"""Covariance assembly with the reference's interface (gpExp/gp_kernel_utilities.py:34-68), on the GPU.

The reference fills K with one `kernel.evaluate` call per row (N Python iterations, an (N,d) `np.tile`
temporary each) and then adds `np.diag(nugget)`; here one tiled HIP kernel (`gpx_kfill`) writes K and the
nugget in a single pass over HBM.  The FITC / Nystrom helpers of the reference file (:70-232) are outside the
hot path and are not provided.
"""
import numpy as np

from . import device as _dev


def _check_nugget(nugget):
    # the reference accepts float or ndarray only; anything else (e.g. an int) dies with UnboundLocalError
    # at gp_kernel_utilities.py:67 -- reported here as a TypeError with the reason
    if not isinstance(nugget, (float, np.ndarray)):
        raise TypeError("nugget must be a float or an ndarray, got %s (gp_kernel_utilities.py:62-67)"
                        % type(nugget).__name__)


def covariance_on_device(kernel, points, nugget=0.0):
    """K(points, points) + diag(nugget) as a DeviceMatrix (stays in HBM for the factorisation)."""
    _check_nugget(nugget)
    points = np.asarray(points, dtype=float)
    assert points.ndim == 2 and points.shape[1] == kernel.dimension, \
        (" Incorrect dimension of input points fed to kernel ", points.shape)
    ctx = _dev.context()
    X = _dev.points(ctx, points)
    return X, _dev.kfill(ctx, kernel._spec(), X, nugget=nugget)


def calculateCovarianceMatrix(kernel, points, nugget=0.0):
    """Dense (N, N) covariance matrix as a NumPy array; `nugget` is a float (scalar * I) or an (N,) ndarray."""
    _, K = covariance_on_device(kernel, points, nugget)
    return K.to_host()
