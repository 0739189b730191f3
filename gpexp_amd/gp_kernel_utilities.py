"""Covariance assembly with the reference's interface (gpExp/gp_kernel_utilities.py:34-68), on the GPU.

The reference fills K with one `kernel.evaluate` call per row (N Python iterations, an (N,d) `np.tile`
temporary each) and then adds `np.diag(nugget)`; here one tiled HIP kernel (`gpx_kfill`) writes K and the
nugget in a single pass over HBM.  SURVEY.md 8 f4: the FITC helper (:70-104) runs on the device FITC model
(gpx_fitc_*), and the Nystrom eigen-basis (:107-194) keeps ARPACK on the host with the covariance resident in HBM and
every operator application a device row reduction.
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


def calculateCovarianceMatrixFITC(kernel, nodes, nugget, fitc, returnCov=False):
    """Dense FITC precision (and covariance) with the reference's return convention (gp_kernel_utilities.py:70-104):
    `fitc` is a float (fraction of the nodes, drawn with np.random.permutation) or an array of inducing points."""
    nodes = np.asarray(nodes, dtype=float)
    nNodes = len(nodes)
    if isinstance(fitc, float):
        nu = int(np.floor(nNodes * fitc))
        indu = np.random.permutation(nNodes)[0:nu]
        snodes = np.array(nodes[indu], dtype=float)
    else:
        snodes = np.asarray(fitc, dtype=float)
    ctx = _dev.context()
    model = _dev.FitcModel(ctx, kernel._spec(), _dev.points(ctx, nodes), _dev.points(ctx, snodes), float(nugget))
    covmat, precMat = model.dense(cov=bool(returnCov), prec=True)
    if returnCov is False:
        return precMat, snodes
    return covmat, precMat, snodes


def covTimesV(b, kernel, mcPoints):
    """K(mcPoints, mcPoints) @ b (gp_kernel_utilities.py:107-143: one kernel row per entry, forked over processes).
    The covariance stays in HBM between calls with the same point set; each application is one device reduction."""
    b = np.asarray(b, dtype=float)
    ctx = _dev.context()
    pts = np.ascontiguousarray(mcPoints, dtype=float)
    # keyed on the CONTENT of the point set (object ids are recycled and arrays can be edited in place): comparing
    # N x d doubles per application is nothing next to the N x N product it guards
    key = (pts.shape, kernel._spec().kind, tuple(kernel._spec().hyp.tolist()))
    c = covTimesV._cache
    if c is None or c[0] != key or not np.array_equal(c[1], pts):
        X = _dev.points(ctx, pts)
        covTimesV._cache = c = (key, pts.copy(), _dev.kfill(ctx, kernel._spec(), X, nugget=0.0))
    return _dev.matvec(ctx, c[2], b.ravel()).reshape(b.shape)


covTimesV._cache = None


def calculateKernelBasisFunctionsMC(kernel, numBasis, mcPoints):
    """Leading eigen-pairs of the kernel by Monte Carlo + Nystrom (gp_kernel_utilities.py:145-194): ARPACK (eigsh) on the
    matrix-free operator v -> K v, descending order, eigenvalues / nMC and eigenvectors * sqrt(nMC)."""
    from scipy.sparse.linalg import LinearOperator, eigsh
    nMC = mcPoints.shape[0]
    numberEigenVectors = int(min(numBasis, nMC))
    A = LinearOperator((nMC, nMC), matvec=lambda v, k=kernel, p=mcPoints: covTimesV(v, k, p), dtype=float)
    eigv, eigve = eigsh(A, k=numberEigenVectors, maxiter=10 * numberEigenVectors)
    eigv = eigv[::-1]
    eigve = eigve[:, ::-1]
    return eigv / float(nMC), eigve * np.sqrt(float(nMC))
