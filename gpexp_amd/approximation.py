"""Domain descriptor with the reference's interface (gpExp/approximation.py:22-36): a plain attribute holder."""


class Space:
    """Describes the design space: dimension, a sampler `sample((n, d))`, a density `probDensity(points)` and an
    optional heteroscedastic noise callable `noiseFunc(points)` (per-point nugget)."""

    dimension = None
    inBoundsBool = None
    sample = None
    probDensity = None
    noiseFunc = None

    def __init__(self, dimensionIn, samplerIn, probDensityIn, noise=None):
        self.dimension = dimensionIn
        self.sample = samplerIn
        self.probDensity = probDensityIn
        self.noiseFunc = noise
