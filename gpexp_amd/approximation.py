"""Domain descriptor with the reference's interface (gpExp/approximation.py:22-36): a plain attribute holder."""


class Space:
    """Design space: `dimension`, a sampler `sample((n, d)) -> (n, d) points`, a density `probDensity(points)` and an
    optional heteroscedastic noise model `noiseFunc` (callable on points -> per-point nugget; `.deriv(points)` when the
    design gradient is wanted, demo2.py:45-58).  Attribute names are the reference's: callers read them directly
    (experimentalDesign.py:72, 107-112)."""

    inBoundsBool = None   # declared by the reference, never set or read anywhere

    def __init__(self, dimensionIn, samplerIn, probDensityIn, noise=None):
        self.dimension, self.sample, self.probDensity, self.noiseFunc = dimensionIn, samplerIn, probDensityIn, noise
