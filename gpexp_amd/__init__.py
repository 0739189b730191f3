"""gpexp_amd -- MI355X-native GP-inference hot path behind the GPEXP class API.

Host side: plain Python mirroring gpExp.{kernels,gp,gp_kernel_utilities,experimentalDesign,approximation};
device side: hand-written HIP kernels for gfx950 in libgpx_hip.so, bound through ctypes (include/gpx.h).
There is no CPU fallback: importing the compute modules without the built library raises ImportError and
creating a context without an MI355X raises RuntimeError.
"""
__version__ = "0.1.0"
