"""gpExp.kernels -> gpexp_amd.kernels (see gpExp/__init__.py)."""
from gpexp_amd.kernels import *  # noqa: F401,F403
from gpexp_amd import kernels as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
