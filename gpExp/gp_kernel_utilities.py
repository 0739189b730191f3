"""gpExp.gp_kernel_utilities -> gpexp_amd.gp_kernel_utilities (see gpExp/__init__.py)."""
from gpexp_amd.gp_kernel_utilities import *  # noqa: F401,F403
from gpexp_amd import gp_kernel_utilities as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
