"""gpExp.gp -> gpexp_amd.gp (see gpExp/__init__.py)."""
from gpexp_amd.gp import *  # noqa: F401,F403
from gpexp_amd import gp as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
