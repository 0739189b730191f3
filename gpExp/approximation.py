"""gpExp.approximation -> gpexp_amd.approximation (see gpExp/__init__.py)."""
from gpexp_amd.approximation import *  # noqa: F401,F403
from gpexp_amd import approximation as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
