"""gpExp.experimentalDesign -> gpexp_amd.experimentalDesign (see gpExp/__init__.py)."""
from gpexp_amd.experimentalDesign import *  # noqa: F401,F403
from gpexp_amd import experimentalDesign as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
