"""Drop-in import name: `import gpExp...` resolves to the MI355X-native implementation in gpexp_amd, so that the
reference's scripts (demo.py: `from gpExp.kernels import ...`, `from gpExp.experimentalDesign import *`) run unmodified."""
