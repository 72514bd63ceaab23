"""Drop-in for the reference's src/function_laplacian_diffusion.py."""
from gnpde_amd.function_laplacian_diffusion import LaplacianODEFunc  # noqa: F401
