"""Drop-in for the reference's src/block_transformer_rewiring.py."""
from gnpde_amd.block_transformer_rewiring import RewireAttODEblock  # noqa: F401
