"""Drop-in for the reference's src/block_mixed.py."""
from gnpde_amd.block_mixed import MixedODEblock  # noqa: F401
