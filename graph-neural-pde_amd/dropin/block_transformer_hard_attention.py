"""Drop-in for the reference's src/block_transformer_hard_attention.py."""
from gnpde_amd.block_transformer_hard_attention import HardAttODEblock  # noqa: F401
