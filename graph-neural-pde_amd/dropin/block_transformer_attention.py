"""Drop-in for the reference's src/block_transformer_attention.py."""
from gnpde_amd.block_transformer_attention import AttODEblock  # noqa: F401
