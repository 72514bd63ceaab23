"""Drop-in for the reference's src/function_transformer_attention.py."""
from gnpde_amd.function_transformer_attention import ODEFuncTransformerAtt, SpGraphTransAttentionLayer  # noqa: F401
