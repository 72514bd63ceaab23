"""Drop-in for the reference's src/function_GAT_attention.py."""
from gnpde_amd.function_GAT_attention import ODEFuncAtt, SpGraphAttentionLayer  # noqa: F401
