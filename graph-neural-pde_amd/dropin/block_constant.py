"""Drop-in for the reference's src/block_constant.py."""
from gnpde_amd.block_constant import ConstantODEblock  # noqa: F401
