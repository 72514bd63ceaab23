"""String -> class registry with the reference's names (src/model_configurations.py:17-44).
All five blocks and three functions of the reference are registered."""
from .function_transformer_attention import ODEFuncTransformerAtt
from .function_GAT_attention import ODEFuncAtt
from .function_laplacian_diffusion import LaplacianODEFunc
from .block_transformer_attention import AttODEblock
from .block_constant import ConstantODEblock
from .block_mixed import MixedODEblock
from .block_transformer_hard_attention import HardAttODEblock
from .block_transformer_rewiring import RewireAttODEblock


class BlockNotDefined(Exception):
  pass


class FunctionNotDefined(Exception):
  pass


_BLOCKS = {'attention': AttODEblock, 'constant': ConstantODEblock, 'mixed': MixedODEblock,
           'hard_attention': HardAttODEblock, 'rewire_attention': RewireAttODEblock}
_FUNCTIONS = {'laplacian': LaplacianODEFunc, 'GAT': ODEFuncAtt, 'transformer': ODEFuncTransformerAtt}


def set_block(opt):
  name = opt['block']
  if name in _BLOCKS:
    return _BLOCKS[name]
  raise BlockNotDefined


def set_function(opt):
  name = opt['function']
  if name in _FUNCTIONS:
    return _FUNCTIONS[name]
  raise FunctionNotDefined
