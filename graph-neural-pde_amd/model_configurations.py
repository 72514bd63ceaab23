"""String -> class registry with the reference's names (src/model_configurations.py:17-44).
Blocks outside SURVEY.md section 8's current rows raise BlockNotDefined with a pointer to it."""
from .function_transformer_attention import ODEFuncTransformerAtt
from .function_GAT_attention import ODEFuncAtt
from .function_laplacian_diffusion import LaplacianODEFunc
from .block_transformer_attention import AttODEblock
from .block_constant import ConstantODEblock
from .block_mixed import MixedODEblock
from .block_transformer_hard_attention import HardAttODEblock


class BlockNotDefined(Exception):
  pass


class FunctionNotDefined(Exception):
  pass


_BLOCKS = {'attention': AttODEblock, 'constant': ConstantODEblock, 'mixed': MixedODEblock,
           'hard_attention': HardAttODEblock}
_FUNCTIONS = {'laplacian': LaplacianODEFunc, 'GAT': ODEFuncAtt, 'transformer': ODEFuncTransformerAtt}


def set_block(opt):
  name = opt['block']
  if name in _BLOCKS:
    return _BLOCKS[name]
  if name in ('rewire_attention',):
    raise BlockNotDefined('block %r is a "next" row of SURVEY.md section 8f, not built yet' % name)
  raise BlockNotDefined


def set_function(opt):
  name = opt['function']
  if name in _FUNCTIONS:
    return _FUNCTIONS[name]
  raise FunctionNotDefined
