"""Name -> class lookup used by the model constructors (`opt['block']`, `opt['function']`), with the names and the
two exception types of the reference's src/model_configurations.py:17-44.  Every block and function the
reference registers is available; the classes are resolved on first use."""
import importlib


class BlockNotDefined(Exception):
  pass


class FunctionNotDefined(Exception):
  pass


_BLOCK_TABLE = {
  'constant': ('block_constant', 'ConstantODEblock'),
  'attention': ('block_transformer_attention', 'AttODEblock'),
  'mixed': ('block_mixed', 'MixedODEblock'),
  'hard_attention': ('block_transformer_hard_attention', 'HardAttODEblock'),
  'rewire_attention': ('block_transformer_rewiring', 'RewireAttODEblock'),
}
_FUNCTION_TABLE = {
  'laplacian': ('function_laplacian_diffusion', 'LaplacianODEFunc'),
  'transformer': ('function_transformer_attention', 'ODEFuncTransformerAtt'),
  'GAT': ('function_GAT_attention', 'ODEFuncAtt'),
}


def _lookup(table, key, error):
  if key not in table:
    raise error('%r is not one of %s' % (key, ', '.join(sorted(table))))
  module, name = table[key]
  return getattr(importlib.import_module('.' + module, __package__), name)


def set_block(opt):
  return _lookup(_BLOCK_TABLE, opt['block'], BlockNotDefined)


def set_function(opt):
  return _lookup(_FUNCTION_TABLE, opt['function'], FunctionNotDefined)
