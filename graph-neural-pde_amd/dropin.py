"""Run the reference's own scripts (`run_GNN.py`, `GNN.py`, `GNN_early.py`, `model_configurations.py`) on the MI355X classes
without editing them.

The reference looks its functions and blocks up by MODULE name (`from function_transformer_attention import
ODEFuncTransformerAtt`, src/model_configurations.py:1-9; `from base_classes import BaseGNN`, src/GNN.py:4; `from
early_stop_solver import EarlyStopInt`, src/GNN_early.py:10).  `install()` answers those imports with the modules of this
package, whatever the order of `sys.path`:

    python -m gnpde_amd.dropin [--native-gnn] /path/to/graph-neural-pde/src/run_GNN.py --dataset Cora --function transformer ...

or, from Python, `import gnpde_amd.dropin; gnpde_amd.dropin.install()` before the first import of a reference module.

* the function / block / early-stopping modules are this package's modules under the reference's names (`MODULES`);
* `base_classes` is MERGED, lazily, at its first import: everything the reference's own file defines (`BaseGNN`, the
  regulariser registry `REGULARIZATION_FNS`, ...) with `ODEFunc`, `ODEblock` and `RegularizedODEfunc` replaced by this
  package's -- the reference file is searched on `sys.path` at that moment, so the path may be set up after `install()`;
* `GNN` (optional, `native_gnn=True` / `--native-gnn`): the model of `gnpde_amd/GNN.py`, whose encoder and `relu -> m2`
  decoder are single native launches at test time; without it the reference's own `GNN.py` runs over the classes above.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys
import types

# reference module name (src/<name>.py) -> module of this package that provides its classes
MODULES = {
  'function_laplacian_diffusion': 'gnpde_amd.function_laplacian_diffusion',      # LaplacianODEFunc
  'function_transformer_attention': 'gnpde_amd.function_transformer_attention',  # ODEFuncTransformerAtt, SpGraphTransAttentionLayer
  'function_GAT_attention': 'gnpde_amd.function_GAT_attention',                  # ODEFuncAtt, SpGraphAttentionLayer
  'block_constant': 'gnpde_amd.block_constant',                                  # ConstantODEblock
  'block_transformer_attention': 'gnpde_amd.block_transformer_attention',        # AttODEblock
  'block_mixed': 'gnpde_amd.block_mixed',                                        # MixedODEblock
  'block_transformer_hard_attention': 'gnpde_amd.block_transformer_hard_attention',   # HardAttODEblock
  'block_transformer_rewiring': 'gnpde_amd.block_transformer_rewiring',          # RewireAttODEblock
  'early_stop_solver': 'gnpde_amd.early_stop_solver',                            # EarlyStopInt, EarlyStopRK4, EarlyStopDopri5, SOLVERS
}
NATIVE_GNN = {'GNN': 'gnpde_amd.GNN'}                                            # GNN, BaseGNN (optional)
MERGED = 'base_classes'
OVERRIDES = ('ODEFunc', 'ODEblock', 'RegularizedODEfunc')                        # what base_classes takes from this package


def _reference_file(name):
  """First `<name>.py` on sys.path that is not part of this package."""
  here = os.path.dirname(os.path.abspath(__file__))
  for p in sys.path:
    cand = os.path.abspath(os.path.join(p or '.', name + '.py'))
    if os.path.isfile(cand) and os.path.dirname(cand) != here:
      return cand
  return None


class _MergedBaseClasses(importlib.abc.MetaPathFinder, importlib.abc.Loader):
  """`import base_classes`: the reference's module with the three hot-path types replaced (see the module docstring)."""

  def find_spec(self, fullname, path=None, target=None):
    if fullname != MERGED:
      return None
    return importlib.util.spec_from_loader(fullname, self)

  def create_module(self, spec):
    return None

  def exec_module(self, module):
    ref_path = _reference_file(MERGED)
    if ref_path is not None:
      spec = importlib.util.spec_from_file_location('_reference_' + MERGED, ref_path)
      ref = importlib.util.module_from_spec(spec)
      sys.modules[spec.name] = ref
      spec.loader.exec_module(ref)
      for name in dir(ref):
        if not name.startswith('__'):
          setattr(module, name, getattr(ref, name))
      module.__file__ = ref_path
    ours = importlib.import_module('gnpde_amd.base_classes')
    for name in OVERRIDES:
      setattr(module, name, getattr(ours, name))
    module.__gnpde_reference__ = ref_path


_finder = _MergedBaseClasses()


def installed():
  return _finder in sys.meta_path


def install(native_gnn=False):
  """Answer the reference's module names with this package (idempotent).  Returns the list of names now served."""
  already = {name: sys.modules[name] for name in list(MODULES) + [MERGED] if name in sys.modules}
  table = dict(MODULES)
  if native_gnn:
    table.update(NATIVE_GNN)
  foreign = [n for n, m in already.items()
             if not getattr(m, '__name__', '').startswith('gnpde_amd') and not hasattr(m, '__gnpde_reference__')]
  if foreign:
    raise ImportError('gnpde_amd.dropin.install(): %s already imported from elsewhere; install the drop-in before the '
                      'first import of a reference module' % ', '.join(sorted(foreign)))
  for name, target in table.items():
    sys.modules[name] = importlib.import_module(target)
  if _finder not in sys.meta_path:
    sys.meta_path.insert(0, _finder)
  return sorted(table) + [MERGED]


def uninstall():
  """Undo install() (tests)."""
  if _finder in sys.meta_path:
    sys.meta_path.remove(_finder)
  for name in list(MODULES) + list(NATIVE_GNN) + [MERGED, '_reference_' + MERGED]:
    m = sys.modules.get(name)
    if isinstance(m, types.ModuleType) and (getattr(m, '__name__', '').startswith('gnpde_amd') or
                                            hasattr(m, '__gnpde_reference__') or name.startswith('_reference_')):
      del sys.modules[name]


def main(argv=None):
  """python -m gnpde_amd.dropin [--native-gnn] SCRIPT [ARGS...]: install(), then run SCRIPT as __main__ (its directory goes
  to the front of sys.path, as `python SCRIPT` would put it)."""
  import runpy
  argv = list(sys.argv[1:] if argv is None else argv)
  native = False
  while argv and argv[0].startswith('--'):
    flag = argv.pop(0)
    if flag == '--native-gnn':
      native = True
    else:
      raise SystemExit('gnpde_amd.dropin: unknown option %s\nusage: python -m gnpde_amd.dropin [--native-gnn] SCRIPT [ARGS...]' % flag)
  if not argv:
    raise SystemExit('usage: python -m gnpde_amd.dropin [--native-gnn] SCRIPT [ARGS...]')
  script = os.path.abspath(argv[0])
  if not os.path.isfile(script):
    raise SystemExit('gnpde_amd.dropin: no such script: %s' % argv[0])
  sys.path.insert(0, os.path.dirname(script))
  install(native_gnn=native)
  sys.argv = [script] + argv[1:]
  runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
  main()
