"""Drop-in for the reference's src/base_classes.py: ODEFunc / ODEblock come from the MI355X package,
everything else (BaseGNN, the regulariser registry) is re-exported from the reference's own module, which
is looked up further down sys.path (put this directory BEFORE the reference's src/)."""
import importlib.util
import os
import sys

_here = os.path.abspath(__file__)
_ref = None
for _p in sys.path:
  _cand = os.path.join(_p or '.', 'base_classes.py')
  if os.path.isfile(_cand) and os.path.abspath(_cand) != _here:
    _spec = importlib.util.spec_from_file_location('_reference_base_classes', _cand)
    _ref = importlib.util.module_from_spec(_spec)
    sys.modules['_reference_base_classes'] = _ref
    _spec.loader.exec_module(_ref)
    break

if _ref is not None:
  for _name in dir(_ref):
    if not _name.startswith('__'):
      globals()[_name] = getattr(_ref, _name)

from gnpde_amd.base_classes import ODEFunc, ODEblock, RegularizedODEfunc  # noqa: E402,F401
