"""Put the reference (/root/reference/src, read-only) on sys.path over the third-party stand-ins.

TEST INFRASTRUCTURE, container-only: /root/reference does not exist on the GPU box, so nothing
that runs there (pytest -m gpu, smoke(), bench.py) may import this module.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get('GNPDE_REFERENCE_ROOT', '/root/reference')


def available():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, 'src'))


def activate():
  if not available():
    raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)
  here = os.path.dirname(os.path.abspath(__file__))
  if here not in sys.path:
    sys.path.insert(0, here)
  from shims import install as _install
  _install.install()
  for sub in ('src', 'test'):
    p = os.path.join(REFERENCE_ROOT, sub)
    if p not in sys.path:
      sys.path.append(p)
