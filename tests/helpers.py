"""Shared test plumbing: golden-fixture loading, module construction from a fixture, parity metric."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-5  # BASELINE.json north_star: 1e-5 relative fp32 (max-norm and 2-norm relative)


class Fixture(object):
  def __init__(self, name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    self.name = name
    self.opt = json.loads(bytes(z['opt_json']).decode())
    self.arr = {k: z[k] for k in z.files if k != 'opt_json' and not k.startswith('param/')}
    self.params = {k[len('param/'):]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith('param/')}

  def t(self, key, device='cpu'):
    return torch.from_numpy(np.array(self.arr[key])).to(device)


def fixtures(prefix):
  return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


class Data(object):
  """What the reference's model touches on `dataset.data`."""

  def __init__(self, x, edge_index, edge_attr=None):
    self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr
    self.num_nodes = x.shape[0]
    self.num_features = x.shape[1]


def parity(a, b):
  a = a.detach().double().cpu()
  b = b.detach().double().cpu()
  e_inf = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
  e_2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
  return e_inf, e_2


def assert_parity(a, b, tol=TOL, what=''):
  assert a.shape == b.shape, '%s: shape %s vs %s' % (what, tuple(a.shape), tuple(b.shape))
  assert torch.isfinite(a).all(), '%s: non-finite output' % what
  e_inf, e_2 = parity(a, b)
  assert e_inf <= tol and e_2 <= tol, '%s: rel max err %.3e, rel l2 err %.3e (tol %.1e)' % (what, e_inf, e_2, tol)


def random_graph(n, avg_deg, seed, hubs=0, hub_deg=0, loops=True, isolated=0, dup=0):
  """Directed edge list [2,E] in shuffled order with optional hub rows (degree > GNPDE_LONG_ROW),
  isolated nodes (empty rows) and duplicate edges."""
  g = torch.Generator().manual_seed(seed)
  m = n * avg_deg
  lim = n - isolated
  row = torch.randint(0, lim, (m,), generator=g)
  col = torch.randint(0, lim, (m,), generator=g)
  parts = [torch.stack([row, col])]
  for hb in range(hubs):
    nb = torch.randint(0, lim, (hub_deg,), generator=g)
    hub = torch.full((hub_deg,), hb * 3 % max(lim, 1))
    parts.append(torch.stack([hub, nb]))
    parts.append(torch.stack([nb, hub]))
  if loops:
    l = torch.arange(lim)
    parts.append(torch.stack([l, l]))
  ei = torch.cat(parts, dim=1)
  if dup:
    ei = torch.cat([ei, ei[:, :dup]], dim=1)
  ei = ei[:, torch.randperm(ei.size(1), generator=g)]
  return ei.long()
