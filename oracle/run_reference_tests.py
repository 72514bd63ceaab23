"""Run the reference's own unit tests (/root/reference/test) over the stand-ins (container-only).

This pins the stand-ins (torch_geometric softmax, torch_sparse spmm, the restated torchdiffeq incl. the pieces
early_stop_solver.py reaches into, ...) against the known-answer facts the reference's tests hold for this path
(SURVEY.md section 8c): exact 0.5 attention on the complete 3-graph, attention rows / columns summing to one,
head-mean linearity, rw / gcn normalisation values, block and GNN forwards producing finite outputs of the right
shape, the early-stopping integrators running end to end.  The tests' setUp downloads Cora / Citeseer through
`data.get_dataset`; there is no network here, so that one function is replaced by a seeded synthetic dataset of the
same interface (x, edge_index, y, masks, num_features, num_classes) -- the assertions are dataset independent.
"""
import sys
import os
import unittest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_env  # noqa: E402

MODULES = ['test_transformer_attention', 'test_attention', 'test_utils', 'test_function_laplacian_diffusion',
           'test_block_mixed', 'test_attention_ode_block', 'test_gnn', 'test_early_stop']


def synthetic_dataset():
  import torch
  from torch_geometric.data import Data

  class Dataset(object):
    def __init__(self, data, num_classes):
      self.data, self.num_classes = data, num_classes
      self.num_features = data.num_features
      self.num_node_features = data.num_features

    def __getitem__(self, i):
      return self.data

  g = torch.Generator().manual_seed(0)
  n, f, c = 300, 50, 7
  a = torch.randint(0, n, (900,), generator=g)
  b = torch.randint(0, n, (900,), generator=g)
  keep = a != b
  ei = torch.unique(torch.cat([torch.stack([a[keep], b[keep]]), torch.stack([b[keep], a[keep]])], 1), dim=1)
  role = torch.randperm(n, generator=g)
  data = Data(x=torch.rand(n, f, generator=g), edge_index=ei, y=torch.randint(0, c, (n,), generator=g),
              train_mask=role < 100, val_mask=(role >= 100) & (role < 200), test_mask=role >= 200)
  return Dataset(data, c)


def main():
  ref_env.activate()
  import data as reference_data
  reference_data.get_dataset = lambda opt, data_dir, use_lcc=False: synthetic_dataset()
  os.chdir(os.path.join(ref_env.REFERENCE_ROOT, 'test'))
  suite = unittest.TestSuite()
  for name in MODULES:
    suite.addTests(unittest.defaultTestLoader.loadTestsFromName(name))
  res = unittest.TextTestRunner(verbosity=1).run(suite)
  print('reference tests: ran %d, failures %d, errors %d' % (res.testsRun, len(res.failures), len(res.errors)))
  return 0 if res.wasSuccessful() else 1


if __name__ == '__main__':
  sys.exit(main())
