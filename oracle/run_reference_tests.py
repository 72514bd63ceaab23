"""Run the reference's own dataset-free unit tests over the stand-ins (container-only).

This pins the stand-ins (torch_geometric softmax, torch_sparse spmm, ...) against the known-answer
facts the reference tests hold for this path: exact 0.5 attention on the
complete 3-graph, attention rows summing to 1, head-mean linearity.  Tests that download Cora /
Citeseer in setUp cannot run (no network) and are listed as skipped.
"""
import sys
import os
import unittest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_env  # noqa: E402

DATASET_FREE = [
  'test_transformer_attention.AttentionTests.test',
  'test_transformer_attention.AttentionTests.test_symmetric_attention',
  'test_transformer_attention.AttentionTests.test_head_aggregation',
  'test_attention.AttentionTests.test',
  'test_attention.AttentionTests.test_symetric_attention',
]


def main():
  ref_env.activate()
  suite = unittest.TestSuite()
  for name in DATASET_FREE:
    suite.addTests(unittest.defaultTestLoader.loadTestsFromName(name))
  res = unittest.TextTestRunner(verbosity=1).run(suite)
  return 0 if res.wasSuccessful() else 1


if __name__ == '__main__':
  sys.exit(main())
