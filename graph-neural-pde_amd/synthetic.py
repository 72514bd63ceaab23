"""Synthetic graphs of the shapes BASELINE.json names: there is no network
for Cora / ogbn-arxiv, so benchmarks and parity tests use seeded graphs with the same node count,
edge count and degree skew, fed through the same preparation as the reference's data.py
(to_undirected + coalesce, then self-loops / rw normalisation inside the blocks)."""
import numpy as np
import torch

CONFIGS = {
  # name: (nodes, undirected pairs before symmetrisation, state width d, attention dim A, heads, kind)
  'cora': dict(n=2708, pairs=5278, d=80, att_dim=128, heads=8, kind='uniform'),
  'arxiv': dict(n=169343, pairs=1157799, d=128, att_dim=16, heads=4, kind='powerlaw'),
  'rmat': dict(n=2 ** 21, pairs=40_000_000, d=256, att_dim=64, heads=4, kind='rmat'),
}


def _symmetrise(a, b, n):
  """Both directions of every distinct non-loop pair, sorted by (row, col) like PyG's coalesce."""
  keep = a != b
  a, b = a[keep], b[keep]
  lo, hi = np.minimum(a, b), np.maximum(a, b)
  key = np.unique(lo.astype(np.int64) * n + hi)
  lo, hi = key // n, key % n
  row = np.concatenate([lo, hi])
  col = np.concatenate([hi, lo])
  order = np.lexsort((col, row))
  return torch.from_numpy(np.stack([row[order], col[order]])).long()


def uniform_graph(n, pairs, seed=0):
  rng = np.random.default_rng(seed)
  return _symmetrise(rng.integers(0, n, pairs), rng.integers(0, n, pairs), n)


def powerlaw_graph(n, pairs, seed=0, exponent=0.75, hub_degree=13000):
  """Citation-like skew: endpoints drawn with probability ~ rank^-exponent, node ids shuffled (no
  locality in the labelling), a few hubs of ogbn-arxiv's maximum degree (~13k)."""
  rng = np.random.default_rng(seed)
  p = np.arange(1, n + 1, dtype=np.float64) ** (-exponent)
  p /= p.sum()
  cdf = np.cumsum(p)
  a = np.searchsorted(cdf, rng.random(pairs))
  b = rng.integers(0, n, pairs)
  a = np.minimum(a, n - 1)
  relabel = rng.permutation(n)
  ei = _symmetrise(relabel[a], relabel[b], n)
  return ei


def rmat_graph(scale, edges, seed=0, a=0.57, b=0.19, c=0.19):
  """R-MAT (Chakrabarti et al.) with the Graph500 parameters; returns symmetrised, de-duplicated edges."""
  rng = np.random.default_rng(seed)
  n = 1 << scale
  src = np.zeros(edges, dtype=np.int64)
  dst = np.zeros(edges, dtype=np.int64)
  for bit in range(scale):
    r = rng.random(edges)
    src_bit = r >= (a + b)
    dst_bit = ((r >= a) & (r < a + b)) | (r >= a + b + c)
    src |= src_bit.astype(np.int64) << bit
    dst |= dst_bit.astype(np.int64) << bit
  return _symmetrise(src, dst, n)


def make_graph(name, seed=0, scale=1.0):
  """edge_index [2,E] int64 (symmetric, no self-loops, sorted) for a named config; `scale` < 1
  shrinks nodes and edges proportionally (used by tests)."""
  cfg = CONFIGS[name]
  n = max(int(cfg['n'] * scale), 8)
  pairs = max(int(cfg['pairs'] * scale), 8)
  if cfg['kind'] == 'uniform':
    return uniform_graph(n, pairs, seed), n
  if cfg['kind'] == 'powerlaw':
    return powerlaw_graph(n, pairs, seed), n
  sc = max(int(round(np.log2(n))), 3)
  return rmat_graph(sc, pairs, seed), 1 << sc
