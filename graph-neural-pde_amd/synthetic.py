"""Synthetic graphs of the shapes BASELINE.json names: there is no network
for Cora / ogbn-arxiv, so benchmarks and parity tests use seeded graphs with the same node count,
edge count and degree skew, fed through the same preparation as the reference's data.py
(to_undirected + coalesce, then self-loops / rw normalisation inside the blocks)."""
import numpy as np
import torch

CONFIGS = {
  # name: (nodes, undirected pairs before symmetrisation, state width d, attention dim A, heads, kind)
  'cora': dict(n=2708, pairs=5278, d=80, att_dim=128, heads=8, kind='uniform'),
  'arxiv': dict(n=169343, pairs=1260000, d=128, att_dim=16, heads=4, kind='community'),
  'arxiv_flat': dict(n=169343, pairs=1157799, d=128, att_dim=16, heads=4, kind='powerlaw'),
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


def community_powerlaw_graph(n, pairs, seed=0, exponent=0.75, n_comm=40, mixing=0.35, comm_exponent=1.0):
  """Power-law degrees AND community structure, node ids shuffled.

  ogbn-arxiv is a citation graph with 40 subject classes and an edge homophily of about 0.65 (two thirds
  of the citations stay inside a class); a structure-free preferential graph has none of that and is the
  worst case for caches and for partitioning.  Here every node belongs to one of `n_comm` communities with
  Zipf-distributed sizes; one endpoint of each edge is drawn by degree propensity (rank^-exponent), the
  other from the SAME community with probability 1 - mixing (again by propensity), else from the whole
  graph.  Ids are shuffled afterwards, so any locality has to be found by the partitioner."""
  rng = np.random.default_rng(seed)
  sizes = np.arange(1, n_comm + 1, dtype=np.float64) ** (-comm_exponent)
  comm = np.sort(rng.choice(n_comm, size=n, p=sizes / sizes.sum()))     # node k (community-sorted) -> community
  w = rng.permutation(np.arange(1, n + 1, dtype=np.float64) ** (-exponent))
  cdf = np.cumsum(w)
  total = cdf[-1]
  a = np.minimum(np.searchsorted(cdf, rng.random(pairs) * total), n - 1)
  # community ranges in the cdf (nodes are laid out community by community)
  start = np.searchsorted(comm, np.arange(n_comm), side='left')
  end = np.searchsorted(comm, np.arange(n_comm), side='right')
  lo = np.where(start > 0, cdf[np.maximum(start - 1, 0)], 0.0)
  hi = cdf[np.maximum(end - 1, 0)]
  ca = comm[a]
  inside = rng.random(pairs) >= mixing
  u = rng.random(pairs)
  target = np.where(inside, lo[ca] + u * (hi[ca] - lo[ca]), u * total)
  b = np.minimum(np.searchsorted(cdf, target), n - 1)
  relabel = rng.permutation(n)
  return _symmetrise(relabel[a], relabel[b], n), relabel, comm


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
  if cfg['kind'] == 'community':
    return community_powerlaw_graph(n, pairs, seed)[0], n
  sc = max(int(round(np.log2(n))), 3)
  return rmat_graph(sc, pairs, seed), 1 << sc
