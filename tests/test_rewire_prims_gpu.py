"""Native edge-set bookkeeping of the hard-attention / rewiring blocks (csrc/rewire.hip): radix-select quantile against
torch.quantile (bit for bit: the kept edge set of a training forward depends on it), stable threshold compaction and the
per-endpoint renormalisation against the reference's op sequence (mask, boolean indexing, scatter-add)."""
import numpy as np
import pytest
import torch

from gnpde_amd import ops
from oracle import restate as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [1, 2, 5, 257, 100_003, 2_480_741])
@pytest.mark.parametrize('q', [0.0, 0.19, 0.5, 0.81, 0.999, 1.0])
def test_quantile_equals_torch_quantile(dev, n, q):
  g = torch.Generator().manual_seed(n)
  v = torch.randn(n, generator=g)
  v[::3] = v[::3].abs() * 1e-3            # attention-like small positive values, plus duplicates below
  if n > 10:
    v[5:9] = v[4]
  vd = v.to(dev)
  got = ops.quantile(vd, q)
  ref = torch.quantile(vd, q)
  assert got.shape == ref.shape == ()
  assert torch.equal(got, ref), (n, q, float(got), float(ref))


def test_quantile_beyond_torch_size_limit(dev):
  """torch.quantile refuses more than 16 M elements (the R-MAT edge count is 77 M); the radix select does not."""
  n = 20_000_000
  v = torch.rand(n, generator=torch.Generator().manual_seed(1)).to(dev)
  with pytest.raises(RuntimeError):
    torch.quantile(v, 0.3)
  got = float(ops.quantile(v, 0.3))
  ref = float(np.quantile(v.cpu().numpy().astype(np.float64), 0.3))
  assert abs(got - ref) < 1e-6


@pytest.mark.parametrize('norm_idx', [0, 1])
@pytest.mark.parametrize('n_edges', [0, 1, 4097, 300_000])
def test_threshold_edges_matches_reference_sequence(dev, norm_idx, n_edges):
  n = 5000
  g = torch.Generator().manual_seed(7 + n_edges)
  ei = torch.randint(0, n, (2, n_edges), generator=g).to(dev)
  score = torch.rand(n_edges, generator=g).to(dev)
  thr = torch.tensor(0.37, device=dev)
  kept, w = ops.threshold_edges(ei, score, thr, norm_idx, n)
  mask = score > thr
  ref_ei = ei[:, mask]
  s = score[mask]
  idx = ref_ei[norm_idx]
  sums = torch.zeros(n, device=dev).index_add_(0, idx, s)
  ref_w = s / (sums[idx] + 1e-16)
  assert kept.is_contiguous() and torch.equal(kept, ref_ei), 'kept edges (and their order) must equal edge_index[:, mask]'
  assert w.shape == ref_w.shape
  if n_edges:
    assert torch.allclose(w, ref_w, rtol=2e-6, atol=0)
    tot = torch.zeros(n, device=dev).index_add_(0, idx, w)
    assert torch.allclose(tot[tot > 0], torch.ones_like(tot[tot > 0]), atol=1e-5)


def test_threshold_from_quantile_keeps_the_requested_share(dev):
  E = 1_000_000
  score = torch.rand(E, generator=torch.Generator().manual_seed(3)).to(dev) ** 3
  ei = torch.randint(0, 1000, (2, E), generator=torch.Generator().manual_seed(4)).to(dev)
  thr = ops.quantile(score, 1 - 0.81)
  kept, w = ops.threshold_edges(ei, score, thr, 0, 1000)
  assert abs(kept.shape[1] / E - 0.81) < 1e-3
  assert kept.shape[1] == int((score > torch.quantile(score, 1 - 0.81)).sum())


@pytest.mark.parametrize('n,deg,hub', [(1, 1, 0), (7, 2, 0), (300, 4, 150), (2049, 6, 900), (5000, 3, 0)])
def test_two_hop_matches_reference_sequence(dev, n, deg, hub):
  g = torch.Generator().manual_seed(n + deg)
  e = n * deg
  ei = torch.randint(0, n, (2, e), generator=g)
  if hub:                                                    # a hub row and a hub column, duplicates, self loops
    extra = torch.randint(0, n, (hub,), generator=g)
    ei = torch.cat([ei, torch.stack([torch.zeros(hub, dtype=torch.long), extra]),
                    torch.stack([extra, torch.full((hub,), 3)]), ei[:, :5], torch.tensor([[2, 5], [2, 5]])], dim=1)
  w = torch.rand(ei.shape[1], generator=g)
  w[::17] = 0.0                                              # explicit zeros keep their entry
  from gnpde_amd.graph import CSRGraph
  graph = CSRGraph(ei.to(dev), n)
  got_ei, got_w = ops.two_hop(graph, w.to(dev))
  ref_ei, ref_w = R.two_hop(ei, w, n)          # the reference's op sequence in dense float64 (oracle/restate.py)
  assert got_ei.dtype == torch.int64 and got_ei.shape == ref_ei.shape, (got_ei.shape, ref_ei.shape)
  assert torch.equal(got_ei.cpu(), ref_ei)                   # same entries in coalesce's (row, col) order
  assert torch.allclose(got_w.cpu().double(), ref_w, rtol=2e-6, atol=1e-7)
  again_ei, again_w = ops.two_hop(graph, w.to(dev))
  assert torch.equal(again_ei, got_ei) and torch.equal(again_w, got_w), 'not run-to-run identical'


def test_two_hop_agrees_with_the_expanded_list_path(dev):
  """At a size the dense check cannot reach: against the test-side torch composite (tests/sparse_composite.py) (expand every product, sort, index_add)."""
  import sparse_composite as B
  from gnpde_amd import synthetic
  from gnpde_amd.graph import CSRGraph
  n = 40_000
  ei = synthetic.powerlaw_graph(n, 4 * n, seed=5, hub_degree=3000).to(dev)    # symmetric, skewed degrees, shuffled ids
  w = torch.rand(ei.shape[1], generator=torch.Generator().manual_seed(6)).to(dev)
  got_ei, got_w = ops.two_hop(CSRGraph(ei, n), w)
  new_ei, new_w = B._spspmm(ei, w, ei, w, n)
  keep = new_ei[0] != new_ei[1]
  ref_ei, ref_w = B._coalesce(torch.cat([ei, new_ei[:, keep]], dim=1), torch.cat([w, new_w[keep]]) / 2, n)
  assert torch.equal(got_ei, ref_ei)
  assert torch.allclose(got_w, ref_w, rtol=1e-5, atol=1e-7)
  assert got_ei.shape[1] > 20 * ei.shape[1]                  # a real densification
