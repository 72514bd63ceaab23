"""BASELINE configs[4] shape (R-MAT, d = 256, attention_dim 64 / 4 heads) on one GPU: the state (1 GiB at half scale)
is far larger than the 256 MiB Infinity Cache and the graph is hub-heavy, so this is the HBM-bound regime -- and it runs
the d = 256 instantiations of every kernel (aggregation with 1-KB rows, row attention with d_k = 16, projection with
m = 128 outputs) that the ogbn-arxiv-shaped tests never reach.

The oracle cannot evaluate the whole graph in seconds ([E,d] temporaries of tens of GB), but f is row-local once the
neighbours are known: rows of a SUBSET (the largest hubs, mid-degree rows, random rows) are checked against
oracle/restate.py evaluated on the sub-graph induced by those rows and all their neighbours.
"""
import pytest
import torch

import gnpde_amd as G
from oracle import restate as R
from helpers import Data, assert_parity

pytestmark = pytest.mark.gpu

OPT = dict(heads=4, attention_dim=64, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False,
           reweight_attention=False, beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9,
           add_source=True, no_alpha_sigmoid=False, mix_features=False, hidden_dim=256, augment=False, adjoint=False,
           tol_scale=1.0, data_norm='rw', method='rk4', step_size=1.0, max_iters=100, block='constant',
           function='transformer', time=1.0)


def _build(dev, scale):
  ei, n = G.synthetic.make_graph('rmat', seed=1, scale=scale)     # 0.5: 2^20 nodes, ~38 M edges; 1.0: BASELINE configs[4]
  d = 256
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(5))
  xd = x.to(dev)
  block = G.ConstantODEblock(G.ODEFuncTransformerAtt, [], OPT, Data(xd, ei.to(dev)), dev, t=torch.tensor([0, 1.0])).to(dev)
  g = torch.Generator().manual_seed(2)
  with torch.no_grad():
    for name, p in block.named_parameters():
      if p.dim() >= 2 and 'multihead_att_layer' in name:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      elif name.endswith('.bias') and 'multihead_att_layer' in name:
        p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
    for f in (block.odefunc, block.reg_odefunc.odefunc):
      f.alpha_train.fill_(0.3)
      f.beta_train.fill_(0.2)
  block.eval()
  block.set_x0(xd)
  return block, x, xd, n


@pytest.fixture(scope='module')
def rmat(dev):
  return _build(dev, 0.5)


def _subset_oracle(block, x, rows):
  """f restricted to `rows`, from the oracle on the sub-graph (rows + all their neighbours)."""
  f = block.odefunc
  lay = f.multihead_att_layer
  cpu = lambda t: t.detach().cpu()
  edge = cpu(f.edge_index)
  n = x.shape[0]
  pick = torch.zeros(n, dtype=torch.bool)
  pick[rows] = True
  keep = pick[edge[0]]
  r, c = edge[0][keep], edge[1][keep]
  nodes = torch.unique(torch.cat([rows, c]))
  rl, cl = torch.searchsorted(nodes, r), torch.searchsorted(nodes, c)
  sub_edge = torch.stack([rl, cl])
  xs = x[nodes]
  out = R.rhs_transformer(xs, sub_edge, cpu(lay.Q.weight), cpu(lay.Q.bias), cpu(lay.K.weight), cpu(lay.K.bias), lay.h,
                          cpu(f.alpha_train), cpu(f.beta_train), xs, False, True)
  return out[torch.searchsorted(nodes, rows)], int(keep.sum())


def test_rmat_d256_rows_against_oracle(rmat):
  _rows_against_oracle(rmat, 2 ** 30, 1000)


def test_rmat_full_scale_rows_against_oracle(dev):
  """The SAME check on the graph BASELINE configs[4] names, at full size: 2^21 nodes, 77 M entries, 2-GiB state -- the graph
  `bench.py --graph rmat` times: hashed XCD row deal with 128-row blocks, 27.9 k hub rows, the 227-chunk hub."""
  import gc
  built = _build(dev, 1.0)
  try:
    block, x, xd, n = built
    assert n == 2 ** 21
    graph = block.odefunc._graph(xd)
    assert graph.struct.xcd_deal == 1, 'the R-MAT row lengths depend on the row id: the builder must pick hashed blocks'
    assert graph.max_row_len > 100 * 512, 'expected a hub of more than 100 chunks (got %d entries)' % graph.max_row_len
    _rows_against_oracle(built, 2 ** 31, 20000)
  finally:
    del built
    gc.collect()
    torch.cuda.empty_cache()


def _rows_against_oracle(rmat, min_state_bytes, min_long_rows):
  block, x, xd, n = rmat
  f = block.odefunc
  graph = f._graph(xd)
  assert n * 256 * 4 >= min_state_bytes, 'state must exceed the Infinity Cache by a wide margin'
  assert graph.n_long_rows > min_long_rows, 'the R-MAT graph should be hub-heavy (got %d long rows)' % graph.n_long_rows
  deg = torch.bincount(f.edge_index[0].cpu(), minlength=n)
  g = torch.Generator().manual_seed(9)
  hubs = torch.topk(deg, 6).indices                                   # many 512-entry chunks each
  long_small = torch.nonzero((deg > 512) & (deg <= 1100)).flatten()[:40]   # two / three chunks
  mid = torch.nonzero((deg > 16) & (deg <= 512)).flatten()
  mid = mid[torch.randperm(mid.numel(), generator=g)[:400]]
  rnd = torch.randperm(n, generator=g)[:3000]                          # mostly short rows (<= 16 entries)
  rows = torch.unique(torch.cat([hubs, long_small, mid, rnd]))
  with torch.no_grad():
    got = f(0.0, xd)
  assert torch.isfinite(got).all()
  ref, n_edges = _subset_oracle(block, x, rows)
  assert n_edges > 200_000
  assert_parity(got[rows.to(got.device)], ref, what='R-MAT d=256: %d rows (%d entries) incl. the 6 largest hubs' % (rows.numel(), n_edges))
  # class by class, so that a defect confined to one degree class cannot hide in the norm of the others
  for name, sel in (('hub rows', hubs), ('2-3 chunk rows', long_small), ('17..512-entry rows', mid)):
    pos = torch.searchsorted(rows, sel)
    assert_parity(got[sel.to(got.device)], ref[pos], what='R-MAT d=256, ' + name)


def test_rmat_d256_one_rk4_step_is_the_composition_of_its_stages(rmat):
  """The solver's fused stage epilogues at this instantiation: one captured rk4 step == the 3/8-rule combination of four
  plain evaluations (device arithmetic in torchdiffeq's operation order)."""
  block, x, xd, n = rmat
  f = block.odefunc
  with torch.no_grad():
    z = block(xd)
    k1 = f(0.0, xd)
    k2 = f(0.0, xd + k1 / 3)
    k3 = f(0.0, xd + (k2 - k1 / 3))
    k4 = f(0.0, xd + (k1 - k2 + k3))
    ref = xd + (k1 + 3 * (k2 + k3) + k4) * 0.125
  assert_parity(z, ref, what='R-MAT d=256 one rk4 step')


def test_rmat_d256_constants_are_stationary(rmat):
  block, x, xd, n = rmat
  f = block.odefunc
  with torch.no_grad():
    c = torch.full_like(xd, 0.7)
    x0_saved = f.x0
    f.x0 = torch.zeros_like(xd)
    out = f(0.0, c)
    f.x0 = x0_saved
  assert out.abs().max().item() < 5e-6
