"""Parity of every C-ABI kernel with the CPU oracle (oracle/restate.py) on seeded inputs."""
import math

import pytest
import torch

import gnpde_amd as G
from gnpde_amd import ops, _lib
from oracle import restate as R
from helpers import assert_parity, random_graph, Data

pytestmark = pytest.mark.gpu


def _weights(e, seed):
  g = torch.Generator().manual_seed(seed)
  return torch.rand(e, generator=g) * 0.3 + 0.01


@pytest.mark.parametrize('d', [1, 7, 24, 80, 128, 162, 256, 300, 520])
@pytest.mark.parametrize('source', [False, True])
def test_spmm_rhs_widths(dev, d, source):
  n = 500
  ei = random_graph(n, 6, seed=d, isolated=5, dup=40)
  w = _weights(ei.size(1), d + 1)
  g = torch.Generator().manual_seed(d + 2)
  x = torch.randn(n, d, generator=g)
  x0 = torch.randn(n, d, generator=g) if source else None
  alpha, beta = torch.tensor(0.3), torch.tensor(-0.7)
  ref = R.rhs_laplacian(x, ei, w, alpha, beta, x0, no_alpha_sigmoid=False, add_source=source)
  graph = G.CSRGraph(ei.to(dev), n)
  w_csr = ops.edge_to_csr_mean(graph, w.to(dev))
  out = ops.spmm_rhs(graph, w_csr, x.to(dev), alpha.to(dev), beta.to(dev), None if x0 is None else x0.to(dev), True)
  assert_parity(out, ref, what='spmm_rhs d=%d' % d)
  plain = ops.spmm(graph, w_csr, x.to(dev))
  assert_parity(plain, R.spmm(ei, w, n, x), what='spmm d=%d' % d)


@pytest.mark.parametrize('d', [32, 128, 162])
def test_spmm_rhs_hub_rows(dev, d):
  """Rows longer than GNPDE_LONG_ROW go through the chunk + reduce path."""
  n = 3000
  ei = random_graph(n, 4, seed=7, hubs=3, hub_deg=2500)
  graph = G.CSRGraph(ei.to(dev), n)
  assert graph.n_long_rows >= 3 and graph.n_long_chunks >= 3 * 5
  w = _weights(ei.size(1), 3)
  g = torch.Generator().manual_seed(5)
  x, x0 = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
  alpha, beta = torch.tensor(-0.2), torch.tensor(0.4)
  ref = R.rhs_laplacian(x, ei, w, alpha, beta, x0, no_alpha_sigmoid=True, add_source=True)
  out = ops.spmm_rhs(graph, ops.edge_to_csr_mean(graph, w.to(dev)), x.to(dev), alpha.to(dev), beta.to(dev), x0.to(dev), False)
  assert_parity(out, ref, what='hub rows d=%d' % d)


def test_spmm_unaligned_view(dev):
  """A leading dimension that breaks 16-byte alignment falls back to narrower loads."""
  n, d = 300, 30
  ei = random_graph(n, 5, seed=11)
  w = _weights(ei.size(1), 1)
  big = torch.randn(n, d + 1, generator=torch.Generator().manual_seed(2))
  x = big[:, :d]
  ref = R.spmm(ei, w, n, x.contiguous())
  graph = G.CSRGraph(ei.to(dev), n)
  xd = big.to(dev)[:, :d]
  out = torch.empty(n, d + 1, device=dev)[:, :d]
  L = _lib.lib()
  _lib.check(L.gnpde_spmm(graph.ref(), _lib.ptr(ops.edge_to_csr_mean(graph, w.to(dev))), _lib.ptr(xd), d, d + 1,
                          _lib.ptr(out), None, 0, _lib.stream_of(xd)))
  # both operands share ld = d+1
  assert_parity(out, ref, what='unaligned ld')


@pytest.mark.parametrize('n,d,m', [(1, 8, 16), (37, 24, 32), (1000, 80, 256), (513, 128, 32), (300, 162, 64),
                                   (129, 30, 20), (64, 5, 3), (2050, 128, 272),
                                   # wide projections: W staged in LDS (linear_lds_kernel<8,256>, <4,256>, <8,128>, <4,128>)
                                   (5003, 256, 128), (3000, 192, 64), (4100, 128, 128), (777, 64, 64), (9, 256, 192)])
def test_linear(dev, n, d, m):
  g = torch.Generator().manual_seed(n + d + m)
  x = torch.randn(n, d, generator=g)
  W = torch.randn(m, d, generator=g) / math.sqrt(d)
  b = torch.randn(m, generator=g)
  ref = torch.nn.functional.linear(x.double(), W.double(), b.double()).float()
  out = ops.linear(x.to(dev), W.to(dev), b.to(dev))
  assert_parity(out, ref, what='linear')
  out2 = ops.linear(x.to(dev), W.to(dev), None)
  assert_parity(out2, torch.nn.functional.linear(x.double(), W.double()).float(), what='linear no bias')


@pytest.mark.parametrize('n,d,m', [(1000, 128, 40), (169, 24, 5), (5003, 256, 128), (777, 64, 64), (64, 5, 3), (513, 81, 40)])
def test_relu_linear(dev, n, d, m):
  """Decoder of GNN.forward: relu on the A operand of every projection kernel; also from a row-strided input (the left
  half of an augmented state, GNN.py:58-59)."""
  g = torch.Generator().manual_seed(n + d + m + 1)
  x = torch.randn(n, 2 * d, generator=g)
  W = torch.randn(m, d, generator=g) / math.sqrt(d)
  b = torch.randn(m, generator=g)
  xd = x.to(dev)
  ref = torch.nn.functional.linear(torch.relu(x[:, :d]).double(), W.double(), b.double()).float()
  assert_parity(ops.linear(xd[:, :d].contiguous(), W.to(dev), b.to(dev), relu_input=True), ref, what='relu_linear')
  assert_parity(ops.linear(xd[:, :d], W.to(dev), b.to(dev), relu_input=True), ref, what='relu_linear strided')
  # the plain entry is unchanged by the flag's plumbing
  ref0 = torch.nn.functional.linear(x[:, :d].double(), W.double(), b.double()).float()
  assert_parity(ops.linear(xd[:, :d], W.to(dev), b.to(dev)), ref0, what='linear strided')


ATT_CASES = [
  ('scaled_dot', 4, 16, 0, False), ('scaled_dot', 4, 16, 1, False), ('scaled_dot', 8, 128, 1, True),
  ('scaled_dot', 8, 128, 0, True), ('scaled_dot', 3, 21, 0, False), ('scaled_dot', 1, 24, 1, False),
  ('cosine_sim', 4, 16, 0, False), ('cosine_sim', 2, 10, 1, False), ('pearson', 4, 32, 1, False),
  ('pearson', 4, 16, 0, True), ('exp_kernel', 4, 16, 0, False), ('exp_kernel', 2, 12, 1, False),
]


@pytest.mark.parametrize('att_type,h,A,norm_idx,sqp', ATT_CASES)
@pytest.mark.parametrize('reweight', [False, True])
def test_edge_attention(dev, att_type, h, A, norm_idx, sqp, reweight):
  n, d = 400, 20
  ei = random_graph(n, 5, seed=h + A, hubs=1, hub_deg=700, isolated=3, dup=25)
  g = torch.Generator().manual_seed(A)
  x = torch.randn(n, d, generator=g)
  Wq, Wk = torch.randn(A, d, generator=g) / math.sqrt(d), torch.randn(A, d, generator=g) / math.sqrt(d)
  bq, bk = torch.randn(A, generator=g) * 0.1, torch.randn(A, generator=g) * 0.1
  ew = torch.rand(ei.size(1), generator=g) + 0.5 if reweight else None
  ov, ls = torch.tensor([1.2]), torch.tensor([0.8])
  att_ref, prods_ref = R.transformer_attention(x, ei, Wq, bq, Wk, bk, h, attention_type=att_type, norm_idx=norm_idx,
                                               square_plus=sqp, edge_weights=ew, reweight=reweight, output_var=ov,
                                               lengthscale=ls)
  graph = G.CSRGraph(ei.to(dev), n)
  qk = ops.linear(x.to(dev), torch.cat([Wq, Wk]).to(dev), torch.cat([bq, bk]).to(dev))
  ew_csr = ops.edge_to_csr_mean(graph, ew.to(dev)) if reweight else None
  st = ops.attention_struct(_lib.ATT_TYPES[att_type], h, A, norm_idx, sqp, q=qk, k=qk[:, A:], ldqk=2 * A,
                            output_var=ov.to(dev), lengthscale=ls.to(dev), edge_w_csr=ew_csr)
  w, att, prods = ops.edge_attention(graph, st, True, True, True, like=qk)
  assert_parity(prods, prods_ref, what='prods')
  assert_parity(att, att_ref, what='attention')
  w_ref = att_ref.mean(dim=1)
  assert_parity(w[:graph.e], w_ref[graph.perm_long.cpu()], what='head-mean weights (CSR order)')


FUSED_CASES = [('scaled_dot', 4, 16), ('scaled_dot', 8, 128), ('scaled_dot', 1, 24), ('scaled_dot', 2, 8),
               ('cosine_sim', 4, 16), ('pearson', 2, 32), ('exp_kernel', 4, 16), ('scaled_dot', 3, 12), ('scaled_dot', 4, 8)]


@pytest.mark.parametrize('att_type,h,A', FUSED_CASES)
@pytest.mark.parametrize('reweight', [False, True])
def test_row_attention_fused_path(dev, att_type, h, A, reweight):
  """Only the head-mean weights requested + softmax over rows -> the fused row kernels (16-lane groups,
  whole-wave rows, hub chunks); (3,12) and (4,8) fall back to the general passes (d_k % 4 != 0 / h = 3)."""
  n, d = 3000, 20
  ei = random_graph(n, 9, seed=h * 7 + A, hubs=2, hub_deg=1300, isolated=3, dup=25)
  extra = random_graph(40, 300, seed=1, loops=False)  # 40 rows with ~300 entries: multi-pass whole-wave rows
  ei = torch.cat([ei, extra], dim=1)
  g = torch.Generator().manual_seed(A)
  x = torch.randn(n, d, generator=g)
  Wq, Wk = torch.randn(A, d, generator=g) / math.sqrt(d), torch.randn(A, d, generator=g) / math.sqrt(d)
  bq, bk = torch.randn(A, generator=g) * 0.1, torch.randn(A, generator=g) * 0.1
  ew = torch.rand(ei.size(1), generator=g) + 0.5 if reweight else None
  ov, ls = torch.tensor([1.2]), torch.tensor([0.8])
  att_ref, _ = R.transformer_attention(x, ei, Wq, bq, Wk, bk, h, attention_type=att_type, norm_idx=0, square_plus=False,
                                       edge_weights=ew, reweight=reweight, output_var=ov, lengthscale=ls)
  graph = G.CSRGraph(ei.to(dev), n)
  assert graph.n_bin16 > 0 and graph.n_bin64 > 40 and graph.n_long_rows >= 2
  qk = ops.linear(x.to(dev), torch.cat([Wq, Wk]).to(dev), torch.cat([bq, bk]).to(dev))
  ew_csr = ops.edge_to_csr_mean(graph, ew.to(dev)) if reweight else None
  st = ops.attention_struct(_lib.ATT_TYPES[att_type], h, A, 0, False, q=qk, k=qk[:, A:], ldqk=2 * A,
                            output_var=ov.to(dev), lengthscale=ls.to(dev), edge_w_csr=ew_csr)
  w, _, _ = ops.edge_attention(graph, st, True, False, False, like=qk)
  assert_parity(w[:graph.e], att_ref.mean(dim=1)[graph.perm_long.cpu()], what='fused head-mean weights')
  # the general path must give the same numbers up to rounding
  w2, _, _ = ops.edge_attention(graph, st, True, True, False, like=qk)
  assert_parity(w2[:graph.e], w[:graph.e], tol=1e-5, what="fused vs general path")


@pytest.mark.parametrize('h,A,norm_idx,slope', [(4, 16, 0, 0.2), (2, 16, 1, 0.05), (3, 9, 0, 0.2)])
def test_gat_attention(dev, h, A, norm_idx, slope):
  n, d = 350, 24
  ei = random_graph(n, 6, seed=A + h, hubs=1, hub_deg=600)
  g = torch.Generator().manual_seed(9)
  x = torch.randn(n, d, generator=g)
  W = torch.randn(d, A, generator=g) / math.sqrt(d)
  a = torch.randn(2 * (A // h), 1, 1, generator=g)
  att_ref, wx_ref = R.gat_attention(x, ei, W, a, h, slope, norm_idx)
  graph = G.CSRGraph(ei.to(dev), n)
  wx = ops.linear(x.to(dev), W.t().contiguous().to(dev))
  assert_parity(wx, wx_ref, what='wx')
  st = ops.attention_struct(_lib.ATT_GAT, h, A, norm_idx, False, q=wx, k=wx, ldqk=A, leaky_slope=slope,
                            gat_a=a.reshape(-1).to(dev))
  _, att, _ = ops.edge_attention(graph, st, False, True, False, like=wx)
  assert_parity(att, att_ref, what='GAT attention')
  if norm_idx == 0:
    w, _, _ = ops.edge_attention(graph, st, True, False, False, like=wx)   # fused row kernel for h in {1,2,4,8}
    assert_parity(w[:graph.e], att_ref.mean(dim=1)[graph.perm_long.cpu()], what='GAT fused head-mean weights')


def test_attention_rows_sum_to_one(dev):
  """Reference test_transformer_attention.py::test_function property at a larger size."""
  n, A, h = 5000, 16, 4
  ei = random_graph(n, 8, seed=3, hubs=2, hub_deg=1500)
  qk = torch.randn(n, 2 * A, device=dev)
  graph = G.CSRGraph(ei.to(dev), n)
  for norm_idx in (0, 1):
    st = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, norm_idx, False, q=qk, k=qk[:, A:], ldqk=2 * A)
    _, att, _ = ops.edge_attention(graph, st, False, True, False, like=qk)
    sums = torch.zeros(n, h, device=dev).index_add_(0, ei[norm_idx].to(dev), att)
    assert torch.all(att > 0) and torch.all(att <= 1 + 1e-6)
    assert torch.allclose(sums, torch.ones_like(sums), atol=1e-4)


def test_symmetric_attention_known_answer(dev):
  """Reference test_symmetric_attention: constant-1e-5 weights, x = 1 on the complete 3-graph -> exactly 0.5."""
  from helpers import Data
  opt = dict(heads=2, attention_dim=32, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False,
             reweight_attention=False, beltrami=False, leaky_relu_slope=0.2)
  layer = G.SpGraphTransAttentionLayer(2, 2, opt, dev).to(dev)
  edge = torch.tensor([[0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1]], device=dev)
  att, _ = layer(torch.ones(3, 2, device=dev), edge)
  assert torch.all(torch.eq(att, 0.5 * torch.ones(6, 2, device=dev)))


def test_degenerate_graphs(dev):
  """Empty edge list, a single self-loop, one hub row among isolated nodes, one-column state."""
  alpha, beta = torch.tensor(0.5), torch.tensor(0.25)
  # (1) no edges at all: A x = 0 -> f = a (0 - x) + b x0
  n, d = 37, 9
  x, x0 = torch.randn(n, d), torch.randn(n, d)
  graph = G.CSRGraph(torch.zeros(2, 0, dtype=torch.long, device=dev), n)
  w = torch.zeros(1, device=dev)
  out = ops.spmm_rhs(graph, w, x.to(dev), alpha.to(dev), beta.to(dev), x0.to(dev), True)
  assert_parity(out, torch.sigmoid(alpha) * (0 - x) + beta * x0, what='empty graph')
  # (2) a single node with a self loop of weight 1: f = b x0
  graph = G.CSRGraph(torch.zeros(2, 1, dtype=torch.long, device=dev), 1)
  out = ops.spmm_rhs(graph, torch.ones(1, device=dev), x[:1].contiguous().to(dev), alpha.to(dev), beta.to(dev),
                     x0[:1].contiguous().to(dev), True)
  assert_parity(out, beta * x0[:1], what='single self-loop')
  # (3) one hub row (long-row path) whose neighbours are otherwise isolated, d = 1
  n = 3000
  nb = torch.arange(1, n)
  ei = torch.stack([torch.zeros(n - 1, dtype=torch.long), nb])
  x1, x01 = torch.randn(n, 1), torch.randn(n, 1)
  w1 = torch.rand(n - 1) / n
  ref = R.rhs_laplacian(x1, ei, w1, alpha, beta, x01, False, True)
  graph = G.CSRGraph(ei.to(dev), n)
  assert graph.n_long_rows == 1 and graph.n_bin16 == 0
  out = ops.spmm_rhs(graph, ops.edge_to_csr_mean(graph, w1.to(dev)), x1.to(dev), alpha.to(dev), beta.to(dev), x01.to(dev), True)
  assert_parity(out, ref, what='single hub, d=1')
  # (4) attention on that hub graph: softmax over one 2999-entry row, every other row empty
  A, h = 8, 2
  g = torch.Generator().manual_seed(0)
  xx = torch.randn(n, 12, generator=g)
  Wq, Wk = torch.randn(A, 12, generator=g) * 0.3, torch.randn(A, 12, generator=g) * 0.3
  att_ref, _ = R.transformer_attention(xx, ei, Wq, torch.zeros(A), Wk, torch.zeros(A), h)
  qk = ops.linear(xx.to(dev), torch.cat([Wq, Wk]).to(dev), None)
  st = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False, q=qk, k=qk[:, A:], ldqk=2 * A)
  wm, att, _ = ops.edge_attention(graph, st, True, True, False, like=qk)
  assert_parity(att, att_ref, what='hub-only attention')
  wf, _, _ = ops.edge_attention(graph, st, True, False, False, like=qk)   # fused row path: hub phases only
  assert_parity(wf[:graph.e], att_ref.mean(dim=1)[graph.perm_long.cpu()], what='hub-only fused attention')


@pytest.mark.parametrize('n', [1, 7, 4096 + 3, 169343 * 8 + 1])
def test_lincomb(dev, n):
  """gnpde_lincomb: base + sum_j c_j v_j, aligned and unaligned (offset view) operands, in place."""
  from gnpde_amd import ops
  g = torch.Generator().manual_seed(n % 1000)
  buf = [torch.randn(n + 1, generator=g).to(dev) for _ in range(5)]
  for off in (0, 1):
    base = buf[0][off:off + n].contiguous() if off == 0 else buf[0][off:off + n]
    vs = [b[off:off + n] for b in buf[1:]]
    cs = [0.5, -1.25, 3.0, 0.125]
    ref = base.double()
    for v, c in zip(vs, cs):
      ref = ref + c * v.double()
    out = ops.lincomb(base, list(zip(vs, cs)))
    assert float((out.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    inplace = base.clone()
    ops.lincomb(inplace, [(vs[0], 2.0)], out=inplace)
    assert torch.allclose(inplace, base + 2.0 * vs[0], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('d', [64, 128, 256])
def test_spmm_row_deal_over_the_xcds(dev, d):
  """Which XCD takes a row does not enter the arithmetic: on a graph whose row length depends on the bits of the row id (as
  in an R-MAT graph) the builder picks the hashed-block deal, and the aggregation gives the SAME bits as with contiguous
  eighths forced (gnpde_tune(10, 1)), hub rows and rk4 stage epilogue included, and agrees with the oracle (csrc/spmm.hip
  "Rows -> XCDs", tools/xcd_check.py; the builder's choice itself is covered on the host, tests/test_host_cpu.py)."""
  import numpy as np
  n = 1 << 16
  ids = np.arange(n)
  pop = np.zeros(n, dtype=np.int64)
  for b in range(16):
    pop += (ids >> b) & 1
  deg = np.minimum(np.maximum((2000.0 * 0.316 ** pop).astype(np.int64), 1), 700)     # 17 hub rows (> 512 entries)
  gen = np.random.default_rng(d)
  row = np.repeat(ids, deg)
  col = gen.integers(0, n, row.size)
  ei = torch.from_numpy(np.stack([row, col]))
  g = torch.Generator().manual_seed(d)
  w = _weights(ei.size(1), d)
  x, x0, y = (torch.randn(n, d, generator=g) for _ in range(3))
  alpha, beta = torch.tensor(0.3), torch.tensor(-0.7)
  graph = G.CSRGraph(ei.to(dev), n)
  assert graph.struct.xcd_deal == _lib.XCD_HASHED and graph.xcd_imbalance_contiguous > 1.5 and graph.n_long_rows > 0
  w_csr = ops.edge_to_csr_mean(graph, w.to(dev))
  xd, x0d, yd = x.to(dev), x0.to(dev), y.to(dev)

  def run():
    f = ops.spmm_rhs(graph, w_csr, xd, alpha.to(dev), beta.to(dev), x0d, True).clone()
    u2 = torch.empty_like(xd)
    ops.spmm_rhs(graph, w_csr, xd, alpha.to(dev), beta.to(dev), x0d, True, stage=_lib.STAGE_RK2C, dt=0.7, y=yd, out_y=u2)
    return f, u2
  hashed = run()
  try:
    _lib.check(_lib.lib().gnpde_tune(_lib.TUNE_XCD_ROWS, 1))
    contiguous = run()
  finally:
    _lib.check(_lib.lib().gnpde_tune(_lib.TUNE_XCD_ROWS, 0))
  assert torch.equal(hashed[0], contiguous[0]) and torch.equal(hashed[1], contiguous[1])
  ref = R.rhs_laplacian(x, ei, w, alpha, beta, x0, no_alpha_sigmoid=False, add_source=True)
  assert_parity(hashed[0], ref, what='spmm_rhs, hashed row deal, d=%d' % d)
  assert_parity(hashed[1], (2 * y - x) + 0.7 * ref, what='RK2C stage, hashed row deal, d=%d' % d)


@pytest.mark.parametrize('att_type', ['cosine_sim', 'pearson'])
@pytest.mark.parametrize('norm_idx,square_plus', [(0, False), (1, True)])
def test_cosine_fused_path_agrees_with_the_generic_score_kernels_near_zero_norms(dev, att_type, norm_idx, square_plus):
  """cosine_sim / pearson in ODEFunc.forward run as scaled-dot scores of rows normalised after the projection (per-vector clamp of
  each norm at 1e-5: torch >= 1.12's cosine_similarity); the layer's [E,h] attention comes from the generic score kernels
  (scores_kernel<COSINE / PEARSON>).  Same evaluation assembled both ways -- with nodes whose head vectors are tiny (1e-7), exactly
  zero, or tiny on a few heads only -- within float32 rounding of each other (round-4 advisor item; the torch-version dependence of
  the clamp itself is pinned in tests/test_properties_cpu.py and stated in INTEGRATION.md)."""
  from gnpde_amd import ops
  n, d, h, A = 900, 24, 4, 16
  ei = random_graph(n, 5, seed=77, hubs=1, hub_deg=600)
  g = torch.Generator().manual_seed(78)
  x = torch.randn(n, d, generator=g) * 0.5
  x[:60] *= 1e-7              # tiny rows: every head vector of q and k below the clamp
  x[60:90] = 0.0              # zero rows
  opt = dict(heads=h, attention_dim=A, attention_type=att_type, attention_norm_idx=norm_idx, square_plus=square_plus, reweight_attention=False,
             beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=1000, add_source=True, no_alpha_sigmoid=False, mix_features=False,
             hidden_dim=d, augment=False, adjoint=False, tol_scale=1.0, data_norm='rw', method='rk4', step_size=1.0, max_iters=100,
             block='constant', function='transformer', time=1.0)
  func = G.ODEFuncTransformerAtt(d, d, opt, Data(x.to(dev), ei.to(dev)), dev).to(dev)
  with torch.no_grad():
    for p in func.parameters():
      if p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
    lay = func.multihead_att_layer
    lay.Q.bias.zero_()
    lay.K.bias.zero_()
    lay.Q.weight[:4] *= 1e-7      # head 0 of every query below the clamp, the other heads ordinary
    func.alpha_train.fill_(0.2)
    func.beta_train.fill_(0.3)
  xd = x.to(dev)
  func.x0 = xd
  with torch.no_grad():
    f_fused = func(0.0, xd)                                       # projection -> normalise rows -> scaled-dot kernels
    att, _ = lay(xd, func.edge_index)                             # generic score kernels, [E,h] in edge order
    graph = func._graph(xd)
    w = ops.edge_to_csr_mean(graph, att)
    f_generic = ops.spmm_rhs(graph, w, xd, func.alpha_train, func.beta_train, xd, True)
  assert torch.isfinite(f_fused).all() and torch.isfinite(att).all()
  assert_parity(f_fused, f_generic, 1e-5, '%s norm_idx %d squareplus %s' % (att_type, norm_idx, square_plus))


@pytest.mark.parametrize('case', ['hubs', 'uniform', 'duplicates', 'isolated', 'empty', 'one_node', 'directed'])
def test_native_graph_builder_equals_the_other_two(dev, case):
  """The arrays of gnpde_graph_t three ways: the host builder (csrc/graph_prep.cpp, edge list on the CPU), the torch-op builder on
  the device (graph.build_arrays_on_device) and the library's device builder (csrc/graph_device.hip, what a device edge list gets
  since round 6: blocks that hand over a new edge set every training forward).  Element for element equal: stable orders, the row
  records by class (longest first in the second class, ties in row order), long rows / columns and their chunks, every count."""
  from gnpde_amd import graph as GR
  g = torch.Generator().manual_seed(5)
  if case == 'hubs':
    n = 3000
    ei = random_graph(n, 6, seed=3, hubs=3, hub_deg=1400)
  elif case == 'uniform':
    n = 5000
    ei = random_graph(n, 20, seed=4)
  elif case == 'duplicates':
    n = 400
    ei = torch.randint(0, n, (2, 9000), generator=g)
    ei = torch.cat([ei, ei[:, :3000]], dim=1)                      # repeated entries keep their relative order
  elif case == 'isolated':
    n = 1000
    ei = torch.randint(0, 300, (2, 5000), generator=g)             # rows / columns 300.. have no entries
    ei[0, :700] = 7                                                # one long row ...
    ei[1, 700:1500] = 11                                           # ... and one long column
  elif case == 'empty':
    n, ei = 17, torch.zeros(2, 0, dtype=torch.int64)
  elif case == 'one_node':
    n, ei = 1, torch.zeros(2, 5, dtype=torch.int64)
  else:
    n = 2000
    ei = torch.stack([torch.randint(0, n, (30000,), generator=g), torch.randint(0, 50, (30000,), generator=g)])   # 50 long columns, no long row
  host = GR.CSRGraph(ei, n, device=dev)                            # CPU edge list: the host builder
  t_torch, c_torch = GR.build_arrays_on_device(ei.to(dev), n)
  t_nat, c_nat = GR.build_arrays_native(ei.to(dev), n)
  assert c_nat == c_torch, (c_nat, c_torch)
  assert (host.n_long_rows, host.n_long_chunks, host.n_long_cols, host.n_bin16, host.n_bin64, host.n_bin_le64, host.max_row_len, host.max_col_len) == \
         tuple(c_nat[k] for k in ('n_long_rows', 'n_long_chunks', 'n_long_cols', 'n_bin16', 'n_bin64', 'n_bin_le64', 'max_row_len', 'max_col_len'))
  assert set(t_nat) == set(t_torch) == set(host.t)
  for k in t_torch:
    a, b, c = t_nat[k].cpu(), t_torch[k].cpu(), host.t[k].cpu()
    assert a.dtype == torch.int32 and a.shape == b.shape == c.shape, (k, a.shape, b.shape, c.shape)
    assert torch.equal(a, b), 'native vs torch builder: %s' % k
    assert torch.equal(a, c), 'native vs host builder: %s' % k
  with pytest.raises(G._lib.GnpdeError):
    bad = torch.tensor([[0, n], [0, 0]], dtype=torch.int64, device=dev)
    GR.build_arrays_native(bad, n)


@pytest.mark.parametrize('att_type,beltrami,norm_idx,sqp', [('scaled_dot', False, 0, False), ('cosine_sim', False, 0, False), ('exp_kernel', False, 0, False),
                                                            ('exp_kernel', True, 0, False), ('scaled_dot', False, 1, True)])
def test_layer_mean_attention_is_the_head_mean_of_the_layer(dev, att_type, beltrami, norm_idx, sqp):
  """SpGraphTransAttentionLayer.mean_attention (what the hard-attention and rewiring blocks take in evaluation mode: the head mean straight
  out of the fused row kernels, in the order of the edge list) against forward(x, edge)[0].mean(dim=1) (the [E,h] attention of the
  generic passes, which the reference-recorded layer fixtures pin) -- incl. BLEND's split feature x positional kernel and hub rows."""
  n, d, h, A = 1500, 32, 4, 16
  ei = random_graph(n, 7, seed=9, hubs=2, hub_deg=700, isolated=2, dup=10).to(dev)
  g = torch.Generator().manual_seed(10)
  x = (torch.randn(n, d, generator=g) * 0.5).to(dev)
  opt = dict(heads=h, attention_dim=A, attention_type=att_type, attention_norm_idx=norm_idx, square_plus=sqp, reweight_attention=False,
             beltrami=beltrami, feat_hidden_dim=20, pos_enc_hidden_dim=12, leaky_relu_slope=0.2, hidden_dim=d, mix_features=False)
  layer = G.SpGraphTransAttentionLayer(d, d, opt, dev).to(dev)
  with torch.no_grad():
    for p in layer.parameters():
      if p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
    att, _ = layer(x, ei)
    w = layer.mean_attention(x, ei)
  assert w.shape == (ei.shape[1],) and torch.isfinite(w).all()
  assert_parity(w, att.mean(dim=1), tol=1e-5, what='mean_attention vs the head mean of the layer (%s)' % att_type)

