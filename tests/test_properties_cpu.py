"""Property tests (hypothesis) of the host-side logic on arbitrary small inputs: graph preparation (both builders), the rows ->
XCD deal, the push order and the partition plan / shard index maps -- sizes 0 and 1, empty rows, duplicates, self loops, more
ranks than nodes."""
import ctypes

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st, HealthCheck

import gnpde_amd as G
from gnpde_amd import _lib, distributed as D
from gnpde_amd.graph import build_arrays_on_device, contiguous_deal_imbalance

FAST = settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])


@st.composite
def edge_lists(draw, max_n=40, max_e=160):
  n = draw(st.integers(1, max_n))
  e = draw(st.integers(0, max_e))
  rows = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
  cols = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
  return n, torch.tensor([rows, cols], dtype=torch.long).reshape(2, e)


@FAST
@given(edge_lists())
def test_graph_builders_agree_and_are_consistent(g):
  n, ei = g
  host = G.CSRGraph(ei, n, device='cpu')
  row, col = ei[0].numpy(), ei[1].numpy()
  order = np.argsort(row, kind='stable')
  assert np.array_equal(host.perm.numpy(), order) and np.array_equal(host.colidx.numpy(), col[order])
  deg = np.bincount(row, minlength=n)
  assert np.array_equal(host.rowptr.numpy(), np.concatenate([[0], np.cumsum(deg)]))
  rec = host.t['bin_rows'].numpy().reshape(-1, 4)[:host.n_bin16 + host.n_bin64]
  assert np.array_equal(np.sort(rec[:, 0]), np.nonzero((deg >= 1) & (deg <= _lib.LONG_ROW))[0])
  assert np.array_equal(rec[:, 2], deg[rec[:, 0]]) and np.array_equal(rec[:, 1], host.rowptr.numpy()[rec[:, 0]])
  arrays, counts = build_arrays_on_device(ei, n)
  for key in ('rowptr', 'cscptr'):
    assert np.array_equal(arrays[key].numpy(), host.t[key].numpy())
  for key, size in (('colidx', host.e), ('perm', host.e), ('rowidx', host.e), ('cscpos', host.e), ('bin_rows', 4 * len(rec))):
    assert np.array_equal(arrays[key].numpy()[:size], host.t[key].numpy()[:size]), key
  assert counts['n_bin16'] == host.n_bin16 and counts['n_bin64'] == host.n_bin64 and counts['n_long_rows'] == host.n_long_rows
  assert host.struct.xcd_deal in (_lib.XCD_CONTIGUOUS, _lib.XCD_HASHED) and host.xcd_imbalance_contiguous >= 1.0 - 1e-12
  assert abs(contiguous_deal_imbalance(arrays['rowptr'], 0, n) - host.xcd_imbalance_contiguous) < 1e-9


@FAST
@given(st.integers(0, 5000), st.integers(0, 5000), st.sampled_from([_lib.XCD_CONTIGUOUS, _lib.XCD_HASHED]))
def test_xcd_row_map_is_a_bijection_for_any_range(a, b, deal):
  rb, re_ = min(a, b), max(a, b)
  L = _lib.lib()
  shift, per = ctypes.c_int32(0), ctypes.c_int32(0)
  _lib.check(L.gnpde_xcd_row_map(rb, re_, deal, ctypes.byref(shift), ctypes.byref(per), None))
  m = np.full((8, max(per.value, 1)), -7, dtype=np.int32)
  _lib.check(L.gnpde_xcd_row_map(rb, re_, deal, ctypes.byref(shift), ctypes.byref(per), m.ctypes.data))
  m = m[:, :per.value]
  rows = m[m >= 0]
  assert np.array_equal(np.sort(rows), np.arange(rb, re_)) and np.all((m >= 0) | (m == -1))
  even, odd = m[:, 0:per.value - per.value % 2:2], m[:, 1:per.value - per.value % 2 + 1:2]
  both = (even >= 0) & (odd >= 0)
  assert np.all(odd[both] == even[both] + 1)


@FAST
@given(st.lists(st.integers(0, 300), min_size=1, max_size=9))
def test_push_order_is_a_proportional_interleaving(counts):
  L = _lib.lib()
  c = np.asarray(counts, dtype=np.int32)
  total = int(c.sum())
  order = np.full(max(total, 1), -1, dtype=np.int32)
  _lib.check(L.gnpde_push_order(c.ctypes.data_as(_lib.c_int_p), len(counts), order.ctypes.data_as(_lib.c_int_p)))
  order = order[:total]
  assert np.array_equal(np.sort(order), np.arange(total))
  seg = np.concatenate([[0], np.cumsum(c)])
  dest = np.searchsorted(seg, order, side='right') - 1
  for p in range(len(counts)):
    assert np.all(np.diff(order[dest == p]) == 1)
    if total:
      taken = np.cumsum(dest == p)
      # merging by (j + 1/2) / count_p keeps every destination within  1/2 + (P / 2) (count_p / total)  rows of its share
      bound = 0.5 + 0.5 * len(counts) * (c[p] / total)
      assert np.all(np.abs(taken - (np.arange(total) + 1) * (c[p] / total)) <= bound + 1e-9)


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(edge_lists(max_n=60, max_e=300), st.integers(1, 6))
def test_partition_plan_and_shards_cover_the_graph(g, world):
  n, ei = g
  plan = D.PartitionPlan(ei, n, world)
  assert plan.part.numel() == n and int(plan.part.min()) >= 0 and int(plan.part.max()) < world
  assert torch.equal(torch.sort(plan.order)[0], torch.arange(n))
  shards = [plan.shard(r) for r in range(world)]
  seen = torch.zeros(ei.shape[1], dtype=torch.long)
  tag = torch.arange(n, dtype=torch.float64) + 0.5
  bufs = []
  for s in shards:
    seen[s.edge_ids] += 1
    assert s.n_own == int((plan.part == s.rank).sum()) and s.n_local == s.n_own + s.n_halo and 0 <= s.n_interior <= s.n_own
    b = torch.full((s.n_local,), float('nan'), dtype=torch.float64)
    b[:s.n_own] = tag[s.own_old_ids]
    bufs.append(b)
  assert torch.all(seen == 1)
  for p, sp in enumerate(shards):           # the push of csrc/sharded.hip, on the host
    seg = 0
    for q, sq in enumerate(shards):
      cnt = sp.send_counts[q]
      assert cnt == sq.recv_counts[p]
      row0 = sq.n_own + sum(sq.recv_counts[:p])
      bufs[q][row0:row0 + cnt] = bufs[p][sp.send_idx[seg:seg + cnt]]
      seg += cnt
    assert seg == int(sp.send_idx.numel())
  for s, b in zip(shards, bufs):
    assert not torch.isnan(b).any()
    assert torch.equal(b[s.edge_index[1]], tag[ei[1][s.edge_ids]]) and torch.equal(b[s.edge_index[0]], tag[ei[0][s.edge_ids]])
    if s.edge_index.shape[1]:
      halo_rows = torch.unique(s.edge_index[0][s.edge_index[1] >= s.n_own])
      assert halo_rows.numel() == s.n_own - s.n_interior and (halo_rows.numel() == 0 or int(halo_rows.min()) >= s.n_interior)
  links = D.pair_traffic(plan.edge_index, plan.part, world)
  for r, s in enumerate(shards):
    assert links[r].tolist() == s.recv_counts


@st.composite
def weighted_edges(draw, max_n=25, max_e=80):
  n, ei = draw(edge_lists(max_n=max_n, max_e=max_e))
  e = ei.shape[1]
  weighted = draw(st.booleans())
  w = None
  if weighted:
    w = torch.tensor(draw(st.lists(st.floats(0.125, 4.0, width=32), min_size=e, max_size=e)), dtype=torch.float32)
  return n, ei, w


@FAST
@given(weighted_edges(), st.sampled_from([0, 1]), st.sampled_from([0.0, 1.0, 2.0]))
def test_normalisation_helpers_match_the_oracle(g, norm_dim, fill):
  """get_rw_adj / gcn_norm_fill_val / add_remaining_self_loops (utils.py: the blocks' graph preparation, reference
  src/utils.py:55-123) against the oracle on arbitrary edge lists -- duplicates, existing self loops (their weight is kept),
  isolated nodes (inf -> 0 in the symmetric normalisation)."""
  from oracle import restate as R
  n, ei, w = g
  a_i, a_w = G.add_remaining_self_loops(ei, w, 1.0 if fill == 0.0 else fill, n)
  b_i, b_w = R.add_remaining_self_loops(ei, w, 1.0 if fill == 0.0 else fill, n)
  assert torch.equal(a_i, b_i) and ((a_w is None and b_w is None) or torch.allclose(a_w, b_w))
  a_i, a_w = G.get_rw_adj(ei, w, norm_dim=norm_dim, fill_value=fill, num_nodes=n, dtype=torch.float32)
  b_i, b_w = R.get_rw_adj(ei, w, norm_dim=norm_dim, fill_value=fill, num_nodes=n, dtype=torch.float32)
  assert torch.equal(a_i, b_i) and torch.allclose(a_w, b_w, rtol=1e-6, atol=1e-7, equal_nan=True)
  a_i, a_w = G.gcn_norm_fill_val(ei, w, fill_value=fill, num_nodes=n, dtype=torch.float32)
  b_i, b_w = R.gcn_norm_fill_val(ei, w, fill_value=fill, num_nodes=n, dtype=torch.float32)
  assert torch.equal(a_i, b_i) and torch.allclose(a_w, b_w, rtol=1e-6, atol=1e-7, equal_nan=True)


@FAST
@given(st.floats(0.05, 40.0), st.floats(0.05, 3.0))
def test_time_grid_is_torchdiffeqs(T, h):
  """odeint.time_grid: ceil(T / h + 1) points t0 + i h with the last one replaced by T, in float32 like torchdiffeq's
  FixedGridODESolver (so the final step may be short, never long, never missing)."""
  from gnpde_amd.odeint import time_grid
  from oracle import restate as R
  t = torch.tensor([0.0, T], dtype=torch.float32)
  grid = time_grid(t, h)
  assert torch.equal(grid, R.time_grid(T, h))
  assert float(grid[0]) == 0.0 and grid[-1] == t[-1] and grid.numel() >= 2
  steps = grid[1:] - grid[:-1]
  assert bool((steps[:-1] > 0).all()) and float(steps.max()) <= h + 1e-6 * max(T, 1.0)      # (grid points are float32)


@FAST
@given(edge_lists(max_n=15, max_e=60), edge_lists(max_n=15, max_e=60), st.integers(0, 2 ** 31 - 1))
def test_rewiring_sparse_helpers_match_dense(a, b, seed):
  """Test-side composite of the rewiring block's two-hop step (tests/sparse_composite.py) (what torch_sparse.spspmm / coalesce compute for the reference,
  src/block_transformer_rewiring.py:68-86): COO product and duplicate-summing coalesce against dense matrices, the result
  sorted row-major without duplicates."""
  from sparse_composite import _spspmm, _coalesce
  n = max(a[0], b[0])
  ia, ib = a[1], b[1]
  g = torch.Generator().manual_seed(seed)
  va, vb = torch.rand(ia.shape[1], generator=g), torch.rand(ib.shape[1], generator=g)
  dense = lambda i, v: torch.zeros(n, n, dtype=torch.float64).index_put_((i[0], i[1]), v.double(), accumulate=True)  # noqa: E731
  ic, vc = _spspmm(ia, va, ib, vb, n)
  assert torch.allclose(dense(ic, vc), dense(ia, va) @ dense(ib, vb), atol=1e-5)
  key = ic[0] * n + ic[1]
  assert ic.shape[1] <= 1 or bool((key[1:] > key[:-1]).all())
  id_, vd = _coalesce(torch.cat([ia, ib], dim=1), torch.cat([va, vb]), n)
  assert torch.allclose(dense(id_, vd), dense(ia, va) + dense(ib, vb), atol=1e-5)
  key = id_[0] * n + id_[1]
  assert id_.shape[1] <= 1 or bool((key[1:] > key[:-1]).all())


@FAST
@given(edge_lists(), st.randoms(use_true_random=False))
def test_relabelled_graph_keeps_every_rows_edges_in_the_callers_order(g, rnd):
  """graph.LocalityView under an ARBITRARY permutation of the nodes (duplicates, self loops, empty rows): row inv[v] of the
  relabelled CSR lists the same caller edges in the same order as row v of the original (equal `perm` slices -- this is what makes
  every row sum on the relabelled graph the same sum), its column ids are the relabelled neighbours, and the dense operator is
  P A P^T."""
  from gnpde_amd.graph import LocalityView
  n, ei = g
  order = list(range(n))
  rnd.shuffle(order)
  order = torch.tensor(order, dtype=torch.int64)
  base = G.CSRGraph(ei, n, device='cpu')
  view = LocalityView(base, order, {})
  v = view.graph
  assert torch.equal(view.inv[view.order], torch.arange(n)) and v.e == base.e and v.n == n
  rp_b, rp_v = base.rowptr.long(), v.rowptr.long()
  for i in range(n):
    old = int(order[i])
    assert torch.equal(v.perm.long()[rp_v[i]:rp_v[i + 1]], base.perm.long()[rp_b[old]:rp_b[old + 1]])
    assert torch.equal(order[v.colidx.long()[rp_v[i]:rp_v[i + 1]]], base.colidx.long()[rp_b[old]:rp_b[old + 1]])
  w = torch.arange(1, base.e + 1, dtype=torch.float64)        # a weight per CALLER edge
  dense = lambda gr: torch.zeros(n, n, dtype=torch.float64).index_put_(   # noqa: E731
      (torch.repeat_interleave(torch.arange(n), (gr.rowptr.long()[1:] - gr.rowptr.long()[:-1])), gr.colidx.long()),
      w[gr.perm.long()], accumulate=True)
  a, b = dense(base), dense(v)
  assert torch.equal(b, a[order][:, order])
  x = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
  assert torch.equal(view.leave(view.enter(x)), x)
  buf = torch.empty(n, 4)[:, :3]                               # a padded (non-contiguous) destination
  assert torch.equal(view.enter(x, out=buf), x[order]) and torch.equal(view.leave(buf, out=torch.empty(n, 3)), x)


@pytest.mark.parametrize('centre', [False, True])
@pytest.mark.parametrize('dk', [4, 8, 24])
def test_cosine_scores_are_scaled_dot_scores_of_normalised_rows(centre, dk):
  """The identity behind csrc/misc.hip normalise_heads_kernel (cosine_sim / pearson run on the scaled-dot kernels): with
  q^ = sqrt(d_k) (q - mean) / max(|q - mean|, eps) and k^ = (k - mean) / max(|k - mean|, eps), eps = 1e-5,
  q^ . k^ / sqrt(d_k) is torch.nn.functional.cosine_similarity(eps = 1e-5) of the (mean-centred) head vectors -- the reference's score
  (src/function_transformer_attention.py:197-206, restated in oracle/restate.py) -- for ordinary, tiny, zero and mixed pairs alike."""
  from oracle import restate as R
  g = torch.Generator().manual_seed(dk + int(centre))
  n, h = 400, 3
  A = h * dk
  q = torch.randn(n, A, generator=g)
  k = torch.randn(n, A, generator=g)
  q[:40] *= 1e-4                      # tiny on both sides
  k[:40] *= 1e-4
  q[40:50] = 0.0                      # zero on both sides
  k[40:50] = 0.0
  q[50:60] *= 1e-7                    # below eps on one side only
  k[60:70] = 0.0
  edge = torch.stack([torch.arange(n), torch.arange(n)])
  eye = torch.eye(A)
  zero = torch.zeros(A)
  _, want = R.transformer_attention(torch.cat([q, k], dim=1), edge, torch.cat([eye, torch.zeros(A, A)], dim=1), zero,
                                    torch.cat([torch.zeros(A, A), eye], dim=1), zero, h,
                                    attention_type='pearson' if centre else 'cosine_sim')

  def normalise(v, scale):
    v = v.view(n, h, dk)
    if centre:
      v = v - v.mean(dim=2, keepdim=True)
    nrm = v.pow(2).sum(dim=2, keepdim=True).sqrt().clamp_min(1e-5)
    return (v * (scale / nrm)).reshape(n, A)
  qn, kn = normalise(q, dk ** 0.5), normalise(k, 1.0)
  got = (qn.view(n, h, dk) * kn.view(n, h, dk)).sum(dim=2) / dk ** 0.5
  assert got.shape == want.shape
  assert float((got - want).abs().max()) < 5e-6


def test_cosine_clamp_against_torch_1_8_product_clamp():
  """The reference pins torch 1.8, whose cosine_similarity divides by max(|x| |y|, eps) -- ONE clamp on the product (ATen
  Distance.cpp: w12 / sqrt(clamp_min(w1 w2, eps^2))); torch >= 1.12 -- the torch the fixtures were recorded with, and what the
  per-vector normalisation of csrc/misc.hip expresses -- divides by max(|x|, eps) max(|y|, eps).  Both give the true cosine, and so
  agree, wherever NEITHER clamps: |x| >= eps, |y| >= eps and |x| |y| >= eps.  They differ (a) when |x| |y| < eps = 1e-5 with both norms
  still above eps -- head vectors of norm ~3e-3 and below on both sides: torch 1.8 shrinks the score by |x| |y| / eps, the per-norm form
  returns the true cosine -- and (b) when exactly one norm is below eps with the product above it: torch 1.8 returns the true cosine, the
  per-norm form shrinks it by |x| / eps.  Pinned here; stated in INTEGRATION.md (torch-version dependence of cosine_sim / pearson)."""
  g = torch.Generator().manual_seed(7)
  n, dk, eps = 4000, 8, 1e-5
  x = torch.randn(n, dk, generator=g, dtype=torch.float64)
  y = torch.randn(n, dk, generator=g, dtype=torch.float64)
  x = x * 10.0 ** torch.empty(n, 1, dtype=torch.float64).uniform_(-8, 2, generator=g)
  y = y * 10.0 ** torch.empty(n, 1, dtype=torch.float64).uniform_(-8, 2, generator=g)
  nx, ny = x.norm(dim=1), y.norm(dim=1)
  dot = (x * y).sum(dim=1)
  true = dot / (nx * ny)
  old = dot / (nx * ny).clamp_min(eps)                          # torch 1.8
  new = dot / (nx.clamp_min(eps) * ny.clamp_min(eps))           # torch >= 1.12 (= torch.nn.functional.cosine_similarity here)
  ref = torch.nn.functional.cosine_similarity(x, y, dim=1, eps=eps)
  assert float((new - ref).abs().max()) < 1e-12
  neither = (nx >= eps) & (ny >= eps) & (nx * ny >= eps)
  assert int(neither.sum()) > 500
  assert float((old[neither] - new[neither]).abs().max()) < 1e-12 and float((new[neither] - true[neither]).abs().max()) < 1e-12
  a = (nx >= eps) & (ny >= eps) & (nx * ny < eps)                # (a): only the product clamp is active
  assert int(a.sum()) > 50
  assert float((new[a] - true[a]).abs().max()) < 1e-12
  assert float((old[a] - true[a] * (nx * ny)[a] / eps).abs().max()) < 1e-12
  b = ((nx < eps) ^ (ny < eps)) & (nx * ny >= eps)               # (b): only a per-norm clamp is active
  assert int(b.sum()) > 20
  assert float((old[b] - true[b]).abs().max()) < 1e-12
  assert float((new[b] - true[b] * torch.minimum(nx, ny)[b] / eps).abs().max()) < 1e-12
