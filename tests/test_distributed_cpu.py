"""World-size-2 (and 3) gloo runs of the sharded solve on CPU: partition plan, halo index maps, the
all-to-all exchange and the sharded rk4 / euler driver against the unpartitioned oracle."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import gnpde_amd as G
from gnpde_amd import distributed as D
from helpers import random_graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_solve_gloo(world):
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py')]
  env = dict(os.environ, OMP_NUM_THREADS='2')
  res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
  assert res.returncode == 0 and 'DIST_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.parametrize('world', [2, 3])
def test_transport_ladder_falls_down_every_rung(world, tmp_path):
  """bench.py --gpus N picks its halo transport by distributed.negotiate_transport (p2p -> rccl -> torch, each rung checked in two
  phases by every rank).  Scripted failures on single ranks, in either phase, by exception or by a wrong result: every rank drops
  the rung (and releases it), every rank ends on the same next rung, the reasons are recorded, nothing is attempted after the
  choice; no rung left -> None on every rank."""
  import json
  out = str(tmp_path / 'ladder.json')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'dist_ladder_worker.py'), out]
  res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS='2'))
  assert res.returncode == 0 and 'LADDER_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
  per_rank = json.load(open(out))
  assert len(per_rank) == world
  for rank, scen in enumerate(per_rank):
    for name, r in scen.items():
      assert r['chosen'] == r['expected'], (rank, name, r)
      assert r['same_on_all_ranks'], (rank, name, r)
      ladder = ['torch'] if name == 'only_torch_offered' else ['p2p', 'rccl', 'torch']
      tried = ladder if r['chosen'] is None else ladder[:ladder.index(r['chosen']) + 1]
      assert r['rejected'] == [c for c in tried if c != r['chosen']], (rank, name, r)          # every dropped rung is released, in order
      assert [c for c, _ in r['calls'] if c not in tried] == [], (rank, name, r)               # nothing past the choice is attempted
      for c in r['rejected']:
        assert c in r['notes'] and r['notes'][c], (rank, name, r)                              # ... with a reason on every rank
  # the rank that failed names its own reason, the others are told it was another rank
  a = per_rank[0]['p2p_fails_on_rank1_phase1']['notes']['p2p']
  b = per_rank[1]['p2p_fails_on_rank1_phase1']['notes']['p2p']
  assert 'another rank' in a and 'scripted failure' in b, (a, b)


def test_plan_is_a_permutation_and_local_graphs_cover_all_edges():
  n = 500
  ei = random_graph(n, 6, seed=9, hubs=1, hub_deg=300)
  plan = D.PartitionPlan(ei, n, 4)
  assert torch.equal(torch.sort(plan.order)[0], torch.arange(n))
  seen = torch.zeros(ei.shape[1], dtype=torch.long)
  for r in range(4):
    sh = plan.shard(r)
    seen[sh.edge_ids] += 1
    # local column ids: owned first, halo after; every halo id maps back to the right global node
    glob_cols = ei[1][sh.edge_ids]
    loc = sh.edge_index[1]
    own = loc < sh.n_own
    assert torch.equal(sh.own_old_ids[loc[own]], glob_cols[own])
    assert torch.equal(plan.order[sh.halo_new[loc[~own] - sh.n_own]], glob_cols[~own])
    assert torch.equal(sh.own_old_ids[sh.edge_index[0]], ei[0][sh.edge_ids])
    # interior rows come first and never reference a halo column
    rows_with_halo = torch.unique(sh.edge_index[0][~own])
    assert rows_with_halo.numel() == sh.n_own - sh.n_interior
    assert rows_with_halo.numel() == 0 or int(rows_with_halo.min()) >= sh.n_interior
  assert torch.all(seen == 1)


def test_native_backend_refuses_cpu():
  n = 50
  ei = random_graph(n, 3, seed=1)
  plan = D.PartitionPlan(ei, n, 2)
  with pytest.raises(G.GnpdeError):
    D.NativeBackend(plan.shard(0), 8, 'cpu', 'laplacian', dict(edge_weight=torch.ones(1)), torch.tensor(0.), torch.tensor(0.))


@pytest.mark.parametrize('world', [2, 3, 5, 8])
def test_push_lands_every_boundary_row_in_the_peers_halo_slot(world):
  """The P2P transport of the native sharded solver (csrc/sharded.hip, distributed.P2PContext) on the host: rank p pushes
  the rows send_idx[segment of q] into peer q's stage buffer starting at row n_own_q + sum(recv_counts_q[:p]).  For every
  pair the counts must agree and the rows that arrive must be, in order, the global nodes q's local graph expects in those
  halo slots -- simulated here with a global tag per row and NaN-poisoned halos, then checked against the local edge lists."""
  n = 1200
  ei = random_graph(n, 7, seed=31 + world, hubs=2, hub_deg=400)
  plan = D.PartitionPlan(ei, n, world)
  shards = [plan.shard(r) for r in range(world)]
  tag = torch.arange(n, dtype=torch.float64) * 3.0 + 1.0                    # a value that identifies the global node
  bufs = []
  for s in shards:
    b = torch.full((s.n_local,), float('nan'), dtype=torch.float64)
    b[:s.n_own] = tag[s.own_old_ids]
    bufs.append(b)
  for p, sp in enumerate(shards):
    assert sp.send_counts[p] == 0 and sp.recv_counts[p] == 0
    seg = 0
    for q, sq in enumerate(shards):
      cnt = sp.send_counts[q]
      assert cnt == sq.recv_counts[p], (p, q)
      row0 = sq.n_own + sum(sq.recv_counts[:p])                                 # P2PContext.peer_row0
      rows = sp.send_idx[seg:seg + cnt]
      assert rows.numel() == cnt and (cnt == 0 or int(rows.max()) < sp.n_own)
      bufs[q][row0:row0 + cnt] = bufs[p][rows]                                  # push_rows_kernel
      seg += cnt
    assert seg == int(sp.send_idx.numel())
  for s, b in zip(shards, bufs):
    assert not torch.isnan(b).any(), 'a halo slot was never written'
    # every local column -- owned or halo -- now holds the tag of the global node the edge points to
    assert torch.equal(b[s.edge_index[1]], tag[ei[1][s.edge_ids]])
    assert sum(s.recv_counts) == s.n_halo


def test_push_order_interleaves_the_destinations():
  """gnpde_push_order (what the P2P push walks, csrc/sharded.hip): a permutation of the destination-grouped send list that
  keeps each destination's rows in order and, at every moment of the walk, has taken from destination p its share
  count_p / total of the rows so far (within a row or so: 1/2 + (P / 2) count_p / total, checked on arbitrary counts in
  tests/test_properties_cpu.py) -- so all xGMI links of the rank are busy from the first row to the last instead of one after
  the other."""
  import ctypes
  import numpy as np
  from gnpde_amd import _lib
  L = _lib.lib()
  for counts in ([0, 5], [7, 0, 7], [0, 17537, 9000, 12000, 1, 0, 15000, 16000], [3, 0, 0, 0], [0, 0], [0, 1, 1, 1, 1, 1, 1, 1],
                 [100000, 0, 3]):
    c = np.asarray(counts, dtype=np.int32)
    total = int(c.sum())
    order = np.full(max(total, 1), -1, dtype=np.int32)
    _lib.check(L.gnpde_push_order(c.ctypes.data_as(_lib.c_int_p), len(counts), order.ctypes.data_as(_lib.c_int_p)))
    order = order[:total]
    assert np.array_equal(np.sort(order), np.arange(total))
    seg = np.concatenate([[0], np.cumsum(c)])
    dest = np.searchsorted(seg, order, side='right') - 1
    for p in range(len(counts)):
      mine = order[dest == p]
      assert np.all(np.diff(mine) == 1) and (mine.size == 0 or mine[0] == seg[p])       # own rows in order
      if total:
        taken = np.cumsum(dest == p)                                                     # after w + 1 rows of the walk
        share = (np.arange(total) + 1) * (c[p] / total)
        assert np.all(np.abs(taken - share) <= 1.0 + 1e-9), (counts, p)
  bad = np.asarray([1, -2], dtype=np.int32)
  out = np.zeros(4, dtype=np.int32)
  with pytest.raises(G.GnpdeError):
    _lib.check(L.gnpde_push_order(bad.ctypes.data_as(_lib.c_int_p), 2, out.ctypes.data_as(_lib.c_int_p)))


def test_boundary_cuts_tile_the_boundary_rows_with_even_entry_counts():
  """distributed.boundary_cuts (the row ranges of the chunked boundary pass): consecutive, non-empty, tiling
  [n_interior, n_own); entry counts within one row's worth of even; degenerate inputs (fewer rows than ranges, no boundary
  rows, rows without entries) handled."""
  n, world = 4000, 4
  ei = random_graph(n, 7, seed=8, hubs=1, hub_deg=500)
  plan = D.PartitionPlan(ei, n, world)
  for rank in range(world):
    sh = plan.shard(rank)
    rows = sh.edge_index[0]
    deg = torch.bincount(rows, minlength=sh.n_own)
    for k in (1, 2, 3, 7):
      b = D.boundary_cuts(rows, sh.n_interior, sh.n_own, k)
      assert b[0] == sh.n_interior and b[-1] == sh.n_own and len(b) == k + 1 and all(x < y for x, y in zip(b, b[1:]))
      per = [int(deg[b[c]:b[c + 1]].sum()) for c in range(k)]
      total = sum(per)
      assert max(abs(p - total / k) for p in per) <= int(deg[sh.n_interior:sh.n_own].max()) + 1, (k, per)
  rows = torch.tensor([0, 0, 1, 5, 5, 5], dtype=torch.int64)
  assert D.boundary_cuts(rows, 4, 6, 8) == [4, 5, 6]          # two boundary rows: at most two ranges
  assert D.boundary_cuts(rows, 6, 6, 3) == [6, 6]             # no boundary rows: one empty range
  assert D.boundary_cuts(rows, 2, 6, 2) in ([2, 5, 6], [2, 6 - 1, 6])   # rows 2..4 have no entries: the cut still advances


def test_chunked_push_order_groups_the_send_slots_by_the_row_range_that_computes_them():
  """NativeShardedSolver._chunked_push_order (the walk of gnpde_sharded_solver_set_boundary_chunks): a permutation of the send
  slots; slots [ptr[c], ptr[c+1]) hold exactly the rows of boundary range c (rows of the interior pass ride with range 0);
  inside a range every destination keeps its own order and the destinations are interleaved in proportion."""
  n, world = 3000, 4
  ei = random_graph(n, 6, seed=5, hubs=1, hub_deg=300)
  plan = D.PartitionPlan(ei, n, world)
  for rank in range(world):
    sh = plan.shard(rank)
    solver = D.NativeShardedSolver.__new__(D.NativeShardedSolver)
    solver.shard = sh
    b0, b1 = sh.n_interior, sh.n_own
    cuts = [b0, b0 + (b1 - b0) // 5, b0 + (b1 - b0) // 2, b1]
    chunks = [(cuts[c], cuts[c + 1], None, None, None) for c in range(3)]
    order, ptr = solver._chunked_push_order(chunks)
    n_send = int(sum(sh.send_counts))
    assert sorted(order) == list(range(n_send)) and ptr[0] == 0 and ptr[-1] == n_send and len(ptr) == 4
    rows = sh.send_idx.to(torch.int64)
    seg = torch.cumsum(torch.tensor([0] + [int(v) for v in sh.send_counts]), 0)
    for c in range(3):
      slots = torch.tensor(order[ptr[c]:ptr[c + 1]], dtype=torch.int64)
      r = rows[slots]
      lo = 0 if c == 0 else cuts[c]
      assert bool(((r >= lo) & (r < cuts[c + 1])).all())
      dest = torch.searchsorted(seg, slots, right=True) - 1
      for p in range(world):
        mine = slots[dest == p]
        assert bool((mine[1:] > mine[:-1]).all())                      # a destination's rows keep their order
        if mine.numel() and slots.numel():
          taken = torch.cumsum((dest == p).to(torch.float64), 0)
          share = (torch.arange(slots.numel()) + 1) * (mine.numel() / slots.numel())
          assert float((taken - share).abs().max()) <= 1.0 + 1e-9


def test_partition_plan_takes_the_candidate_with_the_cheapest_busiest_link():
  """PartitionPlan scores a few runs of the heuristic partitioner by what a partitioned evaluation waits for -- the busiest
  xGMI link (rows one rank receives from ONE peer) and the busiest rank -- and keeps the cheapest; the scoring function agrees
  with the shards' own receive counts, and two ranks building the plan independently reach the same partition."""
  n = 3000
  ei = random_graph(n, 8, seed=77, hubs=3, hub_deg=500)
  plan = D.PartitionPlan(ei, n, 4)
  assert plan.candidates is not None and len(plan.candidates) == len(D.PartitionPlan.CANDIDATES)
  links = D.pair_traffic(plan.edge_index, plan.part, 4)
  for r in range(4):
    assert links[r].tolist() == plan.shard(r).recv_counts
  best = min(plan.candidates, key=lambda c: c['cost'])
  assert int(links.max()) == best['max_link_rows'] and int(links.sum(dim=1).max()) == best['max_halo_rows']
  assert all(c['cost'] == D.LINK_ROW_COST * c['max_link_rows'] + c['max_part_work'] for c in plan.candidates)
  again = D.PartitionPlan(ei, n, 4)
  assert torch.equal(again.part, plan.part)
  assert torch.bincount(plan.part, minlength=4).min() > 0
  # a given partition is taken as it is; one part needs no search
  fixed = D.PartitionPlan(ei, n, 4, part=plan.part)
  assert fixed.candidates is None and torch.equal(fixed.part, plan.part)
  assert len(D.PartitionPlan(ei, n, 1).candidates or []) == 0


@pytest.mark.parametrize('name', ['func_transformer_beltrami_expkernel', 'func_transformer_beltrami_expkernel_sqp',
                                  'func_transformer_beltrami_expkernel_labels_n1', 'func_transformer_expkernel_n0',
                                  'func_transformer_cosine_n0', 'func_transformer_pearson_n1', 'func_gat_n0', 'func_gat_slope'])
def test_partitioned_problem_parameters_reproduce_the_reference_attention(name):
  """What `distributed._sharded_problem` hands the partitioned backend -- (W, b) pairs, score type, kernel scalars -- evaluated by the
  CPU oracle must give the attention the REFERENCE recorded for that function: in particular BLEND's split feature x positional
  kernel as ONE exp kernel over the concatenated, length-scaled projections (heads of width 2 d_k, output_var = ov_x ov_p,
  lengthscale 1), and the GAT layer's transposed weight + flattened `a`.  No GPU involved."""
  from helpers import Fixture, Data
  from oracle import restate as R
  fx = Fixture(name)
  x = fx.t('x')
  fcls = {'transformer': G.ODEFuncTransformerAtt, 'GAT': G.ODEFuncAtt}[fx.opt['function']]
  func = fcls(x.shape[1], x.shape[1], fx.opt, Data(x, fx.t('edge_index')), torch.device('cpu'))
  func.load_state_dict(fx.params, strict=True)
  kind, p = D._sharded_problem(func)
  edge = func.edge_index
  want = fx.t('attention')
  if kind == 'gat':
    dk = p['W'].shape[0] // p['heads']
    got, _ = R.gat_attention(x, edge, p['W'].t(), p['a'].view(2 * dk, 1, 1), p['heads'], p['leaky_slope'], p['norm_idx'])
  else:
    assert kind == 'transformer'
    got, _ = R.transformer_attention(x, edge, p['Wq'], p['bq'], p['Wk'], p['bk'], p['heads'], attention_type=p['att_type'],
                                     norm_idx=p['norm_idx'], square_plus=p['square_plus'], output_var=p.get('output_var'),
                                     lengthscale=p.get('lengthscale'))
  assert got.shape == want.shape
  err = float((got - want).abs().max() / want.abs().max())
  assert err < 2e-6, (name, kind, err)
