"""Worker for tests/test_distributed_cpu.py (launched by torch.distributed.run, gloo, CPU).

Runs the product's partition plan / shard index maps / halo exchange / sharded rk4 driver
(graph-neural-pde_amd/distributed.py) with a CHECKER backend built on the CPU oracle in place of the HIP
backend, and compares the gathered result with the unpartitioned oracle solve."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gnpde_amd as G  # noqa: E402
from gnpde_amd import distributed as D, _lib  # noqa: E402
from oracle import restate as R  # noqa: E402
from helpers import random_graph  # noqa: E402


class OracleBackend(object):
  """Same interface as distributed.NativeBackend; arithmetic from oracle/restate.py (test only)."""

  supports_split = True

  def __init__(self, shard, d, kind, params, alpha, beta):
    self.shard, self.d, self.kind, self.p, self.alpha, self.beta = shard, d, kind, params, alpha, beta

  def empty(self, rows):
    return torch.zeros(rows, self.d)

  def pack(self, u, out):
    out.copy_(u[self.shard.send_idx])
    return out

  def rhs_stage(self, u, x0, stage, part=None, dt=0.0, y=None, k1=None, k2=None, k3=None, out_k=None, out_y=None):
    """part = 'interior' evaluates only the rows without halo neighbours (the halo region of `u` may be stale
    then, so it is poisoned here to prove those rows do not read it), 'boundary' the others."""
    s, p = self.shard, self.p
    n = s.n_own
    ei = s.edge_index
    lo, hi = {None: (0, n), 'interior': (0, s.n_interior), 'boundary': (s.n_interior, n)}[part]
    if part == 'interior':
      keep = ei[0] < s.n_interior
      assert bool((ei[1][keep] < n).all()), 'an interior row references a halo column'
      u = u.clone()
      u[n:] = float('nan')
      ei = ei[:, keep]
      if self.kind != 'transformer':
        p = dict(p, edge_weight=p['edge_weight'][keep])
    rows = slice(lo, hi)
    if self.kind == 'transformer':
      att, _ = R.transformer_attention(u, ei, p['Wq'], p['bq'], p['Wk'], p['bk'], p['heads'])
      w = att.mean(dim=1)
    else:
      w = p['edge_weight']
    ax = R.spmm(ei, w, n, u)
    k = torch.sigmoid(self.alpha) * (ax - u[:n])
    if x0 is not None:
      k = k + self.beta * x0
    third = 1 / 3
    k, un = k[rows], u[:n][rows]
    if stage == _lib.STAGE_EULER:
      out_y[rows] = y[rows] + dt * k
    elif stage == _lib.STAGE_RK1C:
      out_y[rows] = un + dt * k * third
    elif stage == _lib.STAGE_RK2C:
      out_y[rows] = (2 * y[rows] - un) + dt * k
    elif stage == _lib.STAGE_RK3C:
      out_y[rows] = (2 * k1[rows] - un) + dt * k
    elif stage == _lib.STAGE_RK4C:
      out_y[rows] = ((6 * k1[rows] + 3 * un - y[rows]) + dt * k) * 0.125
    elif stage == _lib.STAGE_LINCOMB:          # plain evaluation: k = f(u) (the adaptive solvers' stage call)
      out_k[rows] = k
    else:
      raise ValueError(stage)


def main():
  dist.init_process_group('gloo')
  rank, world = dist.get_rank(), dist.get_world_size()
  n, d, A, h = 600, 12, 8, 2
  ei = random_graph(n, 5, seed=4, hubs=1, hub_deg=200, isolated=4)
  g = torch.Generator().manual_seed(1)
  x = torch.randn(n, d, generator=g)
  params = dict(Wq=torch.randn(A, d, generator=g) / d ** 0.5, Wk=torch.randn(A, d, generator=g) / d ** 0.5,
                bq=torch.randn(A, generator=g) * 0.1, bk=torch.randn(A, generator=g) * 0.1, heads=h)
  alpha, beta = torch.tensor(0.3), torch.tensor(0.2)
  # the ranks share the search for the partition: every rank scores its slice of the settings, the scores are all-gathered,
  # everybody recomputes the winner -- and must end up with the same partition, no worse than the partitioner's default
  plan = D.PartitionPlan.search(ei, n, world, rank=rank, group_size=world, per_rank=4)
  assert len(plan.candidates) == 4 * world and plan.candidates == sorted(plan.candidates, key=lambda c: (c['cost'], c['index']))
  assert sorted(c['index'] for c in plan.candidates) == list(range(4 * world))
  parts = [torch.zeros(n, dtype=torch.long) for _ in range(world)]
  dist.all_gather(parts, plan.part)
  assert all(torch.equal(p_, plan.part) for p_ in parts), 'the ranks disagree on the partition'
  assert D.PartitionPlan.score(plan.edge_index, plan.part, world)[0] == plan.candidates[0]['cost']
  default = D.PartitionPlan(ei, n, world, part=D.partition_rows(D.CSRGraph(ei, n, device='cpu'), world))
  assert plan.candidates[0]['cost'] <= D.PartitionPlan.score(default.edge_index, default.part, world)[0]
  shard = plan.shard(rank)
  # index-map invariants
  assert shard.n_own == int((plan.part == rank).sum())
  assert sum(shard.recv_counts) == shard.n_halo and shard.recv_counts[rank] == 0 and shard.send_counts[rank] == 0
  cnt = torch.tensor(shard.send_counts, dtype=torch.long)
  allc = [torch.zeros(world, dtype=torch.long) for _ in range(world)]
  dist.all_gather(allc, cnt)
  for q in range(world):
    assert int(allc[q][rank]) == shard.recv_counts[q], 'send/recv counts of the exchange do not match'
  ok = True
  for kind, T, method in (('transformer', 2.3, 'rk4'), ('laplacian', 3.0, 'euler'), ('laplacian', 2.0, 'rk4')):
    if kind == 'laplacian':
      _, wfull = G.get_rw_adj(ei, None, norm_dim=0, fill_value=0.0, num_nodes=n, dtype=torch.float32)
      p = dict(edge_weight=wfull[shard.edge_ids])
      rhs = lambda t, y: R.rhs_laplacian(y, ei, wfull, alpha, beta, x, False, True)
    else:
      p = params
      rhs = lambda t, y: R.rhs_transformer(y, ei, params['Wq'], params['bq'], params['Wk'], params['bk'], h, alpha, beta,
                                           x, False, True)
    be = OracleBackend(shard, d, kind, p, alpha, beta)
    solver = D.ShardedSolver(shard, be)
    x_own = D.scatter_rows(x, shard)
    y_own = solver.integrate(x_own, x_own, T, 1.0, method)
    y = D.gather_rows_all(y_own.clone(), plan, shard)
    ref = R.odeint_fixed(rhs, x, T, 1.0, method)
    e_inf, e_2 = R.parity_error(y, ref)
    steps = len(G.time_grid(torch.tensor([0.0, T]), 1.0)) - 1
    assert solver.n_exchanges == steps * (4 if method == 'rk4' else 1)
    if not (e_inf < 1e-5 and e_2 < 1e-5):
      ok = False
      print('rank %d %s %s: mismatch %g %g' % (rank, kind, method, e_inf, e_2))
  # adaptive methods (dopri5 / adaptive_heun) on the partitioned graph: the product's host controller over the sharded
  # evaluations with the all-reduced error norm, against the restated torchdiffeq 0.2.1 (oracle/shims) on the whole graph --
  # same number of evaluations (= same accept / reject sequence) and the same state
  from oracle.shims import install as REF_TORCHDIFFEQ
  _, wfull = G.get_rw_adj(ei, None, norm_dim=0, fill_value=0.0, num_nodes=n, dtype=torch.float32)
  for kind, method, T, rtol, atol in (('laplacian', 'dopri5', 2.5, 1e-4, 1e-6), ('transformer', 'dopri5', 1.5, 1e-5, 1e-7),
                                      ('laplacian', 'adaptive_heun', 1.0, 1e-3, 1e-5)):
    calls = [0]
    if kind == 'laplacian':
      p = dict(edge_weight=wfull[shard.edge_ids])

      def rhs(t, y):
        calls[0] += 1
        return R.rhs_laplacian(y, ei, wfull, alpha, beta, x, False, True)
    else:
      p = params

      def rhs(t, y):
        calls[0] += 1
        return R.rhs_transformer(y, ei, params['Wq'], params['bq'], params['Wk'], params['bk'], h, alpha, beta, x, False, True)
    be = OracleBackend(shard, d, kind, p, alpha, beta)
    solver = D.ShardedSolver(shard, be)
    x_own = D.scatter_rows(x, shard)
    tt = torch.tensor([0.0, T])
    y_own = solver.integrate_adaptive(x_own, x_own, tt, rtol, atol, n, method=method)
    y = D.gather_rows_all(y_own.clone(), plan, shard)
    ref = REF_TORCHDIFFEQ.odeint(rhs, x, tt, method=method, options={}, rtol=rtol, atol=atol)[1]
    e_inf, e_2 = R.parity_error(y, ref)
    if not (e_inf < 1e-5 and e_2 < 1e-5 and solver.n_evals == calls[0] and solver.n_exchanges == calls[0]):
      ok = False
      print('rank %d %s %s: mismatch %g %g, evaluations %d vs %d, exchanges %d' % (rank, kind, method, e_inf, e_2, solver.n_evals,
                                                                                   calls[0], solver.n_exchanges))
  if rank == 0 and ok:
    print('DIST_OK world=%d cut=%.3f halo=%d' % (world, plan.edge_cut(), shard.n_halo))
  dist.destroy_process_group()
  sys.exit(0 if ok else 1)


if __name__ == '__main__':
  main()
