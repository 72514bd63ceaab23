"""The native row-partitioned solver (csrc/sharded.hip: pack + grouped RCCL send/recv + interior / boundary passes, the whole
solve in one hipGraph) on ONE GPU.

A single process cannot own two RCCL ranks on one device, so the exchange is exercised as a SELF exchange: the shard
below mirrors a third of its own rows into a halo region ("rank 0 needs rows of rank 0"), the boundary rows read those
mirrored copies instead of the originals, and every evaluation has to refresh them through ncclSend / ncclRecv to self
inside the graph.  If the exchange were skipped, stale or mis-ordered, the boundary rows would integrate garbage
(the halo region starts as NaN).  Index maps for real multi-rank partitions are covered by the gloo tests
(tests/test_distributed_cpu.py) and by test_sharded_native_backend_one_evaluation (per-rank passes vs the oracle).
"""
import pytest
import torch

import gnpde_amd as G
from gnpde_amd import distributed as D, _lib
from oracle import restate as R
from helpers import assert_parity, random_graph

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method='thread')]


class SelfShard(object):
  """World-1 shard whose boundary rows reference MIRRORED copies of owned rows (see module docstring)."""

  def __init__(self, ei, n, frac_interior=0.6, frac_mirror=0.35, seed=0):
    g = torch.Generator().manual_seed(seed)
    self.rank, self.world = 0, 1
    self.n_own = n
    self.n_interior = int(n * frac_interior)
    mirrored = torch.sort(torch.randperm(n, generator=g)[:int(n * frac_mirror)]).values
    pos = torch.full((n,), -1, dtype=torch.long)
    pos[mirrored] = torch.arange(mirrored.numel())
    row, col = ei
    redirect = (row >= self.n_interior) & (pos[col] >= 0)
    col_local = torch.where(redirect, n + pos[col], col)
    self.edge_index = torch.stack([row, col_local])
    self.n_halo = int(mirrored.numel())
    self.send_idx = mirrored
    self.send_counts = [self.n_halo]
    self.recv_counts = [self.n_halo]
    self.n_redirected = int(redirect.sum())

  @property
  def n_local(self):
    return self.n_own + self.n_halo


def _problem(kind, seed=3):
  n, d, A, h = 6000, 128, 16, 4
  ei = random_graph(n, 8, seed=seed, hubs=2, hub_deg=1500)
  g = torch.Generator().manual_seed(seed)
  x = torch.randn(n, d, generator=g)
  params = dict(Wq=torch.randn(A, d, generator=g) / d ** 0.5, Wk=torch.randn(A, d, generator=g) / d ** 0.5,
                bq=torch.randn(A, generator=g) * 0.1, bk=torch.randn(A, generator=g) * 0.1, heads=h)
  alpha, beta = torch.tensor(0.3), torch.tensor(0.2)
  if kind == 'laplacian':
    _, w = G.get_rw_adj(ei, None, norm_dim=0, fill_value=0.0, num_nodes=n, dtype=torch.float32)
    rhs = lambda t, y: R.rhs_laplacian(y, ei, w, alpha, beta, x, False, True)   # noqa: E731
    p = dict(edge_weight=w)
  else:
    rhs = lambda t, y: R.rhs_transformer(y, ei, params['Wq'], params['bq'], params['Wk'], params['bk'], h, alpha, beta,  # noqa: E731
                                         x, False, True)
    p = params
  return n, d, ei, x, p, alpha, beta, rhs


@pytest.mark.parametrize('kind', ['transformer', 'laplacian'])
@pytest.mark.parametrize('method,T', [('rk4', 3.0), ('euler', 2.5)])
def test_self_exchange_inside_the_graph(dev, kind, method, T):
  n, d, ei, x, p, alpha, beta, rhs = _problem(kind)
  sh = SelfShard(ei, n)
  assert sh.n_redirected > 1000 and 0 < sh.n_interior < sh.n_own
  be = D.NativeBackend(sh, d, dev, kind, p, alpha, beta, True)
  ref = R.odeint_fixed(rhs, x, T, 1.0, method)
  xd = x.to(dev)
  results = {}
  for use_graph in (False, True):
    solver = D.NativeShardedSolver(sh, be, T, 1.0, method)
    solver.y.fill_(float('nan'))                   # halo rows (and everything else) start poisoned
    with torch.no_grad():
      z = solver.integrate(xd, xd, use_graph=use_graph).clone()
      if use_graph:                                  # replay of the captured graph with a different input
        z2 = solver.integrate(2 * xd, xd, use_graph=True).clone()
        z3 = solver.integrate(xd, xd, use_graph=True).clone()
        assert torch.equal(z3, z), 'graph replay is not reproducible'
        assert not torch.equal(z2, z)
    assert solver.n_rhs_evals == (4 if method == 'rk4' else 1) * len(R.time_grid(T, 1.0)[1:])
    assert_parity(z, ref, what='self-exchange %s %s graph=%s' % (kind, method, use_graph))
    results[use_graph] = z
    solver.close()
  assert torch.equal(results[False], results[True]), 'hipGraph replay differs from eager launches'


def test_no_exchange_world1_matches_single_gpu_solver(dev):
  """A world-1 shard without halo (comm NULL): interior = all rows; must equal the single-GPU solver bit for bit."""
  n, d, ei, x, p, alpha, beta, rhs = _problem('transformer', seed=5)
  plan = D.PartitionPlan(ei, n, 1)
  sh = plan.shard(0)
  assert sh.n_halo == 0 and sh.n_interior == sh.n_own
  be = D.NativeBackend(sh, d, dev, 'transformer', p, alpha, beta, True)
  solver = D.NativeShardedSolver(sh, be, 2.0, 1.0, 'rk4')
  x_own = D.scatter_rows(x, sh).to(dev)
  with torch.no_grad():
    z = solver.integrate(x_own, x_own)
  ref = R.odeint_fixed(rhs, x, 2.0, 1.0, 'rk4')
  assert_parity(z, ref[sh.own_old_ids], what='world-1 native sharded solve')
