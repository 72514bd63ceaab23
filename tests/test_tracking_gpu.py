"""Projection tracking of the fixed-step solver (gnpde_solver_set_projection_refresh; csrc/attention.hip TrackArgs): the q||k
table of the next stage input comes out of the row-attention kernel instead of a projection launch.  Replaces the nn.Linear Q
and K of every evaluation (reference src/function_transformer_attention.py:174-175).  The tracked solve must agree with the
solve that projects in every evaluation (refresh 0: the path the golden fixtures pin) to rounding, and with the CPU oracle to
the 1e-5 bar, for every refresh interval."""
import pytest
import torch

import gnpde_amd as G
from oracle import restate as R
from helpers import Data, assert_parity, parity, random_graph

pytestmark = pytest.mark.gpu

BASE = dict(heads=4, attention_dim=16, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False,
            reweight_attention=False, beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9,
            add_source=True, no_alpha_sigmoid=False, mix_features=False, hidden_dim=64, augment=False, adjoint=False,
            tol_scale=1.0, data_norm='rw', method='rk4', step_size=1.0, max_iters=100, block='constant',
            function='transformer', time=5.0)


def _cpu(t):
  return t.detach().cpu()


def _block(opt, ei, n, x, dev, seed=0, bias=True):
  block = G.ConstantODEblock(G.ODEFuncTransformerAtt, [], dict(opt), Data(x, ei), dev, t=torch.tensor([0, opt['time']])).to(dev)
  g = torch.Generator().manual_seed(seed)
  with torch.no_grad():
    for name, p in block.named_parameters():
      if 'multihead_att_layer' in name and p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      elif 'multihead_att_layer' in name and name.endswith('bias'):
        p.copy_((0.3 * torch.randn(p.shape, generator=g) if bias else torch.zeros(p.shape)).to(dev))
    for f in (block.odefunc, block.reg_odefunc.odefunc):
      f.alpha_train.fill_(0.3)
      f.beta_train.fill_(0.15)
  block.eval()
  return block


def _oracle(block, x0, T, method):
  f, lay, o = block.odefunc, block.odefunc.multihead_att_layer, block.opt
  edge = _cpu(f.edge_index)
  rhs = lambda t, y: R.rhs_transformer(y, edge, _cpu(lay.Q.weight), _cpu(lay.Q.bias), _cpu(lay.K.weight), _cpu(lay.K.bias),  # noqa: E731
                                       lay.h, _cpu(f.alpha_train), _cpu(f.beta_train), x0, o['no_alpha_sigmoid'], o['add_source'])
  return R.odeint_fixed(rhs, x0, T, o['step_size'], method)


def _solve(opt, ei, n, x, dev, refresh, bias=True):
  block = _block(dict(opt, gnpde_projection_refresh=refresh), ei.to(dev), n, x.to(dev), dev, bias=bias)
  block.set_x0(x.to(dev))
  with torch.no_grad():
    z = block(x.to(dev))
  solver = next(iter(block.odefunc._solver_state.values()))['solver']
  return block, z, solver.projection_refresh


@pytest.mark.parametrize('heads,att_dim', [(4, 16), (2, 16), (1, 16), (8, 32), (4, 64), (1, 4)])
@pytest.mark.parametrize('method', ['rk4', 'euler'])
def test_tracked_solve_matches_per_evaluation_projection(dev, heads, att_dim, method):
  """Hub rows (two rows of 800 entries -> chunked), duplicate entries, biases, source term: refresh 1, 2 and 'never' against
  the solve that projects in every evaluation, and against the oracle."""
  n, d = 2500, 64
  ei = random_graph(n, 6, seed=11, hubs=2, hub_deg=800, dup=40)
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(12))
  opt = dict(BASE, heads=heads, attention_dim=att_dim, method=method, time=5.0 if method == 'rk4' else 7.0)
  block0, z0, r0 = _solve(opt, ei, n, x, dev, 0)
  assert r0 == 0
  ref = _oracle(block0, x, opt['time'], method)
  assert_parity(z0, ref, what='projection in every evaluation')
  for refresh in (1, 2, 1000):
    _, z, r = _solve(opt, ei, n, x, dev, refresh)
    assert r == refresh, 'tracking is not active (got %d): the test would be vacuous' % r
    e_inf, e_2 = parity(z, z0)
    assert e_inf < 3e-6 and e_2 < 3e-6, 'refresh %d differs from the per-evaluation projection: %.2e / %.2e' % (refresh, e_inf, e_2)
    assert_parity(z, ref, what='tracked, refresh %d' % refresh)


@pytest.mark.parametrize('add_source,no_sigmoid,bias', [(False, False, True), (True, True, False), (False, True, True)])
def test_tracked_solve_option_combinations(dev, add_source, no_sigmoid, bias):
  n, d = 1800, 128
  ei = random_graph(n, 9, seed=21, hubs=1, hub_deg=1300)
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(22))
  opt = dict(BASE, hidden_dim=d, add_source=add_source, no_alpha_sigmoid=no_sigmoid, time=3.4, step_size=0.5)   # short last step
  block0, z0, _ = _solve(opt, ei, n, x, dev, 0, bias=bias)
  ref = _oracle(block0, x, opt['time'], 'rk4')
  _, z, r = _solve(opt, ei, n, x, dev, None, bias=bias)       # library default
  assert r == 1, 'the default is a fresh projection at every step'
  assert_parity(z, ref, what='tracked default')
  e_inf, _ = parity(z, z0)
  assert e_inf < 3e-6


def test_graph_with_an_empty_row_is_not_tracked(dev):
  """A row without entries is in no degree class, so its table row would never be written: such graphs keep the projection
  launch (gnpde_solver_tracks_projection reports 0) and still match the oracle."""
  n, d = 600, 32
  ei = random_graph(n, 5, seed=31, isolated=3, loops=False)
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(32))
  opt = dict(BASE, hidden_dim=d, self_loop_weight=0, time=3.0)
  block, z, r = _solve(opt, ei, n, x, dev, 1)
  deg = torch.bincount(_cpu(block.odefunc.edge_index)[0], minlength=n)
  if int((deg == 0).sum()) == 0:
    pytest.skip('the block added self loops: no empty row')
  assert r == 0
  assert_parity(z, _oracle(block, x, opt['time'], 'rk4'), what='untracked (empty rows)')


def test_tracked_solve_is_reproducible_and_graph_equals_eager(dev):
  import functools
  n, d = 3000, 128
  ei = random_graph(n, 8, seed=41, hubs=2, hub_deg=900)
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(42)).to(dev)
  block = _block(dict(BASE, hidden_dim=d, time=6.0, gnpde_projection_refresh=3), ei.to(dev), n, x, dev)
  block.set_x0(x)
  with torch.no_grad():
    a = block(x).clone()
    b = block(x).clone()
    block.test_integrator = functools.partial(G.odeint, use_graph=False)
    c = block(x).clone()
  assert torch.equal(a, b) and torch.equal(a, c)
