"""Training WITHOUT the adjoint method (opt['adjoint'] = False: the reference's default and its Cora / Citeseer best_params) on the
native recorded dopri5 -- forward = the device-controlled solve leaving the accepted steps' stage inputs on a tape, backward = one
native reverse sweep (csrc/dopri5.hip: gnpde_dopri5_set_tape / gnpde_dopri5_tape_backward) -- against
  * the reference's own gradients (tests/golden/train_*.npz: reference src/ over oracle/shims, torch autograd through torchdiffeq),
  * this package's differentiable host loop over the kernel-backed autograd Functions (opt['gnpde_host_dopri5_training']),
  * the explicit reverse sweep of oracle/tape_reverse.py in float64 on the CPU oracle.
Also: the midpoint method (run_GNN.py --method midpoint) on the native fixed-step solver."""
import importlib

import pytest
import torch

import gnpde_amd as G
from oracle import restate as R
from helpers import Fixture, fixtures, Data, assert_parity, random_graph

pytestmark = pytest.mark.gpu

O = importlib.import_module('gnpde_amd.odeint')
FUNCS = {'laplacian': G.LaplacianODEFunc, 'transformer': G.ODEFuncTransformerAtt, 'GAT': G.ODEFuncAtt}
BLOCKS = {'constant': G.ConstantODEblock, 'attention': G.AttODEblock}
OPT = dict(heads=4, attention_dim=16, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False, reweight_attention=False,
           beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9, add_source=True, no_alpha_sigmoid=False,
           mix_features=False, hidden_dim=20, augment=False, adjoint=False, tol_scale=1.0, data_norm='rw', method='dopri5', step_size=1.0,
           max_iters=100, block='attention', function='laplacian', time=2.0)


def _module_scaled(grads, refs, tol, what):
  """Every gradient within tol of the largest reference gradient of its module (a gradient that is zero in exact arithmetic --
  K.bias under a softmax over rows -- is rounding noise on both sides)."""
  scale = {}
  for k, ref in refs.items():
    mod = k.rsplit('.', 2)[0] if 'multihead' in k else k
    scale[mod] = max(scale.get(mod, 0.0), float(ref.abs().max()))
  for k, ref in refs.items():
    mod = k.rsplit('.', 2)[0] if 'multihead' in k else k
    got = grads[k]
    assert got is not None, '%s: %s received no gradient' % (what, k)
    err = float((got.detach().cpu().reshape(ref.shape) - ref.cpu()).abs().max())
    assert err <= tol * scale[mod], '%s %s: abs err %.3e against module scale %.3e (tol %.1e)' % (what, k, err, scale[mod], tol)


def _fixture_block(fx, dev, **over):
  opt = dict(fx.opt, **over)
  x = fx.t('x', dev)
  block = BLOCKS[opt['block']](FUNCS[opt['function']], [], opt, Data(x, fx.t('edge_index', dev)), dev, t=torch.tensor([0, opt['time']])).to(dev)
  block.load_state_dict(fx.params, strict=True)
  return block, x


@pytest.mark.parametrize('name', [n for n in fixtures('train_') if 'dopri5' in n])
@pytest.mark.parametrize('host_loop', [False, True])
def test_training_without_adjoint_against_the_reference(dev, name, host_loop):
  fx = Fixture(name)
  block, x = _fixture_block(fx, dev, gnpde_host_dopri5_training=host_loop)
  block.train()
  assert block.train_integrator is G.odeint
  xin = x.clone().requires_grad_(True)
  block.set_x0(xin)
  z = block(xin)
  assert z.requires_grad
  f = block.odefunc
  recorded = bool(f.__dict__.get('_tape_state'))
  assert recorded == (not host_loop), 'solve path: %s' % getattr(f, '_last_train_solve', None)
  assert f.nfe == int(fx.arr['nfe']), 'nfe %d vs reference %d' % (f.nfe, int(fx.arr['nfe']))
  assert_parity(z, fx.t('z'), max(1e-5, 2 * fx.opt['tol_scale'] * 1e-7), name + ' z')
  (z * fx.t('c', dev)).sum().backward()
  assert f.nfe == int(fx.arr['nfe_after_backward'])        # the reference's backward evaluates nothing either
  gtol = 2e-4
  assert_parity(xin.grad, fx.t('grad_x'), gtol, name + ' grad_x')
  refs = {k[5:]: fx.t(k) for k in fx.arr if k.startswith('grad/')}
  grads = dict((k, p.grad) for k, p in block.named_parameters())
  _module_scaled(grads, refs, gtol, name)
  for k, p in block.named_parameters():
    if k not in refs:
      assert p.grad is None or float(p.grad.abs().max()) == 0.0, '%s received a gradient the reference does not produce' % k


def _cora_like(dev, n, d, heads, A, seed, hubs=0, hub_deg=0, **over):
  ei = random_graph(n, 5, seed=seed, hubs=hubs, hub_deg=hub_deg)
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(seed + 1)) * 0.5
  opt = dict(OPT, hidden_dim=d, heads=heads, attention_dim=A, **over)
  block = G.AttODEblock(G.LaplacianODEFunc, [], opt, Data(x.to(dev), ei.to(dev)), dev, t=torch.tensor([0, opt['time']])).to(dev)
  g = torch.Generator().manual_seed(seed + 2)
  with torch.no_grad():
    for p in block.parameters():
      if p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      else:
        p.copy_((torch.randn(p.shape, generator=g) * 0.3).to(dev))
  return block, x, ei, opt


def _train_once(block, x, dev, c):
  for p in block.parameters():
    p.grad = None
  block.train()
  xin = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xin)
  block.odefunc.nfe = 0
  z = block(xin)
  (z * c).sum().backward()
  return z.detach(), xin.grad, {k: (None if p.grad is None else p.grad.clone()) for k, p in block.named_parameters()}, block.odefunc.nfe


@pytest.mark.parametrize('case', ['cora', 'hubs', 'd22', 'no_source'])
def test_recorded_solve_equals_the_host_loop(dev, case):
  """Same block, same weights: the recorded solve + reverse sweep against the differentiable host loop (same accept / reject
  sequence, so the same evaluation count) -- at the Cora best_params shape, with hub rows (512-entry chunks in the row kernel),
  with a width that is not a multiple of 4 (padded rows) and without the source term."""
  kw = dict(cora=dict(n=2485, d=80, heads=8, A=128, time=18.294754, tol_scale=821.977, square_plus=True, attention_norm_idx=1),
            hubs=dict(n=1500, d=32, heads=4, A=16, time=3.0, tol_scale=300.0, hubs=2, hub_deg=700),
            d22=dict(n=600, d=22, heads=2, A=8, time=2.5, tol_scale=100.0),
            no_source=dict(n=500, d=16, heads=4, A=16, time=3.0, tol_scale=500.0, add_source=False))[case]
  block, x, ei, opt = _cora_like(dev, seed=31, **kw)
  c = torch.randn(x.shape, generator=torch.Generator().manual_seed(5)).to(dev)
  z1, gx1, g1, nfe1 = _train_once(block, x, dev, c)
  assert block.odefunc.__dict__.get('_tape_state'), 'the recorded solve did not run'
  assert block.odefunc._dopri5_stats['accepted'] >= 2
  block.odefunc.opt['gnpde_host_dopri5_training'] = True
  block.reg_odefunc.odefunc.opt['gnpde_host_dopri5_training'] = True
  z2, gx2, g2, nfe2 = _train_once(block, x, dev, c)
  assert nfe1 == nfe2, (nfe1, nfe2)
  assert_parity(z1, z2, 1e-5, case + ' z')
  assert_parity(gx1, gx2, 2e-4, case + ' grad_x')
  refs = {k: v for k, v in g2.items() if v is not None}
  _module_scaled(g1, refs, 2e-4, case)
  # a second recorded iteration replays the captured trial steps and gives the same numbers bit for bit
  block.odefunc.opt['gnpde_host_dopri5_training'] = False
  z3, gx3, g3, _ = _train_once(block, x, dev, c)
  assert torch.equal(z1, z3) and torch.equal(gx1, gx3)


def test_tape_grows_when_a_solve_accepts_more_steps_than_it_holds(dev, monkeypatch):
  monkeypatch.setattr(O, '_TAPE_BUDGET_BYTES', 1)          # -> the smallest tape (8 slots)
  block, x, ei, opt = _cora_like(dev, n=400, d=16, heads=4, A=16, seed=41, time=30.0, tol_scale=1.0)
  c = torch.randn(x.shape, generator=torch.Generator().manual_seed(6)).to(dev)
  z1, gx1, g1, nfe1 = _train_once(block, x, dev, c)
  sol = next(iter(block.odefunc.__dict__['_tape_state'].values()))['solver']
  assert block.odefunc._dopri5_stats['accepted'] > 8 and sol.tape_capacity >= block.odefunc._dopri5_stats['accepted']
  block.odefunc.opt['gnpde_host_dopri5_training'] = True
  z2, gx2, g2, nfe2 = _train_once(block, x, dev, c)
  assert nfe1 == nfe2
  assert_parity(gx1, gx2, 2e-4, 'grad_x after the tape grew')


def test_two_forwards_before_one_backward_stay_differentiable(dev):
  """Round 6 (advisor item): while the record of forward A awaits its backward, forward B of the same function takes the differentiable
  host loop instead of overwriting it -- both backward passes run, as they do in the reference, and agree with a fresh single pass."""
  block, x, ei, opt = _cora_like(dev, n=300, d=16, heads=4, A=16, seed=51, time=2.0, tol_scale=100.0)
  block.train()
  xa = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xa)
  za = block(xa)
  assert str(block.odefunc._last_train_solve).startswith('native recorded')
  xb = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xb)
  zb = block(xb)
  assert str(block.odefunc._last_train_solve).startswith('differentiable host loop')
  zb.sum().backward()
  za.sum().backward()
  assert_parity(xa.grad, xb.grad, 2e-4, 'grad_x of the recorded pass vs the host-loop pass')
  # the record is free again: the next pass is recorded
  xc = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xc)
  block(xc).sum().backward()
  assert str(block.odefunc._last_train_solve).startswith('native recorded')
  # an abandoned forward pass (output dropped, no backward) does not block the record either
  xd = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xd)
  zd = block(xd)
  del zd
  import gc
  gc.collect()
  xe = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xe)
  block(xe)
  assert str(block.odefunc._last_train_solve).startswith('native recorded')


def test_stale_tape_stamp_only_ever_grows(dev, monkeypatch):
  """The safety net behind the live-record rule: the solver's tape stamp only ever grows (round-5 advisor item: `set_tape` used to reset
  it, so a backward could pass the stale-tape check after a later solve outgrew the tape).  Forced here by clearing the live-record
  marker: forward A, then a forward B that outgrows the 8-slot tape -- A's backward raises instead of differentiating B's record."""
  monkeypatch.setattr(O, '_TAPE_BUDGET_BYTES', 1)          # -> the smallest tape (8 slots)
  block, x, ei, opt = _cora_like(dev, n=400, d=16, heads=4, A=16, seed=41, time=30.0, tol_scale=1.0)
  block.train()
  xa = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xa)
  t_long = block.t.clone()
  block.t = torch.tensor([0.0, 0.5]).to(dev)              # a short solve: fits the 8 slots
  za = block(xa)
  sol = next(iter(block.odefunc.__dict__['_tape_state'].values()))['solver']
  assert sol.tape_capacity == 8
  gen_a = sol.tape_generation
  block.odefunc.__dict__.pop('_tape_live_dopri5')          # (pretend A's pass is not waiting: the case the stamp exists for)
  block.t = t_long                                         # the long one outgrows the tape
  xb = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xb)
  zb = block(xb)
  sol_b = next(iter(block.odefunc.__dict__['_tape_state'].values()))['solver']
  assert sol_b.tape_capacity > 8 and sol_b.tape_generation > gen_a
  with pytest.raises(G.GnpdeError):
    za.sum().backward()
  zb.sum().backward()
  assert torch.isfinite(xb.grad).all()


def test_recorded_gradients_against_the_float64_reverse_sweep(dev):
  """The device gradients against oracle/tape_reverse.py (float64, CPU oracle right-hand side) replaying the DEVICE's own accepted
  step sizes: independent of the float32 accept / reject decisions, so the bar is plain float32 rounding."""
  from oracle import tape_reverse as TR
  block, x, ei, opt = _cora_like(dev, n=700, d=24, heads=4, A=16, seed=61, time=4.0, tol_scale=400.0, square_plus=True, attention_norm_idx=1)
  c = torch.randn(x.shape, generator=torch.Generator().manual_seed(7))
  block.train()
  xin = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xin)
  z = block(xin)
  f = block.odefunc
  att = f.attention_weights
  att.retain_grad()
  (z * c.to(dev)).sum().backward()
  sol = next(iter(f.__dict__['_tape_state'].values()))['solver']
  S = sol.tape_steps()
  assert S == f._dopri5_stats['accepted'] >= 2
  f64 = torch.float64
  n, d = x.shape
  e_n = f.edge_index.cpu()
  w = att.detach().cpu().to(f64).mean(dim=1)
  a = torch.sigmoid(f.alpha_train.detach().cpu().to(f64))
  b = f.beta_train.detach().cpu().to(f64)
  row, col = e_n[0], e_n[1]
  x64 = x.to(f64)

  def fn(u):
    return a * (R.spmm(e_n, w, n, u) - u) + b * x64

  def vjp_u(g):
    return a * (torch.zeros_like(g).index_add_(0, col, g[row] * w.unsqueeze(1)) - g)
  # the device's accepted steps replayed in float64 (its step sizes, its end-point fraction: no second controller to disagree with)
  hs, xfrac = sol.tape_record()
  assert len(hs) == S and abs(sum(hs[:-1]) + xfrac * hs[-1] - opt['time']) <= 1e-4 * opt['time']
  out, tape = TR.dopri5_replay(fn, x64, hs, xfrac)
  acc = {'r': torch.zeros(e_n.shape[1], dtype=f64), 's_a': 0.0, 's_b': 0.0}

  def on_eval(g, u, wv):
    acc['r'] += (g[row] * u[col]).sum(dim=1)
    acc['s_a'] += float((u * wv).sum())
    acc['s_b'] += float((g * x64).sum())
  gy0 = TR.dopri5_tape_reverse(tape, c.to(f64), vjp_u, on_eval)
  assert_parity(z, out, 2e-5, 'z vs float64')
  # d/dx through the initial state alone = total - the part through the attention; compare the pieces the sweep produces
  h = att.shape[1]
  assert_parity(att.grad, ((a * acc['r']) / h).unsqueeze(1).expand(-1, h), 2e-4, 'd attention')
  assert abs(float(f.alpha_train.grad) - acc['s_a'] * float(1 - a)) <= 2e-4 * abs(acc['s_a'] * float(1 - a)) + 1e-7
  assert abs(float(f.beta_train.grad) - acc['s_b']) <= 2e-4 * abs(acc['s_b']) + 1e-7


@pytest.mark.parametrize('name', [n for n in fixtures('train_') if 'midpoint' in n])
def test_midpoint_method(dev, name):
  """run_GNN.py --method midpoint: evaluation on the native fixed-step solver (one hipGraph, two LINCOMB stages per step), training
  on the recorded solve + reverse sweep (round 6; the host loop in test_fixed_grid_training_against_the_reference); both against the
  reference's block."""
  fx = Fixture(name)
  block, x = _fixture_block(fx, dev)
  block.eval()
  block.set_x0(x)
  with torch.no_grad():
    z = block(x)
  assert block.odefunc.__dict__.get('_solver_state'), 'native solver was not used'
  assert_parity(z, fx.t('z_eval'), 1e-5, name + ' eval')
  assert block.odefunc.nfe == int(fx.arr['nfe'])
  with torch.no_grad():
    z_eager = G.odeint(block.odefunc, x, torch.tensor([0, fx.opt['time']], device=dev), method='midpoint', options={'step_size': fx.opt['step_size']},
                       use_graph=False)[1]
  assert torch.equal(z, z_eager)
  block.train()
  block.odefunc.nfe = 0
  xin = x.clone().requires_grad_(True)
  block.set_x0(xin)
  zt = block(xin)
  assert_parity(zt, fx.t('z'), 1e-5, name + ' z')
  (zt * fx.t('c', dev)).sum().backward()
  assert_parity(xin.grad, fx.t('grad_x'), 2e-4, name + ' grad_x')
  refs = {k[5:]: fx.t(k) for k in fx.arr if k.startswith('grad/')}
  _module_scaled(dict((k, p.grad) for k, p in block.named_parameters()), refs, 2e-4, name)


# ---- round 6: fixed-grid training without the adjoint method = recorded native solve + native reverse sweep -------------------
FIXED = [n for n in fixtures('train_') if 'dopri5' not in n]


@pytest.mark.parametrize('name', FIXED)
@pytest.mark.parametrize('host_loop', [False, True])
def test_fixed_grid_training_against_the_reference(dev, name, host_loop):
  """`run_GNN.py --method rk4 / euler / midpoint` with the adjoint off (the reference's default): output, evaluation count and every
  gradient of the reference's own block (torch autograd through the restated torchdiffeq fixed-grid loop), on the recorded native solve
  + reverse sweep (csrc/solver.hip gnpde_solver_set_tape, csrc/adjoint.hip gnpde_adjoint_set_tape) and on the host loop."""
  fx = Fixture(name)
  assert len(FIXED) >= 8
  block, x = _fixture_block(fx, dev, gnpde_host_fixed_training=host_loop)
  block.train()
  assert block.train_integrator is G.odeint
  xin = x.clone().requires_grad_(True)
  block.set_x0(xin)
  z = block(xin)
  assert z.requires_grad
  f = block.odefunc
  recorded = str(getattr(f, '_last_train_solve', '')).startswith('native recorded fixed-grid')
  assert recorded == (not host_loop), 'solve path: %s' % getattr(f, '_last_train_solve', None)
  assert f.nfe == int(fx.arr['nfe']), 'nfe %d vs reference %d' % (f.nfe, int(fx.arr['nfe']))
  assert_parity(z, fx.t('z'), 1e-5, name + ' z')
  (z * fx.t('c', dev)).sum().backward()
  assert f.nfe == int(fx.arr['nfe_after_backward'])        # the reference's backward evaluates nothing either
  gtol = 2e-4
  assert_parity(xin.grad, fx.t('grad_x'), gtol, name + ' grad_x')
  refs = {k[5:]: fx.t(k) for k in fx.arr if k.startswith('grad/')}
  grads = dict((k, p.grad) for k, p in block.named_parameters())
  _module_scaled(grads, refs, gtol, name)
  for k, p in block.named_parameters():
    if k not in refs:
      assert p.grad is None or float(p.grad.abs().max()) == 0.0, '%s received a gradient the reference does not produce' % k


def _fixed_block(dev, kind, block_kind, n, d, heads, A, seed, method, time, step_size=1.0, hubs=0, hub_deg=0, **over):
  ei = random_graph(n, 5, seed=seed, hubs=hubs, hub_deg=hub_deg)
  g = torch.Generator().manual_seed(seed)
  x = torch.randn(n, d, generator=g)
  opt = dict(OPT, hidden_dim=d, heads=heads, attention_dim=A, function=kind, block=block_kind, method=method, time=time, step_size=step_size, **over)
  block = BLOCKS[block_kind](FUNCS[kind], [], opt, Data(x.to(dev), ei.to(dev)), dev, t=torch.tensor([0, time])).to(dev)
  with torch.no_grad():
    for p in block.parameters():
      if p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      else:
        p.copy_((torch.randn(p.shape, generator=g) * 0.3).to(dev))
    for f in (block.odefunc, block.reg_odefunc.odefunc):
      lay = getattr(f, 'multihead_att_layer', None)
      if lay is not None and hasattr(lay, 'lengthscale'):       # exp kernel: a length scale near zero would underflow every score
        lay.lengthscale.fill_(1.7)
        lay.output_var.fill_(0.8)
  return block, x


FIXED_CASES = {
  # GRAND-nl at a Cora-like shape (A = 128 / 8 heads, squareplus over columns), rk4 with a short last step
  'nl_cora_rk4': dict(kind='transformer', block_kind='constant', n=2485, d=80, heads=8, A=128, method='rk4', time=4.3,
                      square_plus=True, attention_norm_idx=1),
  # hub rows (512-entry chunks in the row kernel and in the attention backward)
  'nl_hubs_rk4': dict(kind='transformer', block_kind='constant', n=1500, d=32, heads=4, A=16, method='rk4', time=3.0, hubs=2, hub_deg=700),
  # a width that is not a multiple of 4 (padded rows), euler, no source term
  'nl_d22_euler': dict(kind='transformer', block_kind='constant', n=600, d=22, heads=2, A=8, method='euler', time=2.5, step_size=0.5, add_source=False),
  'nl_midpoint': dict(kind='transformer', block_kind='constant', n=700, d=64, heads=4, A=32, method='midpoint', time=2.2, step_size=0.4),
  # GRAND-l under the attention block: the edge weights carry gradients into the attention layer
  'l_attention_rk4_hubs': dict(kind='laplacian', block_kind='attention', n=1500, d=32, heads=4, A=16, method='rk4', time=3.5, hubs=2, hub_deg=700),
  'l_attention_midpoint_d22': dict(kind='laplacian', block_kind='attention', n=600, d=22, heads=2, A=8, method='midpoint', time=2.0, step_size=0.5),
  'l_constant_euler': dict(kind='laplacian', block_kind='constant', n=900, d=48, heads=4, A=16, method='euler', time=4.0),
  # cosine_sim / pearson scores and the raw alpha of opt['no_alpha_sigmoid'] on the native VJP stage
  'nl_pearson_rk4': dict(kind='transformer', block_kind='constant', n=800, d=32, heads=4, A=32, method='rk4', time=2.3, attention_type='pearson'),
  'nl_cosine_cols_euler': dict(kind='transformer', block_kind='constant', n=800, d=32, heads=2, A=32, method='euler', time=2.0, step_size=0.5,
                               attention_type='cosine_sim', attention_norm_idx=1, square_plus=True),
  'nl_raw_alpha_rk4': dict(kind='transformer', block_kind='constant', n=800, d=32, heads=4, A=16, method='rk4', time=2.0, no_alpha_sigmoid=True),
  'nl_exp_kernel_rk4': dict(kind='transformer', block_kind='constant', n=800, d=32, heads=4, A=16, method='rk4', time=2.3, attention_type='exp_kernel'),
  'gat_rk4_hubs': dict(kind='GAT', block_kind='constant', n=1500, d=32, heads=4, A=16, method='rk4', time=2.3, hubs=2, hub_deg=700),
  'gat_midpoint_cols': dict(kind='GAT', block_kind='constant', n=700, d=48, heads=2, A=32, method='midpoint', time=2.0, step_size=0.5, attention_norm_idx=1),
  'l_attention_raw_alpha_rk4': dict(kind='laplacian', block_kind='attention', n=800, d=32, heads=4, A=16, method='rk4', time=2.0, no_alpha_sigmoid=True),
}


@pytest.mark.parametrize('case', sorted(FIXED_CASES))
def test_recorded_fixed_grid_equals_the_host_loop(dev, case):
  """Same block, same weights: recorded solve + reverse sweep against the differentiable host loop over the kernel-backed autograd
  Functions; a second recorded iteration replays both captured graphs and reproduces the first bit for bit."""
  kw = dict(FIXED_CASES[case])
  block, x = _fixed_block(dev, seed=61, **kw)
  c = torch.randn(x.shape, generator=torch.Generator().manual_seed(7)).to(dev)
  z1, gx1, g1, nfe1 = _train_once(block, x, dev, c)
  assert str(block.odefunc._last_train_solve).startswith('native recorded fixed-grid'), block.odefunc._last_train_solve
  block.odefunc.opt['gnpde_host_fixed_training'] = True
  block.reg_odefunc.odefunc.opt['gnpde_host_fixed_training'] = True
  z2, gx2, g2, nfe2 = _train_once(block, x, dev, c)
  assert nfe1 == nfe2, (nfe1, nfe2)
  assert_parity(z1, z2, 1e-5, case + ' z')
  assert_parity(gx1, gx2, 2e-4, case + ' grad_x')
  refs = {k: v for k, v in g2.items() if v is not None}
  assert refs, 'the host loop produced no parameter gradients'
  _module_scaled(g1, refs, 2e-4, case)
  block.odefunc.opt['gnpde_host_fixed_training'] = False
  z3, gx3, g3, _ = _train_once(block, x, dev, c)
  assert torch.equal(z1, z3) and torch.equal(gx1, gx3)
  for k, v in g1.items():
    if v is not None:
      assert torch.equal(v, g3[k]), k


def test_recorded_fixed_grid_on_the_relabelled_graph(dev):
  """opt['gnpde_reorder'] = 'degree': forward and reverse sweep run on graph.LocalityView (nodes relabelled); output and gradients come
  back in the caller's order and agree with the run on the graph as given."""
  kw = dict(FIXED_CASES['nl_hubs_rk4'])
  block, x = _fixed_block(dev, seed=62, **kw)
  c = torch.randn(x.shape, generator=torch.Generator().manual_seed(8)).to(dev)
  block.odefunc.opt['gnpde_reorder'] = '0'
  z1, gx1, g1, _ = _train_once(block, x, dev, c)
  block.odefunc.opt['gnpde_reorder'] = 'degree'
  z2, gx2, g2, _ = _train_once(block, x, dev, c)
  ent = next(iter(block.odefunc.__dict__['_fixed_tape_state'].values()))
  assert ent['view'] is not None
  assert_parity(z2, z1, 1e-6, 'z relabelled')
  assert_parity(gx2, gx1, 2e-5, 'grad_x relabelled')
  _module_scaled(g2, {k: v for k, v in g1.items() if v is not None}, 2e-5, 'relabelled')


def test_fixed_grid_two_forwards_before_one_backward(dev):
  """While the record of forward A awaits its backward, forward B takes the host loop; both backward passes run and agree.  With the
  live-record marker cleared (the case the tape stamp exists for) A's backward after B's recorded forward raises."""
  block, x = _fixed_block(dev, seed=63, **FIXED_CASES['nl_d22_euler'])
  block.train()
  xa = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xa)
  za = block(xa)
  xb = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xb)
  zb = block(xb)
  assert str(block.odefunc._last_train_solve).startswith('differentiable host loop')
  zb.sum().backward()
  za.sum().backward()
  assert_parity(xa.grad, xb.grad, 2e-4, 'grad_x recorded vs host loop')
  xc = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xc)
  zc = block(xc)
  assert str(block.odefunc._last_train_solve).startswith('native recorded fixed-grid')
  block.odefunc.__dict__.pop('_tape_live_fixed')
  xd = x.to(dev).clone().requires_grad_(True)
  block.set_x0(xd)
  zd = block(xd)
  assert str(block.odefunc._last_train_solve).startswith('native recorded fixed-grid')
  zd.sum().backward()
  with pytest.raises(G.GnpdeError):
    zc.sum().backward()


def test_recorded_fixed_grid_falls_back_to_the_host_loop_beyond_its_memory_budget(dev, monkeypatch):
  """A record that would not fit (budget: 96 GiB or 70 % of the free device memory) is never allocated: the differentiable host loop runs
  instead and says so."""
  monkeypatch.setattr(O, '_FIXED_TAPE_BUDGET_BYTES', 1 << 16)
  block, x = _fixed_block(dev, seed=64, **FIXED_CASES['nl_d22_euler'])
  c = torch.randn(x.shape, generator=torch.Generator().manual_seed(9)).to(dev)
  z, gx, g, nfe = _train_once(block, x, dev, c)
  assert str(block.odefunc._last_train_solve).startswith('differentiable host loop'), block.odefunc._last_train_solve
  assert not block.odefunc.__dict__.get('_fixed_tape_state')
  assert torch.isfinite(gx).all()


@pytest.mark.parametrize('case', ['nl_hubs_rk4', 'nl_d22_euler', 'nl_midpoint', 'l_attention_rk4_hubs', 'gat_rk4_hubs', 'nl_pearson_rk4'])
def test_cotangent_side_sweep_equals_the_first_form(dev, case):
  """Round 6: the reverse sweep gathers the COTANGENT rows only (row kernel on the transposed graph with the roles exchanged, then one pass
  that closes the stage algebra with S + P) instead of gathering the recorded state rows again and running a second aggregation.  The
  same products; the row sums of A^T u_a come from another kernel (another grouping of a row's entries), so the two forms
  (gnpde_tune(15, 1): the first) agree to rounding (2e-6 for dL/dx and the weight gradients' modules, 5e-5 for the scalars), not bit for bit."""
  from gnpde_amd import ops
  kw = dict(FIXED_CASES[case])
  c = None
  res = []
  for knob in (0, 1):
    ops.tune(15, knob)
    try:
      block, x = _fixed_block(dev, seed=66, **kw)
      if c is None:
        c = torch.randn(x.shape, generator=torch.Generator().manual_seed(10)).to(dev)
      z, gx, g, nfe = _train_once(block, x, dev, c)
      ent = next(iter(block.odefunc.__dict__['_fixed_tape_state'].values()))
      res.append((z, gx, g, bool(ent['sweep'].swapped)))
    finally:
      ops.tune(15, 0)
  (z1, gx1, g1, sw1), (z0, gx0, g0, sw0) = res
  assert sw1 and not sw0, (sw1, sw0)
  assert torch.equal(z1, z0)
  assert_parity(gx1, gx0, 2e-6, case + ' grad_x')
  refs = {k: v for k, v in g0.items() if v is not None}
  _module_scaled(g1, refs, 5e-5, case)      # (the scalar gradients are sums over all rows: the first form subtracts beta <u_a, x0> back out of a float sum)


def test_recorded_dopri5_tape_growth_is_capped(dev, monkeypatch):
  """A solve whose accepted steps would need a tape beyond the cap runs the differentiable host loop instead of doubling without bound
  (round-5 advisor item)."""
  monkeypatch.setattr(O, '_TAPE_BUDGET_BYTES', 1)          # -> the smallest tape (8 slots)
  monkeypatch.setattr(O, '_TAPE_GROWTH_CAP_BYTES', 1 << 10)
  block, x, ei, opt = _cora_like(dev, n=400, d=16, heads=4, A=16, seed=41, time=30.0, tol_scale=1.0)
  c = torch.randn(x.shape, generator=torch.Generator().manual_seed(6)).to(dev)
  z1, gx1, g1, nfe1 = _train_once(block, x, dev, c)
  assert str(block.odefunc._last_train_solve).startswith('differentiable host loop'), block.odefunc._last_train_solve
  monkeypatch.setattr(O, '_TAPE_GROWTH_CAP_BYTES', 96 << 30)
  z2, gx2, g2, nfe2 = _train_once(block, x, dev, c)
  assert str(block.odefunc._last_train_solve).startswith('native recorded')
  assert nfe1 == nfe2
  assert_parity(gx1, gx2, 2e-4, 'grad_x host loop vs recorded')
