"""The native adjoint solve (csrc/adjoint.hip: the whole backward interval as one hipGraph) against the stage-by-stage loop of
odeint._adjoint_fixed_grid, which the reference-gradient fixtures of test_adjoint_gpu.py pin to torchdiffeq's odeint_adjoint
(reference src/base_classes.py:44-47, src/block_constant.py:45-55).  Same blocks, same inputs, both paths; and the native path
itself against float64 autograd through the oracle's right-hand side for one small case."""
import pytest
import torch

import gnpde_amd as G
from helpers import Data, assert_parity, random_graph

pytestmark = pytest.mark.gpu


def _opt(**over):
  opt = dict(heads=4, attention_dim=16, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False,
             reweight_attention=False, beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9,
             add_source=True, no_alpha_sigmoid=False, mix_features=False, hidden_dim=32, augment=False, adjoint=True,
             adjoint_method='rk4', adjoint_step_size=1.0, tol_scale=1.0, tol_scale_adjoint=1.0, data_norm='rw',
             method='rk4', step_size=1.0, max_iters=100, block='constant', function='transformer', time=3.0,
             att_samp_pct=1.0, use_flux=False)
  opt.update(over)
  return opt


def _run(dev, opt, ei, x, seed, host):
  o = dict(opt, gnpde_host_adjoint=bool(host))
  fcls = {'transformer': G.ODEFuncTransformerAtt, 'laplacian': G.LaplacianODEFunc, 'GAT': G.ODEFuncAtt}[o['function']]
  bcls = {'constant': G.ConstantODEblock, 'attention': G.AttODEblock}[o['block']]
  block = bcls(fcls, [], o, Data(x, ei), dev, t=torch.tensor([0, o['time']])).to(dev)
  g = torch.Generator().manual_seed(seed)
  with torch.no_grad():
    for name, p in block.named_parameters():
      if p.dim() >= 2 and 'multihead_att_layer' in name:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      elif name.endswith('.bias') and 'multihead_att_layer' in name:
        p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
    for f in (block.odefunc, block.reg_odefunc.odefunc):
      f.alpha_train.fill_(0.3)
      f.beta_train.fill_(0.2)
      lay = getattr(f, 'multihead_att_layer', None)
      if lay is not None and hasattr(lay, 'lengthscale'):       # exp kernel: scalars away from their initial 1
        lay.lengthscale.fill_(1.7)
        lay.output_var.fill_(0.8)
  block.train()
  xin = x.clone().requires_grad_(True)
  block.set_x0(xin)
  z = block(xin)
  c = torch.randn(z.shape, generator=torch.Generator().manual_seed(seed + 7)).to(dev)
  (z * c).sum().backward()
  grads = {k: p.grad.detach().clone() for k, p in block.named_parameters() if p.grad is not None}
  used_native = bool(block.odefunc.__dict__.get('_adjoint_state'))
  return z.detach(), xin.grad.detach().clone(), grads, used_native, block.odefunc.nfe


CASES = {
  'nl_rk4': dict(),
  'nl_euler': dict(adjoint_method='euler', adjoint_step_size=0.5, method='euler', step_size=0.5, time=2.0),
  'nl_short_last_step': dict(time=2.3),
  'nl_no_source': dict(add_source=False),
  'nl_squareplus_cols': dict(square_plus=True, attention_norm_idx=1, attention_dim=32, heads=2),
  'nl_softmax_cols': dict(attention_norm_idx=1),
  'nl_heads8_dk16': dict(attention_dim=128, heads=8, hidden_dim=80),
  'nl_d162_padded': dict(hidden_dim=162, attention_dim=32, heads=2),
  'nl_d256': dict(hidden_dim=256, attention_dim=64, heads=4, time=2.0),
  # weight gradients on the matrix cores (adjoint_gram_mfma_kernel: 2 A in {16, 32, 64}, d a multiple of 64)
  'nl_d128_mfma': dict(hidden_dim=128, attention_dim=16, heads=4),                      # the ogbn-arxiv shape: <2, 2>
  'nl_d64_m16_mfma': dict(hidden_dim=64, attention_dim=8, heads=2, time=2.3),           # <1, 1>, short last step
  'nl_d128_m64_mfma': dict(hidden_dim=128, attention_dim=32, heads=2),                  # <4, 2>
  'nl_d192_mfma': dict(hidden_dim=192, attention_dim=16, heads=4, add_source=False),    # <2, 3>
  'nl_d256_m16_mfma': dict(hidden_dim=256, attention_dim=8, heads=1, time=2.0),         # <1, 4>
  # round 6: cosine_sim / pearson scores (unit head vectors: normalisation + its backward around the scaled-dot kernels) and the raw
  # alpha of opt['no_alpha_sigmoid']
  'nl_cosine': dict(attention_type='cosine_sim'),
  'nl_cosine_squareplus_cols_dk16': dict(attention_type='cosine_sim', square_plus=True, attention_norm_idx=1, attention_dim=32, heads=2),
  'nl_pearson': dict(attention_type='pearson', attention_dim=32, heads=4, time=2.3),
  'nl_pearson_d128_mfma': dict(attention_type='pearson', hidden_dim=128, attention_dim=16, heads=4, add_source=False),
  'nl_raw_alpha': dict(no_alpha_sigmoid=True),
  'nl_raw_alpha_cols_euler': dict(no_alpha_sigmoid=True, attention_norm_idx=1, adjoint_method='euler', adjoint_step_size=0.5),
  'l_raw_alpha': dict(function='laplacian', no_alpha_sigmoid=True),
  # exp_kernel scores: d q / d k through the squared distance, d output_var and d lengthscale
  'nl_exp_kernel': dict(attention_type='exp_kernel'),
  'nl_exp_kernel_cols_squareplus': dict(attention_type='exp_kernel', attention_norm_idx=1, square_plus=True, attention_dim=32, heads=2, time=2.3),
  # the GAT function (reference src/function_GAT_attention.py) on the native stage
  'gat_rk4': dict(function='GAT'),
  'gat_cols_euler_h8': dict(function='GAT', attention_norm_idx=1, attention_dim=64, heads=8, adjoint_method='euler', adjoint_step_size=0.5),
  'gat_d128_no_source': dict(function='GAT', hidden_dim=128, attention_dim=32, heads=2, add_source=False, time=2.3),
  'l_rk4': dict(function='laplacian'),
  'l_euler': dict(function='laplacian', adjoint_method='euler', adjoint_step_size=1.0),
  'l_attention_block': dict(function='laplacian', block='attention'),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_native_adjoint_matches_stage_loop(dev, case):
  opt = _opt(**CASES[case])
  hubs = 2 if case in ('nl_rk4', 'l_rk4', 'nl_d162_padded', 'nl_heads8_dk16', 'nl_d128_mfma', 'gat_rk4', 'nl_exp_kernel') else 0     # (d = 80 with hubs: the row-pair kernel's chunk and long-row paths)
  n = 21000 if case == 'nl_d128_mfma' else 700       # (21000 rows: 42-row slabs -- several K steps per wave, ragged last steps)
  ei = random_graph(n, 6, seed=11, hubs=hubs, hub_deg=700, isolated=3, dup=20).to(dev)
  x = (0.5 * torch.randn(n, opt['hidden_dim'], generator=torch.Generator().manual_seed(3))).to(dev)
  z_h, gx_h, gp_h, native_h, nfe_h = _run(dev, opt, ei, x, 5, host=True)
  z_n, gx_n, gp_n, native_n, nfe_n = _run(dev, opt, ei, x, 5, host=False)
  assert not native_h and native_n, 'path selection: host %s native %s' % (native_h, native_n)
  assert nfe_h == nfe_n
  assert torch.equal(z_h, z_n)
  tol = 2e-4     # long reductions in a different order (same bound as test_adjoint_gpu.py)
  assert_parity(gx_n, gx_h, tol, case + ' grad_x')
  assert set(gp_n) == set(gp_h)
  checked = 0
  scale = max(float(v.abs().max()) for v in gp_h.values())
  for k in sorted(gp_h):
    ref = gp_h[k]
    if float(ref.abs().max()) < 1e-5 * max(scale, 1.0):      # mathematically zero (a row softmax does not see the key bias): rounding noise on both sides
      assert float(gp_n[k].abs().max()) < 1e-4 * max(scale, 1.0), k
    else:
      assert_parity(gp_n[k], ref, tol, case + ' ' + k)
      checked += 1
  assert checked >= (2 if opt['function'] == 'laplacian' and opt['add_source'] else 1)


def test_native_adjoint_replays_and_follows_parameter_updates(dev):
  """A second backward replays the captured graph; after an optimiser-like in-place update of the parameters and a new x0 the
  same solver object must produce the gradients of the NEW parameters (weights are read through stable pointers)."""
  opt = _opt()
  n = 500
  ei = random_graph(n, 5, seed=2).to(dev)
  x = (0.5 * torch.randn(n, opt['hidden_dim'], generator=torch.Generator().manual_seed(1))).to(dev)

  def grads_of(block, xin):
    for p in block.parameters():
      p.grad = None
    block.set_x0(xin)
    z = block(xin)
    z.pow(2).sum().backward()
    return xin.grad.detach().clone(), {k: p.grad.detach().clone() for k, p in block.named_parameters() if p.grad is not None}

  blocks = {}
  for host in (True, False):
    block = G.ConstantODEblock(G.ODEFuncTransformerAtt, [], dict(opt, gnpde_host_adjoint=host), Data(x, ei), dev,
                               t=torch.tensor([0, opt['time']])).to(dev)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
      for name, p in block.named_parameters():
        if p.dim() >= 2 and 'multihead_att_layer' in name:
          p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
        elif name.endswith('.bias') and 'multihead_att_layer' in name:     # (nn.Linear draws its default bias from the global RNG)
          p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
    block.train()
    blocks[host] = block
  for it in range(3):
    res = {}
    for host in (True, False):
      xin = (x * (1.0 + 0.1 * it)).clone().requires_grad_(True)
      res[host] = grads_of(blocks[host], xin)
      with torch.no_grad():          # the same in-place "optimiser step" on both
        for name, p in blocks[host].named_parameters():
          if 'multihead_att_layer.Q' in name or 'multihead_att_layer.K' in name or name.endswith('alpha_train'):
            p.add_(0.01 * (it + 1))
    assert_parity(res[False][0], res[True][0], 2e-4, 'iteration %d grad_x' % it)
    for k in res[True][1]:
      if float(res[True][1][k].abs().max()) > 1e-4:
        assert_parity(res[False][1][k], res[True][1][k], 2e-4, 'iteration %d %s' % (it, k))
  sol = blocks[False].odefunc.__dict__['_adjoint_state']
  assert len(sol) == 1, 'one live adjoint solver per function object'


def test_native_adjoint_against_float64_autograd(dev):
  """Independent of the stage loop: d/dx0 and d/dtheta of sum(c * z(T)) by float64 autograd through the oracle's f and a
  plain rk4 (3/8) loop -- the discretise-then-optimise gradient, which the adjoint solve approaches as O(h^4); at this step
  size and smooth dynamics the two agree to ~1e-3, the bound used for the adaptive fixtures."""
  from oracle import restate as R
  opt = _opt(time=2.0, step_size=0.25, adjoint_step_size=0.25, hidden_dim=16, attention_dim=8, heads=2)
  n = 120
  ei = random_graph(n, 4, seed=4).to(dev)
  x = (0.3 * torch.randn(n, 16, generator=torch.Generator().manual_seed(8))).to(dev)
  z, gx, gp, native, _ = _run(dev, opt, ei, x, 5, host=False)
  assert native
  # float64 reference on the CPU
  block = G.ConstantODEblock(G.ODEFuncTransformerAtt, [], dict(opt), Data(x, ei), dev, t=torch.tensor([0, opt['time']])).to(dev)
  g = torch.Generator().manual_seed(5)
  with torch.no_grad():
    for name, p in block.named_parameters():
      if p.dim() >= 2 and 'multihead_att_layer' in name:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      elif name.endswith('.bias') and 'multihead_att_layer' in name:
        p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
  f = block.odefunc
  lay = f.multihead_att_layer
  dd = lambda t: t.detach().double().cpu()
  wq, bq, wk, bk = (dd(t).requires_grad_(True) for t in (lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias))
  al = torch.tensor(0.3, dtype=torch.float64, requires_grad=True)
  be = torch.tensor(0.2, dtype=torch.float64, requires_grad=True)
  x64 = dd(x).requires_grad_(True)
  x0_64 = dd(x)                      # the source term is a detached copy (ODEblock.set_x0): no gradient through it
  edge = f.edge_index.cpu()
  rhs = lambda y: R.rhs_transformer(y, edge, wq, bq, wk, bk, lay.h, al, be, x0_64, False, True)
  y = x64
  h = 0.25
  for _ in range(8):
    k1 = rhs(y)
    k2 = rhs(y + h * k1 / 3)
    k3 = rhs(y + h * (k2 - k1 / 3))
    k4 = rhs(y + h * (k1 - k2 + k3))
    y = y + h * (k1 + 3 * (k2 + k3) + k4) / 8
  c = torch.randn(z.shape, generator=torch.Generator().manual_seed(5 + 7)).double()
  (y * c).sum().backward()
  assert_parity(z, y.detach().float(), 1e-5, 'z vs float64 oracle')
  assert_parity(gx, x64.grad.float(), 2e-3, 'grad_x vs float64 autograd')
  assert_parity(gp['odefunc.multihead_att_layer.Q.weight'], wq.grad.float(), 2e-3, 'dWq')
  assert_parity(gp['odefunc.multihead_att_layer.K.weight'], wk.grad.float(), 2e-3, 'dWk')
  assert_parity(gp['odefunc.alpha_train'].reshape(()), al.grad.float(), 2e-3, 'dalpha')
  assert_parity(gp['odefunc.beta_train'].reshape(()), be.grad.float(), 2e-3, 'dbeta')


ADAPTIVE = {
  'pubmed_like_heun': dict(block='attention', function='laplacian', method='dopri5', tol_scale=500.0, adjoint_method='adaptive_heun',
                           tol_scale_adjoint=3000.0, time=4.0, attention_type='cosine_sim', heads=1, attention_dim=16),
  'coauthor_like_dopri5': dict(block='attention', function='laplacian', method='dopri5', tol_scale=2000.0, adjoint_method='dopri5',
                               tol_scale_adjoint=1500.0, time=3.0, add_source=False, attention_norm_idx=1, square_plus=True),
  'constant_dopri5_tight': dict(block='constant', function='laplacian', method='dopri5', tol_scale=10.0, adjoint_method='dopri5',
                                tol_scale_adjoint=10.0, time=2.0),
  'constant_heun_d22': dict(block='constant', function='laplacian', method='rk4', adjoint_method='adaptive_heun', tol_scale_adjoint=200.0,
                            time=2.0, hidden_dim=22),
}


@pytest.mark.parametrize('name', sorted(ADAPTIVE))
def test_adaptive_adjoint_native_stages_match_the_flat_host_loop(dev, name):
  """adjoint_method adaptive_heun (the reference's default; best_params Pubmed) / dopri5 (CoauthorCS, Computers) on the Laplacian
  function: the component-wise solve with native stages (odeint._adjoint_adaptive_native: no autograd graph, no flat vector) against
  torchdiffeq's flat-vector formulation through the kernel-backed autograd Functions (opt['gnpde_host_adjoint']), same blocks, same
  inputs: same number of evaluations (the controllers see the same numbers to rounding) and gradients to the solver's tolerance."""
  opt = _opt(**ADAPTIVE[name])
  n, d = 1200, opt['hidden_dim']
  ei = random_graph(n, 5, seed=91, hubs=1, hub_deg=600).to(dev)
  x = (torch.randn(n, d, generator=torch.Generator().manual_seed(92)) * 0.5).to(dev)
  z1, gx1, g1, _, nfe1 = _run(dev, opt, ei, x, 93, host=False)
  z2, gx2, g2, _, nfe2 = _run(dev, opt, ei, x, 93, host=True)
  assert torch.equal(z1, z2)
  # same accept / reject sequence -> agreement to float32 rounding; a ratio within rounding of 1 may flip one decision, then the two
  # solves agree to their tolerance like any two adaptive solves
  tol = 1e-4 if nfe1 == nfe2 else 2e-3
  assert abs(nfe1 - nfe2) <= 24, (nfe1, nfe2)
  assert_parity(gx1, gx2, tol, name + ' grad_x')
  checked = 0
  for k, ref in g2.items():
    if float(ref.abs().max()) < 1e-7:
      continue
    assert k in g1, k
    assert_parity(g1[k], ref, tol, name + ' ' + k)
    checked += 1
  assert checked >= 1


def test_adaptive_adjoint_native_stages_are_reproducible(dev):
  """Two runs of the same training step through the native adaptive adjoint give the same bits (no atomics, fixed-order reductions)."""
  opt = _opt(**ADAPTIVE['pubmed_like_heun'])
  n, d = 3000, opt['hidden_dim']
  ei = random_graph(n, 5, seed=94).to(dev)
  x = (torch.randn(n, d, generator=torch.Generator().manual_seed(95)) * 0.5).to(dev)
  z1, gx1, g1, _, nfe1 = _run(dev, opt, ei, x, 96, host=False)
  z2, gx2, g2, _, nfe2 = _run(dev, opt, ei, x, 96, host=False)
  assert torch.equal(z1, z2) and nfe1 == nfe2
  assert torch.equal(gx1, gx2)
  for k in g1:
    assert torch.equal(g1[k], g2[k]), k


@pytest.mark.parametrize('name', sorted(ADAPTIVE))
def test_adaptive_adjoint_device_controller(dev, name):
  """adjoint_method = adaptive_heun (the reference's default) / dopri5: the controller on the device (csrc/adjoint_adaptive.hip, one hipGraph
  replay per trial step) against the same component-wise solve with the controller on the host (opt['gnpde_host_controller_adjoint']) --
  same evaluations, gradients to float32 rounding -- with hub rows and (d = 22) padded rows; and a second iteration replays the captured
  trial steps bit for bit."""
  opt = _opt(**ADAPTIVE[name])
  n, d = 1500, opt['hidden_dim']
  ei = random_graph(n, 5, seed=97, hubs=2, hub_deg=700).to(dev)
  x = (torch.randn(n, d, generator=torch.Generator().manual_seed(98)) * 0.5).to(dev)
  z1, gx1, g1, _, nfe1 = _run(dev, opt, ei, x, 99, host=False)
  z2, gx2, g2, _, nfe2 = _run(dev, dict(opt, gnpde_host_controller_adjoint=True), ei, x, 99, host=False)
  assert torch.equal(z1, z2)
  # The scalar components' error estimates h sum_j e_j Ks_j are differences of float32 dot products over n d terms (sum_j e_j = 0): at
  # tight tolerances their rounding -- a tree sum on the host path, per-wave partial sums folded in double on the device -- is of the
  # size of the tolerance, in torchdiffeq's formulation as much as here, and can flip an accept / reject decision.  Same decisions ->
  # agreement to rounding; otherwise two adaptive solves, agreement to their tolerance.
  tol = 2e-5 if nfe1 == nfe2 else 2e-3
  assert abs(nfe1 - nfe2) <= 24, (nfe1, nfe2)
  assert_parity(gx1, gx2, tol, name + ' grad_x')
  for k, ref in g2.items():
    if float(ref.abs().max()) < 1e-7:
      continue
    assert_parity(g1[k], ref, tol, name + ' ' + k)
  z3, gx3, g3, _, nfe3 = _run(dev, opt, ei, x, 99, host=False)
  assert torch.equal(gx1, gx3) and nfe1 == nfe3
