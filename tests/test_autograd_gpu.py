"""Gradients of the right-hand side and of a whole differentiated solve against CPU autograd through the
oracle (same op sequence as the reference)."""
import pytest
import torch

import gnpde_amd as G
from oracle import restate as R
from helpers import Data, assert_parity, random_graph

pytestmark = pytest.mark.gpu

OPT = dict(heads=4, attention_dim=16, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False,
           reweight_attention=False, beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9,
           add_source=True, no_alpha_sigmoid=False, mix_features=False, hidden_dim=20, augment=False, adjoint=False,
           tol_scale=1.0, data_norm='rw', method='rk4', step_size=1.0, max_iters=100, block='constant',
           function='transformer', time=2.0)
GTOL = 2e-4  # gradients accumulate more rounding than values (long reductions in different orders)


def _rand_params(mod, seed, dev):
  g = torch.Generator().manual_seed(seed)
  with torch.no_grad():
    for p in mod.parameters():
      if p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      else:
        p.copy_((torch.randn(p.shape, generator=g) * 0.3).to(dev))


def _cpu(t):
  return t.detach().cpu().clone().requires_grad_(True)


@pytest.mark.parametrize('block_kind', ['constant', 'attention'])
def test_laplacian_native_backward(dev, block_kind):
  n, d = 700, 20
  ei = random_graph(n, 5, seed=3, hubs=1, hub_deg=600)
  g = torch.Generator().manual_seed(1)
  x, x0, go = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
  opt = dict(OPT, function='laplacian', block=block_kind)
  func = G.LaplacianODEFunc(d, d, opt, Data(x, ei), dev).to(dev)
  _rand_params(func, 2, dev)
  w_edge = torch.rand(ei.shape[1], 4 if block_kind == 'attention' else 1, generator=g) + 0.1
  if block_kind != 'attention':
    w_edge = w_edge[:, 0].contiguous()
  func.edge_index = ei.to(dev)
  wd = w_edge.to(dev).requires_grad_(True)
  func.edge_weight, func.attention_weights, func.x0 = wd, wd, x0.to(dev)
  xd = x.to(dev).requires_grad_(True)
  f = func(0.0, xd)
  f.backward(go.to(dev))
  # oracle on CPU
  xc, wc, ac, bc = _cpu(x), _cpu(w_edge), _cpu(func.alpha_train), _cpu(func.beta_train)
  fr = R.rhs_laplacian(xc, ei, wc, ac, bc, x0, False, True)
  assert_parity(f, fr, what='value')
  fr.backward(go)
  assert_parity(xd.grad, xc.grad, tol=GTOL, what='dx')
  assert_parity(wd.grad, wc.grad, tol=GTOL, what='d edge weights')
  assert_parity(func.alpha_train.grad.reshape(-1), ac.grad.reshape(-1), tol=GTOL, what='dalpha')
  assert_parity(func.beta_train.grad.reshape(-1), bc.grad.reshape(-1), tol=GTOL, what='dbeta')


@pytest.mark.parametrize('function,composite', [('transformer', False), ('transformer', True), ('GAT', True)])
def test_training_step_through_the_solver(dev, function, composite):
  """loss.backward() through a 2-step rk4 solve of a block in training mode (what run_GNN.py's train() does);
  GRAND-nl both through the native VJP and through the composite backward."""
  n, d = 400, 20
  ei = random_graph(n, 5, seed=5, hubs=1, hub_deg=150)
  g = torch.Generator().manual_seed(4)
  x = torch.randn(n, d, generator=g)
  opt = dict(OPT, function=function, gnpde_composite_backward=composite)
  fcls = G.ODEFuncTransformerAtt if function == 'transformer' else G.ODEFuncAtt
  block = G.ConstantODEblock(fcls, [], opt, Data(x.to(dev), ei.to(dev)), dev, t=torch.tensor([0, opt['time']])).to(dev)
  _rand_params(block, 6, dev)
  block.train()
  xd = x.to(dev).requires_grad_(True)
  block.set_x0(xd)
  z = block(xd)
  loss = (z ** 2).sum()
  loss.backward()
  f = block.odefunc
  lay = f.multihead_att_layer
  edge = f.edge_index.cpu()
  xc = _cpu(x)
  ac, bc = _cpu(f.alpha_train), _cpu(f.beta_train)
  if function == 'transformer':
    ps = [_cpu(p) for p in (lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias)]
    rhs = lambda t, y: R.rhs_transformer(y, edge, ps[0], ps[1], ps[2], ps[3], 4, ac, bc, x, False, True)
    ours = [lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias]
  else:
    ps = [_cpu(lay.W), _cpu(lay.a)]
    rhs = lambda t, y: R.rhs_gat(y, edge, ps[0], ps[1], 4, ac, bc, x, False, True, 0.2, 0)
    ours = [lay.W, lay.a]
  zr = R.odeint_fixed(rhs, xc, opt['time'], 1.0, 'rk4')
  assert_parity(z, zr, what='z')
  (zr ** 2).sum().backward()
  assert_parity(xd.grad, xc.grad, tol=GTOL, what='dx')
  scale = max(float(b.grad.abs().max()) for b in ps)
  for a, b in zip(ours, ps):
    # (the gradient w.r.t. K.bias is identically zero -- a row softmax ignores q_i.b_k -- so compare absolutely)
    assert float((a.grad.cpu() - b.grad).abs().max()) <= GTOL * scale, 'dparam'
  assert_parity(f.alpha_train.grad.reshape(-1), ac.grad.reshape(-1), tol=GTOL, what='dalpha')


@pytest.mark.parametrize('square_plus', [False, True])
@pytest.mark.parametrize('norm_idx', [0, 1])
def test_native_vjp_every_normaliser_against_float64(dev, square_plus, norm_idx):
  """Native VJP (no composite) of GRAND-nl for softmax / squareplus over rows / columns -- squareplus + attention_norm_idx = 1
  is what run_GNN.py trains Cora / Citeseer / Pubmed / CoauthorCS with.  The reference gradient is the oracle evaluated in
  FLOAT64; the fp32 CPU autograd gradient of the same op sequence is measured against it too, which shows what GTOL has to
  cover: the GPU result must be as close to the float64 truth as fp32 arithmetic allows (within 3x of the CPU fp32 error
  or 2e-5 relative), not merely close to another fp32 number."""
  from gnpde_amd import autograd as AG
  n, d = 500, 24
  ei = random_graph(n, 6, seed=15, hubs=1, hub_deg=700)
  g = torch.Generator().manual_seed(16)
  x = torch.randn(n, d, generator=g)
  opt = dict(OPT, hidden_dim=d, square_plus=square_plus, attention_norm_idx=norm_idx, time=2.0)
  block = G.ConstantODEblock(G.ODEFuncTransformerAtt, [], opt, Data(x.to(dev), ei.to(dev)), dev, t=torch.tensor([0, 2.0])).to(dev)
  _rand_params(block, 17, dev)
  block.train()
  f = block.odefunc
  assert AG._native_transformer_vjp_ok(f), 'this configuration must not fall back to the composite backward'
  AG._warned.discard('ODEFuncTransformerAtt')
  xd = x.to(dev).requires_grad_(True)
  block.set_x0(xd)
  z = block(xd)
  (z ** 2).sum().backward()
  assert 'ODEFuncTransformerAtt' not in AG._warned, 'the composite backward announced itself'
  lay = f.multihead_att_layer
  edge = f.edge_index.cpu()
  ours = [lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias, f.alpha_train, f.beta_train]

  def reference(dtype):
    cast = lambda t: t.detach().cpu().to(dtype).clone().requires_grad_(True)   # noqa: E731
    xc = cast(x)
    ps = [cast(p) for p in ours]
    x0 = x.to(dtype)
    rhs = lambda t, y: R.rhs_transformer(y, edge, ps[0], ps[1], ps[2], ps[3], 4, ps[4], ps[5], x0, False, True,   # noqa: E731
                                         norm_idx=norm_idx, square_plus=square_plus)
    zr = R.odeint_fixed(rhs, xc, 2.0, 1.0, 'rk4')
    (zr ** 2).sum().backward()
    return zr.detach(), [xc.grad] + [p.grad for p in ps]

  z64, g64 = reference(torch.float64)
  z32, g32 = reference(torch.float32)
  assert_parity(z, z64.float(), what='z')
  got = [xd.grad] + [p.grad for p in ours]
  names = ['dx', 'dWq', 'dbq', 'dWk', 'dbk', 'dalpha', 'dbeta']
  scale = max(float(t.abs().max()) for t in g64[1:5])
  for name, a, b32, b64 in zip(names, got, g32, g64):
    ref_scale = float(b64.abs().max()) if name in ('dx', 'dalpha', 'dbeta') else scale
    e_gpu = float((a.detach().cpu().double().reshape(b64.shape) - b64).abs().max()) / ref_scale
    e_cpu = float((b32.double() - b64).abs().max()) / ref_scale
    assert e_gpu <= max(3 * e_cpu, 2e-5), '%s: GPU error %.2e vs float64, CPU fp32 error %.2e' % (name, e_gpu, e_cpu)
    assert e_gpu <= GTOL


def _grad_close(name, got, ref64, ref32, scale=None):
  scale = float(ref64.abs().max()) if scale is None else scale
  e_gpu = float((got.detach().cpu().double().reshape(ref64.shape) - ref64).abs().max()) / scale
  e_cpu = float((ref32.double() - ref64).abs().max()) / scale
  assert e_gpu <= max(3 * e_cpu, 3e-5), '%s: GPU error %.2e vs float64, CPU fp32 error %.2e' % (name, e_gpu, e_cpu)


@pytest.mark.parametrize('att_type,norm_idx,square_plus', [('cosine_sim', 0, False), ('pearson', 1, False), ('exp_kernel', 0, False),
                                                          ('exp_kernel', 1, True), ('cosine_sim', 1, True)])
def test_native_vjp_other_score_functions(dev, att_type, norm_idx, square_plus):
  """cosine / pearson / exp_kernel attention: node-level transforms in PyTorch, every per-edge step native (no composite);
  gradients of one evaluation of f against the float64 oracle."""
  from gnpde_amd import autograd as AG
  n, d = 400, 24
  ei = random_graph(n, 6, seed=25, hubs=1, hub_deg=600)
  g = torch.Generator().manual_seed(26)
  x, go = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
  opt = dict(OPT, hidden_dim=d, attention_type=att_type, attention_norm_idx=norm_idx, square_plus=square_plus)
  func = G.ODEFuncTransformerAtt(d, d, opt, Data(x.to(dev), ei.to(dev)), dev).to(dev)
  _rand_params(func, 27, dev)
  lay = func.multihead_att_layer
  if att_type == 'exp_kernel':
    with torch.no_grad():
      lay.lengthscale.fill_(1.3)
      lay.output_var.fill_(0.8)
  AG._warned.clear()
  xd = x.to(dev).requires_grad_(True)
  func.x0 = x.to(dev)
  f = func(0.0, xd)
  f.backward(go.to(dev))
  assert not AG._warned, 'a composite backward announced itself: %s' % AG._warned
  edge = func.edge_index.cpu()
  names = ['Q.weight', 'Q.bias', 'K.weight', 'K.bias'] + (['output_var', 'lengthscale'] if att_type == 'exp_kernel' else [])
  ours = [dict(lay.named_parameters())[k] for k in names] + [func.alpha_train, func.beta_train]

  def reference(dtype):
    cast = lambda t: t.detach().cpu().to(dtype).clone().requires_grad_(True)   # noqa: E731
    xc = cast(x)
    ps = [cast(p) for p in ours]
    kw = dict(attention_type=att_type, norm_idx=norm_idx, square_plus=square_plus)
    if att_type == 'exp_kernel':
      kw.update(output_var=ps[4], lengthscale=ps[5])
    fr = R.rhs_transformer(xc, edge, ps[0], ps[1], ps[2], ps[3], 4, ps[-2], ps[-1], x.to(dtype), False, True, **kw)
    fr.backward(go.to(dtype))
    return fr.detach(), [xc.grad] + [p.grad for p in ps]

  f64, g64 = reference(torch.float64)
  f32, g32 = reference(torch.float32)
  assert_parity(f, f64.float(), what='f')
  scale = max(float(t.abs().max()) for t in g64[1:5])
  for name, a, b32, b64 in zip(['dx'] + names + ['dalpha', 'dbeta'], [xd.grad] + [p.grad for p in ours], g32, g64):
    own = name in ('dx', 'dalpha', 'dbeta', 'output_var', 'lengthscale')
    _grad_close(name, a, b64, b32, None if own else scale)


@pytest.mark.parametrize('mix', [False, True])
def test_native_vjp_gat(dev, mix):
  """GAT attention (and the mix_features aggregation A(x) (x W) Wout the reference trains with when asked to): per-edge
  backward native, gradients against a float64 evaluation of the reference op sequence."""
  from gnpde_amd import autograd as AG
  n, d = 300, 16
  ei = random_graph(n, 6, seed=35)
  g = torch.Generator().manual_seed(36)
  x, go = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
  opt = dict(OPT, hidden_dim=d, function='GAT', mix_features=mix, attention_dim=16, heads=4)
  func = G.ODEFuncAtt(d, d, opt, Data(x.to(dev), ei.to(dev)), dev).to(dev)
  _rand_params(func, 37, dev)
  lay = func.multihead_att_layer
  AG._warned.clear()
  xd = x.to(dev).requires_grad_(True)
  func.x0 = x.to(dev)
  f = func(0.0, xd)
  f.backward(go.to(dev))
  assert not AG._warned
  edge = func.edge_index.cpu()
  ours = [lay.W, lay.a] + ([lay.Wout] if mix else []) + [func.alpha_train, func.beta_train]

  def reference(dtype):
    cast = lambda t: t.detach().cpu().to(dtype).clone().requires_grad_(True)   # noqa: E731
    xc = cast(x)
    ps = [cast(p) for p in ours]
    if not mix:
      fr = R.rhs_gat(xc, edge, ps[0], ps[1], 4, ps[-2], ps[-1], x.to(dtype), False, True, 0.2, 0)
    else:      # reference src/function_GAT_attention.py:33-38,56-64
      att, wx = R.gat_attention(xc, edge, ps[0], ps[1], 4, 0.2, 0)
      ax = torch.mm(R.spmm(edge, att.mean(dim=1), n, wx), ps[2])
      fr = torch.sigmoid(ps[-2]) * (ax - xc) + ps[-1] * x.to(dtype)
    fr.backward(go.to(dtype))
    return fr.detach(), [xc.grad] + [p.grad for p in ps]

  f64, g64 = reference(torch.float64)
  f32, g32 = reference(torch.float32)
  assert_parity(f, f64.float(), what='f')
  for name, a, b32, b64 in zip(['dx', 'dW', 'da'] + (['dWout'] if mix else []) + ['dalpha', 'dbeta'],
                               [xd.grad] + [p.grad for p in ours], g32, g64):
    _grad_close(name, a, b64, b32)


def test_native_vjp_blend_attention_block(dev):
  """AttODEblock + Laplacian function with the BLEND split kernel (exp kernels on feature and positional channels): the
  attention computed ONCE per forward carries gradients back into Qx, Kx, Qp, Kp, both length scales and output variances --
  per-edge work native (no composite), against float64 autograd through the oracle."""
  from gnpde_amd import autograd as AG
  n, f0, p0 = 300, 12, 8
  d = f0 + p0
  ei = random_graph(n, 6, seed=45)
  g = torch.Generator().manual_seed(46)
  x, c = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
  opt = dict(OPT, hidden_dim=d, function='laplacian', block='attention', method='euler', time=2.0, attention_type='exp_kernel',
             beltrami=True, feat_hidden_dim=f0, pos_enc_hidden_dim=p0, attention_dim=16, heads=2)
  block = G.AttODEblock(G.LaplacianODEFunc, [], opt, Data(x.to(dev), ei.to(dev)), dev, t=torch.tensor([0, 2.0])).to(dev)
  _rand_params(block, 47, dev)
  lay = block.multihead_att_layer
  with torch.no_grad():
    for nm, v in (('lengthscale_x', 1.2), ('lengthscale_p', 0.9), ('output_var_x', 1.1), ('output_var_p', 0.7)):
      getattr(lay, nm).fill_(v)
  block.train()
  AG._warned.clear()
  xd = x.to(dev).requires_grad_(True)
  block.set_x0(xd)
  z = block(xd)
  (z * c.to(dev)).sum().backward()
  assert not AG._warned, AG._warned
  f = block.odefunc
  e_n = f.edge_index.cpu()
  names = [k for k, _ in lay.named_parameters() if k.split('.')[0] in ('Qx', 'Kx', 'Qp', 'Kp') or 'lengthscale' in k or 'output_var' in k]
  ours = [dict(lay.named_parameters())[k] for k in names]

  def reference(dtype):
    cast = lambda t: t.detach().cpu().to(dtype).clone().requires_grad_(True)   # noqa: E731
    xc = cast(x)
    P = {k: cast(p) for k, p in zip(names, ours)}
    al, be = cast(f.alpha_train), cast(f.beta_train)
    att, _ = R.transformer_attention_split(xc, e_n, P, 2, f0, p0, edge_weights=f.edge_weight.cpu().to(dtype), reweight=False)
    x0 = x.to(dtype)                                    # ODEblock.set_x0 detaches the source term
    rhs = lambda t, y: R.rhs_from_attention(y, e_n, att, al, be, x0, False, True)   # noqa: E731
    zr = R.odeint_fixed(rhs, xc, 2.0, 1.0, 'euler')
    (zr * c.to(dtype)).sum().backward()
    return zr.detach(), [xc.grad] + [P[k].grad for k in names]

  z64, g64 = reference(torch.float64)
  z32, g32 = reference(torch.float32)
  assert_parity(z, z64.float(), what='z')
  wscale = max(float(t.abs().max()) for t in g64[1:] if t.dim() >= 2)
  for name, a, b32, b64 in zip(['dx'] + names, [xd.grad] + [p.grad for p in ours], g32, g64):
    assert a is not None, name
    _grad_close(name, a, b64, b32, None if (name == 'dx' or b64.dim() < 2 and b64.numel() == 1) else wscale)


def test_attention_block_training(dev):
  """AttODEblock in training mode: attention computed once (with history), Laplacian function native backward,
  edge-weight gradients through gnpde_sddmm into the attention layer's parameters."""
  n, d = 300, 20
  ei = random_graph(n, 5, seed=8)
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(7))
  opt = dict(OPT, function='laplacian', block='attention', method='euler', time=3.0)
  block = G.AttODEblock(G.LaplacianODEFunc, [], opt, Data(x.to(dev), ei.to(dev)), dev, t=torch.tensor([0, 3.0])).to(dev)
  _rand_params(block, 9, dev)
  block.train()
  xd = x.to(dev).requires_grad_(True)
  block.set_x0(xd)
  z = block(xd)
  (z ** 2).sum().backward()
  lay, f = block.multihead_att_layer, block.odefunc
  xc = _cpu(x)
  ps = [_cpu(p) for p in (lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias)]
  ac, bc = _cpu(f.alpha_train), _cpu(f.beta_train)
  e_n, w_n = R.get_rw_adj(ei, None, 1, 1, n)
  att, _ = R.transformer_attention(xc, e_n, ps[0], ps[1], ps[2], ps[3], 4)
  rhs = lambda t, y: R.rhs_laplacian(y, e_n, att, ac, bc, x, False, True)
  zr = R.odeint_fixed(rhs, xc, 3.0, 1.0, 'euler')
  assert_parity(z, zr, what='z')
  (zr ** 2).sum().backward()
  assert_parity(xd.grad, xc.grad, tol=GTOL, what='dx')
  scale = max(float(b.grad.abs().max()) for b in ps)
  for a, b in zip((lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias), ps):
    assert float((a.grad.cpu() - b.grad).abs().max()) <= GTOL * scale, 'd attention params'



def test_sddmm(dev):
  from gnpde_amd import ops
  n, d = 900, 36
  ei = random_graph(n, 6, seed=2, hubs=1, hub_deg=700)
  g = torch.Generator().manual_seed(3)
  a, b = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
  graph = G.CSRGraph(ei.to(dev), n)
  out = ops.sddmm(graph, a.to(dev), b.to(dev))
  ref = (a[ei[0]] * b[ei[1]]).sum(dim=1)
  assert_parity(out[:graph.e], ref[graph.perm_long.cpu()], what='sddmm')


@pytest.mark.parametrize('heads,att_dim,reweight', [(4, 16, False), (8, 128, False), (1, 8, True), (2, 32, True)])
def test_transformer_native_vjp_one_evaluation(dev, heads, att_dim, reweight):
  """Native VJP of one GRAND-nl evaluation (hub rows, biases, optional reweighting) against CPU autograd
  through the oracle."""
  n, d = 900, 24
  ei = random_graph(n, 6, seed=heads, hubs=1, hub_deg=700, dup=10)
  g = torch.Generator().manual_seed(att_dim)
  x, x0, go = (torch.randn(n, d, generator=g) for _ in range(3))
  opt = dict(OPT, heads=heads, attention_dim=att_dim, hidden_dim=d, reweight_attention=reweight, self_loop_weight=0)
  ew = torch.rand(ei.shape[1], generator=g) + 0.5
  data = Data(x.to(dev), ei.to(dev), edge_attr=ew.to(dev) if reweight else None)
  func = G.ODEFuncTransformerAtt(d, d, opt, data, dev).to(dev)
  _rand_params(func, 3, dev)
  func.x0 = x0.to(dev)
  xd = x.to(dev).requires_grad_(True)
  f = func(0.0, xd)
  f.backward(go.to(dev))
  lay = func.multihead_att_layer
  ps = [_cpu(p) for p in (lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias)]
  ac, bc, xc = _cpu(func.alpha_train), _cpu(func.beta_train), _cpu(x)
  fr = R.rhs_transformer(xc, ei, ps[0], ps[1], ps[2], ps[3], heads, ac, bc, x0, False, True,
                         edge_weights=ew if reweight else None, reweight=reweight)
  assert_parity(f, fr, what='value')
  fr.backward(go)
  assert_parity(xd.grad, xc.grad, tol=GTOL, what='dx')
  scale = max(float(b.grad.abs().max()) for b in ps)
  for a, b in zip((lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias), ps):
    assert float((a.grad.cpu() - b.grad).abs().max()) <= GTOL * scale, 'dparam'
  assert_parity(func.alpha_train.grad.reshape(-1), ac.grad.reshape(-1), tol=GTOL, what='dalpha')
  assert_parity(func.beta_train.grad.reshape(-1), bc.grad.reshape(-1), tol=GTOL, what='dbeta')


def test_cora_best_params_training_step(dev):
  """The reference's Cora configuration in TRAINING mode: attention block (squareplus, attention_norm_idx 1,
  8 heads) computed once with autograd history, Laplacian function with native backward, dopri5 through the
  differentiable host controller -- loss gradients w.r.t. the input and the attention parameters against CPU
  autograd through the oracle driven by the same controller."""
  n, d = 500, 24
  ei = random_graph(n, 5, seed=13)
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(14)) * 0.5
  opt = dict(OPT, function='laplacian', block='attention', method='dopri5', time=4.0, tol_scale=800.0, heads=8,
             attention_dim=32, hidden_dim=d, square_plus=True, attention_norm_idx=1)
  block = G.AttODEblock(G.LaplacianODEFunc, [], opt, Data(x.to(dev), ei.to(dev)), dev, t=torch.tensor([0, 4.0])).to(dev)
  _rand_params(block, 15, dev)
  block.train()
  xd = x.to(dev).requires_grad_(True)
  block.set_x0(xd)
  z = block(xd)
  (z ** 2).sum().backward()
  lay, f = block.multihead_att_layer, block.odefunc
  xc = _cpu(x)
  ps = [_cpu(p) for p in (lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias)]
  ac, bc = _cpu(f.alpha_train), _cpu(f.beta_train)
  e_n, w_n = R.get_rw_adj(ei, None, 1, 1, n)
  att, _ = R.transformer_attention(xc, e_n, ps[0], ps[1], ps[2], ps[3], 8, norm_idx=1, square_plus=True)
  rhs = lambda t, y: R.rhs_laplacian(y, e_n, att, ac, bc, x, False, True)
  # (reference side: the differentiable host loop of this package on the CPU oracle -- the restated torchdiffeq of oracle/shims
  #  assigns its stages in place, which autograd refuses; the forward-only full-size tests use it, tests/test_solver_gpu.py)
  zr = G.odeint(rhs, xc, torch.tensor([0, 4.0]), method='dopri5', options={}, atol=800.0 * 1e-7, rtol=800.0 * 1e-9)[1]
  assert_parity(z, zr, tol=1e-4, what='z')
  (zr ** 2).sum().backward()
  assert_parity(xd.grad, xc.grad, tol=2e-3, what='dx')
  scale = max(float(b.grad.abs().max()) for b in ps)
  for a, b in zip((lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias), ps):
    assert float((a.grad.cpu() - b.grad).abs().max()) <= 2e-3 * scale, 'd attention params'


def test_split_kernel_gradients(dev):
  """BLEND split feature / positional exp kernel: native forward (one combined projection + exp_kernel), composite
  backward, against CPU autograd through the oracle's restatement of the reference branch."""
  from helpers import Fixture
  fx = Fixture('func_transformer_beltrami_expkernel_labels_n1')
  opt = fx.opt
  x = fx.t('x')
  func = G.ODEFuncTransformerAtt(x.shape[1], x.shape[1], opt, Data(x.to(dev), fx.t('edge_index', dev)), dev).to(dev)
  func.load_state_dict(fx.params, strict=True)
  func.x0 = fx.t('x0', dev)
  g = torch.Generator().manual_seed(4)
  go = torch.randn(x.shape, generator=g)
  xd = x.to(dev).requires_grad_(True)
  f = func(0.0, xd)
  f.backward(go.to(dev))
  # oracle
  P = {k: v.clone().requires_grad_(True) for k, v in fx.params.items()}
  lay = {k[len('multihead_att_layer.'):]: v for k, v in P.items() if k.startswith('multihead_att_layer.')}
  xc = x.clone().requires_grad_(True)
  edge = fx.t('func_edge_index')
  att, _ = R.transformer_attention_split(xc, edge, lay, opt['heads'], opt['feat_hidden_dim'], opt['pos_enc_hidden_dim'],
                                         opt['attention_norm_idx'], opt['square_plus'])
  fr = R.rhs_from_attention(xc, edge, att, P['alpha_train'], P['beta_train'], fx.t('x0'), opt['no_alpha_sigmoid'], opt['add_source'])
  assert_parity(f, fr, what='value')
  fr.backward(go)
  assert_parity(xd.grad, xc.grad, tol=GTOL, what='dx')
  named = dict(func.named_parameters())
  for k in ('multihead_att_layer.Qx.weight', 'multihead_att_layer.Kp.weight', 'multihead_att_layer.Kx.bias',
            'multihead_att_layer.lengthscale_x', 'multihead_att_layer.output_var_p', 'alpha_train'):
    assert_parity(named[k].grad.reshape(-1), P[k].grad.reshape(-1), tol=GTOL, what=k)


@pytest.mark.parametrize('d', [128, 36, 162, 27])
def test_sddmm_hub_chunks_and_widths(dev, d):
  """Chunked hub rows (two hubs of 1300 entries = 3 chunks each), vector widths 4 / 2 / 1, scaled by sigmoid(alpha)."""
  from gnpde_amd import ops
  n = 2500
  ei = random_graph(n, 5, seed=d, hubs=2, hub_deg=1300)
  g = torch.Generator().manual_seed(d)
  a, b = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
  graph = G.CSRGraph(ei.to(dev), n)
  assert graph.n_long_chunks >= 6
  alpha = torch.tensor([0.3], device=dev)
  out = ops.sddmm(graph, a.to(dev), b.to(dev), scale=alpha, scale_sigmoid=True)
  ref = torch.sigmoid(torch.tensor(0.3)) * (a[ei[0]] * b[ei[1]]).sum(dim=1)
  assert_parity(out[:graph.e], ref[graph.perm_long.cpu()], what='sddmm d=%d' % d)


def test_tall_skinny_gram(dev):
  from gnpde_amd import ops
  g = torch.Generator().manual_seed(0)
  for n in (40017, 1000):
    a, b = torch.randn(n, 32, generator=g).to(dev), torch.randn(n, 128, generator=g).to(dev)
    ref = (a.double().t() @ b.double()).float()
    assert_parity(ops.tall_skinny_gram(a, b), ref, tol=2e-5, what='gram n=%d' % n)


def test_transformer_native_vjp_raw_alpha(dev):
  """no_alpha_sigmoid: d alpha from sum_e w_e (g_row . x_col) instead of the saved forward value."""
  n, d = 900, 24
  ei = random_graph(n, 6, seed=5, hubs=1, hub_deg=700)
  g = torch.Generator().manual_seed(6)
  x, x0, go = (torch.randn(n, d, generator=g) for _ in range(3))
  opt = dict(OPT, hidden_dim=d, no_alpha_sigmoid=True, self_loop_weight=0)
  func = G.ODEFuncTransformerAtt(d, d, opt, Data(x.to(dev), ei.to(dev)), dev).to(dev)
  _rand_params(func, 7, dev)
  func.x0 = x0.to(dev)
  xd = x.to(dev).requires_grad_(True)
  f = func(0.0, xd)
  f.backward(go.to(dev))
  lay = func.multihead_att_layer
  ps = [_cpu(p) for p in (lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias)]
  ac, bc, xc = _cpu(func.alpha_train), _cpu(func.beta_train), _cpu(x)
  fr = R.rhs_transformer(xc, ei, ps[0], ps[1], ps[2], ps[3], 4, ac, bc, x0, True, True)
  assert_parity(f, fr, what='value')
  fr.backward(go)
  assert_parity(xd.grad, xc.grad, tol=GTOL, what='dx')
  assert_parity(func.alpha_train.grad.reshape(-1), ac.grad.reshape(-1), tol=GTOL, what='dalpha')
  assert_parity(lay.Q.weight.grad, ps[0].grad, tol=GTOL, what='dWq')


@pytest.mark.parametrize('heads,dk,reweight', [(4, 4, False), (8, 16, False), (1, 8, True), (2, 4, True), (4, 16, False)])
def test_attention_rows_bwd_matches_two_step_path(dev, heads, dk, reweight):
  """One-pass scores + softmax + backward against per-head attention followed by gnpde_softmax_rows_bwd, on a graph
  with hub rows (block-wide three-pass branch), empty rows and duplicate edges."""
  from gnpde_amd import ops, _lib
  n, A = 1500, heads * dk
  ei = random_graph(n, 7, seed=heads + dk, hubs=2, hub_deg=900, isolated=5, dup=20)
  g = torch.Generator().manual_seed(dk)
  qk = (torch.randn(n, 2 * A, generator=g) * 0.7).to(dev)
  r = torch.randn(ei.shape[1], generator=g).to(dev)
  graph = G.CSRGraph(ei.to(dev), n)
  assert graph.n_long_rows >= 2
  ew = (torch.rand(ei.shape[1], generator=g) + 0.5).to(dev) if reweight else None
  ew_csr = ops.edge_to_csr_mean(graph, ew) if reweight else None
  st = ops.attention_struct(_lib.ATT_SCALED_DOT, heads, A, 0, False, q=qk, k=qk[:, A:], ldqk=2 * A, edge_w_csr=ew_csr)
  alpha = torch.tensor([0.4], device=dev)
  r_csr = ops.edge_to_csr_mean(graph, r)
  fused = ops.attention_rows_bwd(graph, st, r_csr, heads, scale=alpha, scale_sigmoid=True)
  assert fused is not None
  _, att_edge, _ = ops.edge_attention(graph, st, False, True, False, like=qk)
  ref = ops.softmax_rows_bwd(graph, att_edge, r_csr, edge_w_csr=ew_csr, scale=alpha, scale_sigmoid=True)
  assert_parity(fused[:graph.e], ref[:graph.e], tol=2e-5, what='ds')


def test_weight_snapshot_survives_id_reuse(dev):
  """Round-1 review: the backward snapshot of the Laplacian weights was keyed on id() of the attention tensor; a train ->
  eval -> train sequence frees the first tensor and hands its id to the third, which made a stale snapshot compare equal.
  The snapshot is keyed on a generation counter now: gradients of the third forward must use the third weights."""
  n, d = 300, 16
  ei = random_graph(n, 5, seed=71)
  g = torch.Generator().manual_seed(72)
  x, go = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
  opt = dict(OPT, hidden_dim=d, function='laplacian', block='attention')
  func = G.LaplacianODEFunc(d, d, opt, Data(x.to(dev), ei.to(dev)), dev).to(dev)
  func.edge_index = ei.to(dev)
  func.x0 = x.to(dev)
  E = ei.shape[1]

  def grad_for(w):
    func.attention_weights = w
    xd = x.to(dev).requires_grad_(True)
    func(0.0, xd).backward(go.to(dev))
    return xd.grad.clone()

  w1 = (torch.rand(E, 4, generator=g) + 0.1).to(dev).requires_grad_(True)
  g1 = grad_for(w1)
  del w1
  with torch.no_grad():                                   # eval forward with other weights: replaces the live entry
    func.attention_weights = (torch.rand(E, 4, generator=g) + 0.1).to(dev)
    func(0.0, x.to(dev))
  w3_cpu = torch.rand(E, 4, generator=g) + 0.1
  got = []
  for _ in range(8):                                      # new tensors until one reuses a freed id (usually the first)
    w3 = w3_cpu.to(dev).requires_grad_(True)
    got.append(grad_for(w3))
    del w3
  ref = R.rhs_laplacian(x.clone().requires_grad_(True), ei, w3_cpu.mean(dim=1), func.alpha_train.detach().cpu(),
                        func.beta_train.detach().cpu(), x, False, True)
  xr = x.clone().requires_grad_(True)
  R.rhs_laplacian(xr, ei, w3_cpu.mean(dim=1), func.alpha_train.detach().cpu(), func.beta_train.detach().cpu(), x, False,
                  True).backward(go)
  for gg in got:
    assert_parity(gg, xr.grad, tol=GTOL, what='dx with the CURRENT weights')
  assert float((g1.cpu() - xr.grad).abs().max()) > 1e-3, 'the first weights must give a different gradient for the test to bite'
