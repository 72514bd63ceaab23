"""Pins the CPU oracle (oracle/restate.py) against the golden vectors recorded from the reference's
own code, and against the known-answer facts of the reference's unit tests."""
import numpy as np
import pytest
import torch

from oracle import restate as R
from helpers import Fixture, fixtures, assert_parity

TIGHT = 2e-6  # same op order as the reference on the same CPU: agreement to a few ulp


def _att_kwargs(opt):
  return dict(attention_type=opt['attention_type'], norm_idx=opt['attention_norm_idx'], square_plus=opt['square_plus'])


def _qk(p, prefix=''):
  return (p[prefix + 'multihead_att_layer.Q.weight'], p[prefix + 'multihead_att_layer.Q.bias'],
          p[prefix + 'multihead_att_layer.K.weight'], p[prefix + 'multihead_att_layer.K.bias'])


def _is_split(opt):
  return bool(opt['beltrami'] and opt['attention_type'] == 'exp_kernel')


def _split_attention(opt, p, prefix, x, edge, edge_weights=None):
  lay = prefix + 'multihead_att_layer.'
  P = {k[len(lay):]: v for k, v in p.items() if k.startswith(lay)}
  return R.transformer_attention_split(x, edge, P, opt['heads'], opt['feat_hidden_dim'], opt['pos_enc_hidden_dim'],
                                       opt['attention_norm_idx'], opt['square_plus'], edge_weights,
                                       opt['reweight_attention'])


def _transformer_rhs(fx, p, prefix, edge, x0):
  opt = fx.opt
  if _is_split(opt):
    return lambda t, x: R.rhs_from_attention(x, edge, _split_attention(opt, p, prefix, x, edge)[0], p[prefix + 'alpha_train'],
                                             p[prefix + 'beta_train'], x0, opt['no_alpha_sigmoid'], opt['add_source'])
  kw = _att_kwargs(opt)
  if opt['attention_type'] == 'exp_kernel':
    kw.update(output_var=p[prefix + 'multihead_att_layer.output_var'], lengthscale=p[prefix + 'multihead_att_layer.lengthscale'])
  Wq, bq, Wk, bk = _qk(p, prefix)
  return lambda t, x: R.rhs_transformer(x, edge, Wq, bq, Wk, bk, opt['heads'], p[prefix + 'alpha_train'],
                                        p[prefix + 'beta_train'], x0, opt['no_alpha_sigmoid'], opt['add_source'], **kw)


@pytest.mark.parametrize('name', fixtures('func_transformer_'))
def test_transformer_function(name):
  fx = Fixture(name)
  x, x0, p, opt = fx.t('x'), fx.t('x0'), fx.params, fx.opt
  n = x.shape[0]
  edge = fx.t('edge_index')
  if opt['self_loop_weight'] > 0:
    edge, _ = R.add_remaining_self_loops(edge, None, opt['self_loop_weight'], int(edge.max()) + 1)
  assert torch.equal(edge, fx.t('func_edge_index'))
  if _is_split(opt):
    att, prods = _split_attention(opt, p, '', x, edge)
  else:
    kw = _att_kwargs(opt)
    if opt['attention_type'] == 'exp_kernel':
      kw.update(output_var=p['multihead_att_layer.output_var'], lengthscale=p['multihead_att_layer.lengthscale'])
    att, prods = R.transformer_attention(x, edge, *_qk(p), opt['heads'], **kw)
  assert_parity(prods, fx.t('prods'), TIGHT, 'prods')
  assert_parity(att, fx.t('attention'), TIGHT, 'attention')
  f = _transformer_rhs(fx, p, '', edge, x0)(0.0, x)
  assert_parity(f, fx.t('f'), TIGHT, 'f')


def test_layer_reweight():
  fx = Fixture('layer_reweight')
  p = fx.params
  att, prods = R.transformer_attention(fx.t('x'), fx.t('func_edge_index'), p['Q.weight'], p['Q.bias'], p['K.weight'],
                                       p['K.bias'], fx.opt['heads'], edge_weights=fx.t('edge_weight'), reweight=True,
                                       **_att_kwargs(fx.opt))
  assert_parity(prods, fx.t('prods'), TIGHT, 'prods')
  assert_parity(att, fx.t('attention'), TIGHT, 'attention')


@pytest.mark.parametrize('name', fixtures('func_gat_'))
def test_gat_function(name):
  fx = Fixture(name)
  p, opt = fx.params, fx.opt
  edge = fx.t('func_edge_index')
  W, a = p['multihead_att_layer.W'], p['multihead_att_layer.a']
  att, wx = R.gat_attention(fx.t('x'), edge, W, a, opt['heads'], opt['leaky_relu_slope'], opt['attention_norm_idx'])
  assert_parity(att, fx.t('attention'), TIGHT, 'attention')
  assert_parity(wx, fx.t('wx'), TIGHT, 'wx')
  f = R.rhs_gat(fx.t('x'), edge, W, a, opt['heads'], p['alpha_train'], p['beta_train'], fx.t('x0'),
                opt['no_alpha_sigmoid'], opt['add_source'], opt['leaky_relu_slope'], opt['attention_norm_idx'],
                opt['mix_features'], p['multihead_att_layer.Wout'])
  assert_parity(f, fx.t('f'), TIGHT, 'f')


@pytest.mark.parametrize('name', fixtures('func_laplacian_'))
def test_laplacian_function(name):
  fx = Fixture(name)
  p, opt = fx.params, fx.opt
  w = fx.t('attention_weights') if opt['block'] in ('attention', 'hard_attention', 'mixed') else fx.t('edge_weight')
  f = R.rhs_laplacian(fx.t('x'), fx.t('func_edge_index'), w, p['alpha_train'], p['beta_train'], fx.t('x0'),
                      opt['no_alpha_sigmoid'], opt['add_source'])
  assert_parity(f, fx.t('f'), TIGHT, 'f')


def test_normalisations():
  fx = Fixture('norms')
  ei, w = fx.t('edge_index'), fx.t('edge_weight')
  for fill in (0.0, 0.3, 1.0, 3.2):
    for nd in (0, 1):
      e2, w2 = R.get_rw_adj(ei, w, nd, fill, 50)
      assert torch.equal(e2, fx.t('rw_ei_f%g_n%d' % (fill, nd)))
      assert_parity(w2, fx.t('rw_w_f%g_n%d' % (fill, nd)), TIGHT, 'rw')
    e2, w2 = R.gcn_norm_fill_val(ei, w, fill, 50)
    assert torch.equal(e2, fx.t('gcn_ei_f%g' % fill))
    assert_parity(w2, fx.t('gcn_w_f%g' % fill), TIGHT, 'gcn')
  # reference test_function_laplacian_diffusion.py::test_block_toy identity: column sums of the rw weights are 1
  e2, w2 = R.get_rw_adj(ei, None, 1, 1.0, 50)
  assert torch.allclose(R.scatter_sum(w2, e2[1], 50), torch.ones(50), atol=1e-6)


def _block_rhs(fx):
  """f(t,x) of a block fixture assembled from oracle pieces (parameters of `odefunc.`)."""
  opt, p = fx.opt, fx.params
  x = fx.t('x')
  n = x.shape[0]
  ei = fx.t('edge_index')
  x0 = x.clone()
  pre = 'odefunc.'
  if opt['data_norm'] == 'rw':
    e_n, w_n = R.get_rw_adj(ei, None, 1, opt['self_loop_weight'], n)
  else:
    e_n, w_n = R.gcn_norm_fill_val(ei, None, opt['self_loop_weight'], n)
  if opt['function'] == 'laplacian':
    w = w_n
    if opt['block'] in ('attention', 'mixed', 'hard_attention', 'rewire_attention'):
      # block-level attention layer, evaluated once at x(0); the mixed block's layer carries no edge weights
      ew = None if opt['block'] == 'mixed' else w_n
      if _is_split(opt):
        att, _ = _split_attention(opt, p, '', x, e_n, ew)
      else:
        att, _ = R.transformer_attention(x, e_n, p['multihead_att_layer.Q.weight'], p['multihead_att_layer.Q.bias'],
                                         p['multihead_att_layer.K.weight'], p['multihead_att_layer.K.bias'], opt['heads'],
                                         edge_weights=ew, reweight=opt['reweight_attention'], **_att_kwargs(opt))
      w = att
      if opt['block'] == 'mixed':        # block_mixed.py:41-45
        gam = torch.sigmoid(p['gamma'])
        w = att.mean(dim=1) * (1 - gam) + w_n * gam
      elif opt['block'] in ('hard_attention', 'rewire_attention'):   # eval mode: all edges, head-mean attention
        w = att.mean(dim=1)                    # (block_transformer_hard_attention.py:67-69, block_transformer_rewiring.py:203-207)
    return lambda t, y: R.rhs_laplacian(y, e_n, w, p[pre + 'alpha_train'], p[pre + 'beta_train'], x0,
                                        opt['no_alpha_sigmoid'], opt['add_source'])
  # transformer / GAT functions use their own self-loop-augmented edge list, not the block's -- except under the
  # hard-attention block, which overwrites the function's edge_index with its own (:68)
  edge, _ = R.add_remaining_self_loops(ei, None, opt['self_loop_weight'], int(ei.max()) + 1)
  if opt['block'] in ('hard_attention', 'rewire_attention'):
    edge = e_n
  if opt['function'] == 'transformer':
    return _transformer_rhs(fx, p, pre, edge, x0)
  return lambda t, y: R.rhs_gat(y, edge, p[pre + 'multihead_att_layer.W'], p[pre + 'multihead_att_layer.a'], opt['heads'],
                                p[pre + 'alpha_train'], p[pre + 'beta_train'], x0, opt['no_alpha_sigmoid'],
                                opt['add_source'], opt['leaky_relu_slope'], opt['attention_norm_idx'])


@pytest.mark.parametrize('name', [n for n in fixtures('block_') if 'dopri5' not in n] + fixtures('rewire_') +
                         [n for n in fixtures('train_') if 'dopri5' not in n])
def test_block_fixed_step(name):
  fx = Fixture(name)
  z = R.odeint_fixed(_block_rhs(fx), fx.t('x'), fx.opt['time'], fx.opt['step_size'], fx.opt['method'])
  assert_parity(z, fx.t('z'), TIGHT, name)


@pytest.mark.parametrize('name', [n for n in fixtures('early_') if 'rk4' in n])
def test_early_stop_rk4(name):
  """Oracle restatement of EarlyStopRK4 against the reference's own run (best step and every step's accuracies)."""
  fx = Fixture(name)
  masks = [fx.t(k).bool() for k in ('train_mask', 'val_mask', 'test_mask')]
  z, best, steps = R.odeint_rk4_early_stop(_block_rhs(fx), fx.t('x'), fx.opt['earlystopxT'] * fx.opt['time'],
                                           fx.opt['step_size'], fx.t('m2_weight'), fx.t('m2_bias'), fx.t('labels'), masks)
  assert_parity(z, fx.t('z'), TIGHT, name)
  assert np.allclose(np.array(steps), fx.arr['steps'], rtol=0, atol=1e-6)
  assert np.allclose(np.array(best), fx.arr['best'], rtol=0, atol=1e-6)


def test_time_grid_short_last_step():
  """T = 18.2948 (Cora best_params), step 1 -> 19 steps, the last of 0.2948."""
  g = R.time_grid(18.294754260552843, 1.0)
  assert len(g) == 20 and abs(float(g[-1] - g[-2]) - 0.294754) < 1e-5
  assert len(R.time_grid(100.0, 1.0)) == 101 and len(R.time_grid(4.0, 1.0)) == 5


def test_symmetric_attention_is_half():
  """reference test_transformer_attention.py::test_symmetric_attention (:92-99)."""
  edge = torch.tensor([[0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1]])
  W = torch.full((32, 2), 1e-5)
  att, _ = R.transformer_attention(torch.ones(3, 2), edge, W, torch.zeros(32), W, torch.zeros(32), 2)
  assert torch.all(torch.eq(att, 0.5 * torch.ones(6, 2)))


def test_head_aggregation_linearity():
  """reference test_head_aggregation (:110-121): mean_h spmm(a_h, x) == spmm(mean_h a_h, x)."""
  edge = torch.tensor([[0, 2, 2, 1], [1, 0, 1, 2]])
  x = torch.tensor([[1., 2.], [3., 2.], [4., 5.]])
  g = torch.Generator().manual_seed(0)
  att = torch.rand(4, 4, generator=g)
  a1 = torch.mean(torch.stack([R.spmm(edge, att[:, i], 3, x) for i in range(4)]), dim=0)
  assert torch.allclose(a1, R.spmm(edge, att.mean(dim=1), 3, x))


@pytest.mark.parametrize('name', fixtures('train_'))
def test_training_without_the_adjoint_method(name):
  """opt['adjoint'] = False (reference default; best_params Cora / Citeseer): the reference's block in training mode, loss.backward()
  through the solver -- recorded from the reference's own code over oracle/shims (torchdiffeq's rk_common with _UncheckedAssign, its
  controller under no_grad).  The oracle's right-hand side under this package's differentiable host loops with torch CPU autograd
  reproduces output, evaluation count and every gradient."""
  import importlib
  O = importlib.import_module('gnpde_amd.odeint')
  fx = Fixture(name)
  opt = fx.opt
  p = {k: v.clone().requires_grad_(True) for k, v in fx.params.items()}
  x = fx.t('x').clone().requires_grad_(True)
  x0 = fx.t('x')
  n = x.shape[0]
  ei = fx.t('edge_index')
  e_n, w_n = R.get_rw_adj(ei, None, 1, opt['self_loop_weight'], n)
  calls = [0]
  if opt['function'] == 'laplacian':
    w = w_n
    if opt['block'] == 'attention':
      w, _ = R.transformer_attention(x, e_n, p['multihead_att_layer.Q.weight'], p['multihead_att_layer.Q.bias'], p['multihead_att_layer.K.weight'],
                                     p['multihead_att_layer.K.bias'], opt['heads'], edge_weights=w_n, reweight=opt['reweight_attention'], **_att_kwargs(opt))

    def rhs(t, y):
      calls[0] += 1
      return R.rhs_laplacian(y, e_n, w, p['odefunc.alpha_train'], p['odefunc.beta_train'], x0, opt['no_alpha_sigmoid'], opt['add_source'])
  elif opt['function'] == 'GAT':
    edge, _ = R.add_remaining_self_loops(ei, None, opt['self_loop_weight'], int(ei.max()) + 1)
    pre = 'odefunc.multihead_att_layer.'

    def rhs(t, y):
      calls[0] += 1
      return R.rhs_gat(y, edge, p[pre + 'W'], p[pre + 'a'], opt['heads'], p['odefunc.alpha_train'], p['odefunc.beta_train'], x0,
                       opt['no_alpha_sigmoid'], opt['add_source'], opt['leaky_relu_slope'], opt['attention_norm_idx'])
  else:
    edge, _ = R.add_remaining_self_loops(ei, None, opt['self_loop_weight'], int(ei.max()) + 1)
    pre = 'odefunc.multihead_att_layer.'

    def rhs(t, y):
      calls[0] += 1
      return R.rhs_transformer(y, edge, p[pre + 'Q.weight'], p[pre + 'Q.bias'], p[pre + 'K.weight'], p[pre + 'K.bias'], opt['heads'],
                               p['odefunc.alpha_train'], p['odefunc.beta_train'], x0, opt['no_alpha_sigmoid'], opt['add_source'], **_att_kwargs(opt))
  t = torch.tensor([0, opt['time']])
  if opt['method'] == 'dopri5':
    z = O._solve_dopri5(rhs, x, t, opt['tol_scale'] * 1e-9, opt['tol_scale'] * 1e-7)[1]
  else:
    z = O._solve_fixed_host(rhs, x, t, opt['method'], opt['step_size'])[1]
  assert calls[0] == int(fx.arr['nfe']) == int(fx.arr['nfe_after_backward'])       # the backward evaluates nothing
  assert_parity(z, fx.t('z'), 1e-5, name + ' z')
  (z * fx.t('c')).sum().backward()
  assert_parity(x.grad, fx.t('grad_x'), 1e-4, name + ' grad_x')
  # (a gradient that is zero in exact arithmetic -- K.bias under a softmax over rows: the scores of a row all move by q_i . b -- is
  #  rounding noise on both sides: errors are measured against the largest gradient of the same module)
  keys = [k for k in fx.arr if k.startswith('grad/')]
  scale = {}
  for k in keys:
    mod = k[5:].rsplit('.', 2)[0] if 'multihead' in k else k
    scale[mod] = max(scale.get(mod, 0.0), float(fx.t(k).abs().max()))
  for k in keys:
    got = p[k[5:]].grad
    assert got is not None, k
    mod = k[5:].rsplit('.', 2)[0] if 'multihead' in k else k
    err = float((got.reshape(fx.arr[k].shape) - fx.t(k)).abs().max())
    assert err <= 1e-4 * scale[mod], '%s %s: abs err %.3e against module scale %.3e' % (name, k, err, scale[mod])
