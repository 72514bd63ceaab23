"""Regularised training (reference src/regularized_ODE_function.py + registry src/base_classes.py:10-29):
state, the integral of every regulariser and the gradients of  <z, c> + sum_j coeff_j mean(reg_j)  against fixtures
recorded from the reference's own RegularizedODEfunc / ConstantODEblock (oracle/gen_golden.py gen_regularised)."""
import pytest
import torch

import gnpde_amd as G
from helpers import Fixture, fixtures, Data, assert_parity

pytestmark = pytest.mark.gpu

FUNCS = {'laplacian': G.LaplacianODEFunc, 'transformer': G.ODEFuncTransformerAtt, 'GAT': G.ODEFuncAtt}
GTOL = 2e-4


@pytest.mark.parametrize('name', fixtures('reg_'))
def test_regularised_training_forward_and_gradients(dev, name):
  fx = Fixture(name)
  opt = fx.opt
  fns, coeffs = G.create_regularization_fns(opt)
  assert len(fns) == len(fx.arr['coeffs']) and list(coeffs) == list(fx.arr['coeffs'])
  x = fx.t('x', dev)
  block = G.ConstantODEblock(FUNCS[opt['function']], fns, opt, Data(x, fx.t('edge_index', dev)), dev,
                             t=torch.tensor([0, opt['time']])).to(dev)
  block.load_state_dict(fx.params, strict=True)
  if opt['function'] == 'GAT':     # plain-tensor parameters of the GAT layer (reference quirk) take their gradients here
    for f in (block.odefunc, block.reg_odefunc.odefunc):
      lay = f.multihead_att_layer
      for nm in ('W', 'Wout', 'a'):
        getattr(lay, nm).requires_grad_(True)
  block.train()
  xin = x.clone().requires_grad_(True)
  block.set_x0(xin)
  z, regs = block(xin)
  assert block.nreg == len(regs) == len(fns)
  # the regularised solve integrates the block's FIRST function object (reg_odefunc.odefunc), as in the reference
  assert block.reg_odefunc.odefunc.nfe == int(fx.arr['nfe']) and block.odefunc.nfe == 0
  assert_parity(z, fx.t('z'), what=name + ' z')
  for j, r in enumerate(regs):
    ref = fx.t('reg%d' % j)
    assert r.shape == ref.shape
    if float(ref.abs().max()) < 1e-5:       # jacobian_norm2 of a column-stochastic Laplacian: rounding-level values
      assert float((r.detach().cpu() - ref).abs().max()) < 1e-5
    else:
      # (integrals of SQUARED first / second derivatives of f evaluated in fp32 by two different op orders)
      assert_parity(r, ref, tol=1e-4, what='%s reg%d' % (name, j))
  loss = (z * fx.t('c', dev)).sum() + sum(cf * r.mean() for cf, r in zip(coeffs, regs))
  loss.backward()
  assert_parity(xin.grad, fx.t('grad_x'), tol=GTOL, what=name + ' grad_x')
  grads = dict(block.named_parameters())
  checked = 0
  for k in fx.arr:
    if not k.startswith('grad/'):
      continue
    ref = fx.t(k)
    got = grads[k[5:]].grad
    assert got is not None, k
    if float(ref.abs().max()) < 1e-6:
      assert float(got.abs().max()) < 1e-4, k
    else:
      assert_parity(got, ref, tol=GTOL, what=name + ' ' + k)
    checked += 1
  assert checked >= 2


def test_total_derivative_raises_like_the_reference(dev):
  fx = Fixture('reg_transformer_rk4_kinetic')
  opt = dict(fx.opt, kinetic_energy=None, total_deriv=0.1)
  fns, _ = G.create_regularization_fns(opt)
  x = fx.t('x', dev)
  block = G.ConstantODEblock(G.ODEFuncTransformerAtt, fns, opt, Data(x, fx.t('edge_index', dev)), dev,
                             t=torch.tensor([0, opt['time']])).to(dev)
  block.train()
  block.set_x0(x)
  with pytest.raises(RuntimeError, match='No partial derivative with respect to time'):
    block(x.clone().requires_grad_(True))


def test_no_regulariser_in_eval_mode(dev):
  fx = Fixture('reg_laplacian_rk4_all')
  opt = fx.opt
  fns, _ = G.create_regularization_fns(opt)
  x = fx.t('x', dev)
  block = G.ConstantODEblock(G.LaplacianODEFunc, fns, opt, Data(x, fx.t('edge_index', dev)), dev,
                             t=torch.tensor([0, opt['time']])).to(dev)
  block.load_state_dict(fx.params, strict=True)
  block.eval()
  block.set_x0(x)
  with torch.no_grad():
    z = block(x)
  assert torch.is_tensor(z) and z.shape == x.shape      # eval: plain state, second function object, native solver
