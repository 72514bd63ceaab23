"""HIP path vs the golden vectors recorded from the reference's own code (oracle/gen_golden.py)."""
import pytest
import torch

import gnpde_amd as G
from helpers import Fixture, fixtures, Data, assert_parity

pytestmark = pytest.mark.gpu

FUNCS = {'laplacian': G.LaplacianODEFunc, 'transformer': G.ODEFuncTransformerAtt, 'GAT': G.ODEFuncAtt}
BLOCKS = {'constant': G.ConstantODEblock, 'attention': G.AttODEblock, 'mixed': G.MixedODEblock,
          'hard_attention': G.HardAttODEblock}


@pytest.mark.parametrize('name', fixtures('func_transformer_') + fixtures('func_gat_'))
def test_function_forward(dev, name):
  fx = Fixture(name)
  x = fx.t('x', dev)
  cls = G.ODEFuncTransformerAtt if 'transformer' in name else G.ODEFuncAtt
  func = cls(x.shape[1], x.shape[1], fx.opt, Data(x, fx.t('edge_index', dev)), dev).to(dev)
  missing, unexpected = func.load_state_dict(fx.params, strict=True)
  assert torch.equal(func.edge_index.cpu(), fx.t('func_edge_index')), 'self-loop insertion differs from the reference'
  func.x0 = fx.t('x0', dev)
  att, aux = func.multihead_att_layer(x, func.edge_index)
  assert_parity(att, fx.t('attention'), what=name + ' attention')
  if 'transformer' in name:
    assert_parity(aux[1], fx.t('prods'), what=name + ' prods')
  else:
    assert_parity(aux, fx.t('wx'), what=name + ' wx')
  with torch.no_grad():
    f = func(0.0, x)
  assert func.nfe == 1
  assert_parity(f, fx.t('f'), what=name + ' f')


def test_layer_reweight(dev):
  fx = Fixture('layer_reweight')
  x = fx.t('x', dev)
  layer = G.SpGraphTransAttentionLayer(x.shape[1], x.shape[1], fx.opt, dev, edge_weights=fx.t('edge_weight', dev)).to(dev)
  layer.load_state_dict(fx.params, strict=True)
  att, (v, prods) = layer(x, fx.t('func_edge_index', dev))
  assert_parity(prods, fx.t('prods'), what='prods')
  assert_parity(att, fx.t('attention'), what='attention')


@pytest.mark.parametrize('name', fixtures('func_laplacian_'))
def test_laplacian_forward(dev, name):
  fx = Fixture(name)
  x = fx.t('x', dev)
  func = G.LaplacianODEFunc(x.shape[1], x.shape[1], fx.opt, Data(x, fx.t('edge_index', dev)), dev).to(dev)
  func.load_state_dict(fx.params, strict=True)
  func.edge_index, func.edge_weight = fx.t('func_edge_index', dev), fx.t('edge_weight', dev)
  func.attention_weights = fx.t('attention_weights', dev)
  func.x0 = fx.t('x0', dev)
  with torch.no_grad():
    f = func(0.0, x)
  assert_parity(f, fx.t('f'), what=name)


@pytest.mark.parametrize('name', fixtures('block_'))
@pytest.mark.parametrize('use_graph', [True, False])
def test_block_forward(dev, name, use_graph):
  fx = Fixture(name)
  if fx.opt['method'] == 'dopri5' and not use_graph:
    pytest.skip('adaptive solver has no graph/eager split')
  x = fx.t('x', dev)
  data = Data(x, fx.t('edge_index', dev))
  block = BLOCKS[fx.opt['block']](FUNCS[fx.opt['function']], [], fx.opt, data, dev,
                                  t=torch.tensor([0, fx.opt['time']])).to(dev)
  block.load_state_dict(fx.params, strict=True)
  block.eval()
  if not use_graph:
    import functools
    block.test_integrator = functools.partial(G.odeint, use_graph=False)
  block.set_x0(x)
  with torch.no_grad():
    z = block(x)
  # dopri5: accept/reject decisions amplify rounding; the solver tolerance itself is >= 1e-5 here
  tol = 1e-5 if fx.opt['method'] != 'dopri5' else max(1e-5, 20 * fx.opt['tol_scale'] * 1e-7)
  assert_parity(z, fx.t('z'), tol=tol, what=name)
  assert block.odefunc.nfe == int(fx.arr['nfe']), 'nfe %d vs reference %d' % (block.odefunc.nfe, int(fx.arr['nfe']))
  # second forward reuses the captured graph and must reproduce the result bit for bit
  block.set_x0(x)
  with torch.no_grad():
    z2 = block(x)
  if fx.opt['method'] != 'dopri5':
    assert torch.equal(z, z2), 'replay is not deterministic'


def _gnn_of(fx, dev):
  feat = int(fx.arr['num_features']) if 'num_features' in fx.arr else fx.arr['x'].shape[1]
  data = Data(fx.t('x', dev)[:, :feat], fx.t('edge_index', dev))
  model = G.GNN(dict(fx.opt), G.DummyDataset(data, int(fx.arr['num_classes'])), dev).to(dev)
  model.load_state_dict(fx.params, strict=True)
  return model.eval()


@pytest.mark.parametrize('name', fixtures('gnn_') + fixtures('gnnopt_'))
def test_gnn_end_to_end(dev, name):
  """Encoder -> ODE block -> decoder of the reference's GNN.forward (src/GNN.py:17-72, eval mode) through gnpde_amd.GNN:
  native encoder / relu+decoder launches around the native block, with the reference's state_dict loaded strictly.
  gnnopt_*: use_mlp + fc_out, batch_norm + augment, use_labels, beltrami."""
  fx = Fixture(name)
  model = _gnn_of(fx, dev)
  pos = fx.t('pos', dev) if 'pos' in fx.arr else None
  with torch.no_grad():
    out = model(fx.t('x', dev), pos)
  assert_parity(out, fx.t('out'), what=name)
  assert model.getNFE() == int(fx.arr['nfe'])
  model.resetNFE()
  assert model.getNFE() == 0
  # the autograd route (PyTorch encoder / decoder around the native block) gives the same answer
  with torch.enable_grad():
    out_t = model(fx.t('x', dev), pos)
  assert out_t.requires_grad
  assert_parity(out_t.detach(), fx.t('out'), what=name + ' (grad mode)')


def test_gnn_runs_the_reference_train_and_test_step(dev):
  """run_GNN.py's train() / test() protocol on the native GNN (src/run_GNN.py:62-96, 137-148): forward in train mode,
  loss, `model.fm.update(model.getNFE())`, resetNFE, backward, optimiser step, `model.bm.update(...)`; then an eval
  forward.  The meters are what the epoch log prints (`model.fm.sum`, `model.bm.sum`)."""
  fx = Fixture('gnn_constant_transformer_rk4')
  model = _gnn_of(fx, dev)
  n, c = fx.arr['x'].shape[0], int(fx.arr['num_classes'])
  gen = torch.Generator().manual_seed(0)
  y = torch.randint(0, c, (n, 1), generator=gen).to(dev)
  mask = (torch.rand(n, generator=gen) < 0.5).to(dev)
  optim = torch.optim.Adam(model.parameters(), lr=1e-3)
  assert isinstance(model.fm, G.Meter) and model.fm.cnt == 0 and model.bm.sum == 0
  before = [p.detach().clone() for p in model.parameters()]
  for step in range(2):
    model.train()
    optim.zero_grad()
    out = model(fx.t('x', dev), None)
    loss = torch.nn.CrossEntropyLoss()(out[mask], y.squeeze()[mask])
    model.fm.update(model.getNFE())
    model.resetNFE()
    loss.backward()
    optim.step()
    model.bm.update(model.getNFE())
    model.resetNFE()
    assert torch.isfinite(loss)
  assert model.fm.cnt == 2 and model.fm.sum == 2 * int(fx.arr['nfe']) and model.fm.get_value() == int(fx.arr['nfe'])
  assert model.fm.get_average() == int(fx.arr['nfe']) and model.bm.cnt == 2
  assert any(not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
  model.eval()
  with torch.no_grad():
    logits = model(fx.t('x', dev), None)
  pred = logits[mask].max(1)[1]
  acc = pred.eq(y.squeeze()[mask]).sum().item() / mask.sum().item()
  assert 0.0 <= acc <= 1.0


def test_max_nfe(dev):
  fx = Fixture('block_constant_transformer_rk4')
  opt = dict(fx.opt, max_nfe=5)
  x = fx.t('x', dev)
  block = G.ConstantODEblock(G.ODEFuncTransformerAtt, [], opt, Data(x, fx.t('edge_index', dev)), dev,
                             t=torch.tensor([0, opt['time']])).to(dev)
  block.eval()
  block.set_x0(x)
  with torch.no_grad(), pytest.raises(G.MaxNFEException):
    block(x)
  f = block.odefunc
  f.nfe = 0
  with torch.no_grad():
    for _ in range(6):
      f(0.0, x)
    with pytest.raises(G.MaxNFEException):
      f(0.0, x)


@pytest.mark.parametrize('name', fixtures('block_hard_'))
def test_hard_attention_training_mode_sampling(dev, name):
  """Training-mode forward of the hard-attention block (no gradients needed for the check): the quantile edge
  selection must pick exactly the reference's edges, the renormalised weights and the state must match."""
  fx = Fixture(name)
  x = fx.t('x', dev)
  block = G.HardAttODEblock(FUNCS[fx.opt['function']], [], fx.opt, Data(x, fx.t('edge_index', dev)), dev,
                            t=torch.tensor([0, fx.opt['time']])).to(dev)
  block.load_state_dict(fx.params, strict=True)
  block.train()
  block.set_x0(x)
  with torch.no_grad():
    z = block(x)
  assert torch.equal(block.odefunc.edge_index.cpu(), fx.t('train_edge_index')), 'a different edge subset was sampled'
  assert_parity(block.odefunc.attention_weights, fx.t('train_attention'), what='renormalised attention')
  assert_parity(z, fx.t('z_train'), what='z (training mode)')
