"""Generate tests/golden/*.npz by running the REFERENCE's own code (container-only).

Imports /root/reference/src/*.py unmodified (over oracle/shims) and records, for seeded synthetic
inputs with RANDOM (never the constant-1e-5 init) weights:
  * one right-hand-side evaluation f(t,x) of each ODEFunc,
  * attention [E,h] in the reference's edge order (+ raw scores `prods`),
  * the state after a whole ODEblock.forward (euler / rk4 with a short last step / dopri5),
  * the rw / sym normalisations, and one end-to-end GNN.forward.
Each fixture holds its inputs (raw edge_index, x, opt as JSON, the module's full state_dict), so
the tests on the GPU box need neither the reference nor this script.

Usage (in the build container):  python oracle/gen_golden.py
"""
import os
import sys
import json
import copy
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_env  # noqa: E402

ref_env.activate()
from test_params import OPT  # noqa: E402  (reference test/test_params.py)
from torch_geometric.data import Data  # noqa: E402  (stand-in)
from utils import DummyDataset, get_rw_adj, gcn_norm_fill_val  # noqa: E402
from function_transformer_attention import ODEFuncTransformerAtt, SpGraphTransAttentionLayer  # noqa: E402
from function_GAT_attention import ODEFuncAtt  # noqa: E402
from function_laplacian_diffusion import LaplacianODEFunc  # noqa: E402
from block_constant import ConstantODEblock  # noqa: E402
from block_transformer_attention import AttODEblock  # noqa: E402
from block_mixed import MixedODEblock  # noqa: E402
from block_transformer_hard_attention import HardAttODEblock  # noqa: E402
from block_transformer_rewiring import RewireAttODEblock  # noqa: E402
from GNN import GNN  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

BASE = {**OPT, 'mix_features': False, 'attention_dim': 16, 'heads': 4, 'hidden_dim': 24,
        'max_nfe': 100000, 'add_source': True, 'block': 'constant', 'function': 'transformer',
        'method': 'rk4', 'step_size': 1.0, 'time': 2.3, 'self_loop_weight': 1, 'data_norm': 'rw',
        'tol_scale': 1.0, 'adjoint': False, 'augment': False, 'max_iters': 100}


def make_graph(n, avg_deg, seed, with_loops=False):
  """Random undirected simple graph, both directions listed, edges in shuffled order."""
  g = torch.Generator().manual_seed(seed)
  m = n * avg_deg // 2
  a = torch.randint(0, n, (m,), generator=g)
  b = torch.randint(0, n, (m,), generator=g)
  keep = a != b
  a, b = a[keep], b[keep]
  key = torch.unique(torch.minimum(a, b) * n + torch.maximum(a, b))
  a, b = key // n, key % n
  ei = torch.cat([torch.stack([a, b]), torch.stack([b, a])], dim=1)
  if with_loops:
    loops = torch.arange(0, n, 7)
    ei = torch.cat([ei, torch.stack([loops, loops])], dim=1)
  ei = ei[:, torch.randperm(ei.size(1), generator=g)]
  return ei.long()


def randomise(module, seed):
  """Replace every parameter by seeded noise with a sensible scale."""
  g = torch.Generator().manual_seed(seed)
  with torch.no_grad():
    for name, p in module.named_parameters():
      if p.dim() >= 2:
        p.copy_(torch.randn(p.shape, generator=g) / np.sqrt(p.shape[-1]) * 1.5)
      elif name.endswith('alpha_train') or name.endswith('beta_train'):
        p.copy_(torch.randn(p.shape, generator=g) * 0.5)
      elif 'lengthscale' in name or 'output_var' in name:
        p.copy_(1.0 + 0.3 * torch.rand(p.shape, generator=g))
      else:
        p.copy_(torch.randn(p.shape, generator=g) * 0.1)
  # GAT layer keeps W, Wout, a as plain attributes (`nn.Parameter(...).to(device)` on cpu is still a Parameter)


def save(name, opt, tensors, module=None):
  rec = {'opt_json': np.frombuffer(json.dumps(opt, sort_keys=True).encode(), dtype=np.uint8)}
  for k, v in tensors.items():
    rec[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
  if module is not None:
    for k, v in module.state_dict().items():
      rec['param/' + k] = v.detach().cpu().numpy()
  path = os.path.join(OUT, name + '.npz')
  np.savez_compressed(path, **rec)
  print('%-44s %6.1f KB' % (name, os.path.getsize(path) / 1024))


def data_of(ei, x, edge_attr=None):
  d = Data(x=x, edge_index=ei, edge_attr=edge_attr)
  return d


def gen_norms():
  ei = make_graph(50, 6, 11, with_loops=True)
  g = torch.Generator().manual_seed(12)
  w = torch.rand(ei.size(1), generator=g) + 0.1
  rec = {'edge_index': ei, 'edge_weight': w}
  for fill in (0.0, 0.3, 1.0, 3.2):
    for nd in (0, 1):
      e2, w2 = get_rw_adj(ei, edge_weight=w, norm_dim=nd, fill_value=fill, num_nodes=50, dtype=torch.float32)
      rec['rw_ei_f%g_n%d' % (fill, nd)] = e2
      rec['rw_w_f%g_n%d' % (fill, nd)] = w2
    e2, w2 = gcn_norm_fill_val(ei, edge_weight=w, fill_value=fill, num_nodes=50, dtype=torch.float32)
    rec['gcn_ei_f%g' % fill] = e2
    rec['gcn_w_f%g' % fill] = w2
  e2, w2 = get_rw_adj(ei, edge_weight=None, norm_dim=1, fill_value=1.0, num_nodes=50, dtype=torch.float32)
  rec['rw_ei_unweighted'] = e2
  rec['rw_w_unweighted'] = w2
  save('norms', {}, rec)


def gen_funcs():
  n, d = 180, 24
  ei = make_graph(n, 8, 1, with_loops=True)
  g = torch.Generator().manual_seed(2)
  x = torch.randn(n, d, generator=g)
  x0 = torch.randn(n, d, generator=g)
  variants = {
    'sd_softmax_n0': {},
    'sd_softmax_n1': {'attention_norm_idx': 1},
    'sd_squareplus_n1': {'attention_norm_idx': 1, 'square_plus': True},
    'sd_squareplus_n0': {'square_plus': True},
    'cosine_n0': {'attention_type': 'cosine_sim'},
    'pearson_n1': {'attention_type': 'pearson', 'attention_norm_idx': 1},
    'expkernel_n0': {'attention_type': 'exp_kernel'},
    'sd_rawalpha_nosrc': {'no_alpha_sigmoid': True, 'add_source': False},
    'sd_h1_A24': {'heads': 1, 'attention_dim': 24},
    'sd_noloops': {'self_loop_weight': 0},
  }
  for i, (name, over) in enumerate(variants.items()):
    opt = {**BASE, **over}
    func = ODEFuncTransformerAtt(d, d, opt, data_of(ei, x), torch.device('cpu'))
    randomise(func, 100 + i)
    func.x0 = x0
    with torch.no_grad():
      att, (v, prods) = func.multihead_att_layer(x, func.edge_index)
      f = func(0.0, x)
    save('func_transformer_' + name, opt,
         {'edge_index': ei, 'x': x, 'x0': x0, 'func_edge_index': func.edge_index,
          'attention': att, 'prods': prods, 'f': f}, func)

  # layer with reweight_attention (as constructed inside AttODEblock, block_transformer_attention.py:29-30)
  opt = {**BASE, 'reweight_attention': True}
  e2, w2 = get_rw_adj(ei, edge_weight=None, norm_dim=1, fill_value=1, num_nodes=n, dtype=torch.float32)
  layer = SpGraphTransAttentionLayer(d, d, opt, torch.device('cpu'), edge_weights=w2)
  randomise(layer, 150)
  with torch.no_grad():
    att, (v, prods) = layer(x, e2)
  save('layer_reweight', opt, {'edge_index': ei, 'x': x, 'func_edge_index': e2, 'edge_weight': w2,
                               'attention': att, 'prods': prods}, layer)

  # GAT
  for i, (name, over) in enumerate({'n0': {}, 'n1': {'attention_norm_idx': 1},
                                    'mix': {'mix_features': True},
                                    'slope': {'leaky_relu_slope': 0.05, 'heads': 2}}.items()):
    opt = {**BASE, 'function': 'GAT', **over}
    func = ODEFuncAtt(d, d, opt, data_of(ei, x), torch.device('cpu'))
    randomise(func, 200 + i)
    func.x0 = x0
    with torch.no_grad():
      att, wx = func.multihead_att_layer(x, func.edge_index)
      f = func(0.0, x)
    save('func_gat_' + name, opt, {'edge_index': ei, 'x': x, 'x0': x0, 'func_edge_index': func.edge_index,
                                   'attention': att, 'wx': wx, 'f': f}, func)

  # Laplacian with each weight source (function_laplacian_diffusion.py:28-36)
  e2, w2 = get_rw_adj(ei, edge_weight=None, norm_dim=1, fill_value=1, num_nodes=n, dtype=torch.float32)
  att_h = torch.rand(e2.size(1), 4, generator=g)
  for i, (name, blk) in enumerate({'constant': 'constant', 'attention': 'attention', 'hard': 'hard_attention'}.items()):
    opt = {**BASE, 'function': 'laplacian', 'block': blk}
    func = LaplacianODEFunc(d, d, opt, data_of(ei, x), torch.device('cpu'))
    randomise(func, 300 + i)
    func.edge_index, func.edge_weight = e2, w2
    func.attention_weights = att_h if blk == 'attention' else att_h[:, 0].contiguous()
    func.x0 = x0
    with torch.no_grad():
      f = func(0.0, x)
    save('func_laplacian_' + name, opt, {'edge_index': ei, 'x': x, 'x0': x0, 'func_edge_index': e2,
                                         'edge_weight': w2, 'attention_weights': func.attention_weights,
                                         'f': f}, func)


def gen_blocks():
  n, d = 150, 24
  ei = make_graph(n, 6, 21)
  g = torch.Generator().manual_seed(22)
  x = torch.randn(n, d, generator=g)
  cases = {
    'constant_laplacian_euler': dict(block='constant', function='laplacian', method='euler', time=4.0),
    'constant_laplacian_gcn_rk4': dict(block='constant', function='laplacian', method='rk4', time=3.0, data_norm='gcn'),
    'constant_transformer_rk4': dict(block='constant', function='transformer', method='rk4', time=2.3),
    'constant_transformer_sqp_n1_rk4': dict(block='constant', function='transformer', method='rk4', time=3.2948,
                                            square_plus=True, attention_norm_idx=1),
    'constant_transformer_euler_h05': dict(block='constant', function='transformer', method='euler', time=2.2,
                                           step_size=0.5),
    'constant_gat_rk4': dict(block='constant', function='GAT', method='rk4', time=2.3),
    'attention_laplacian_euler': dict(block='attention', function='laplacian', method='euler', time=3.0),
    'attention_laplacian_rk4_sqp': dict(block='attention', function='laplacian', method='rk4', time=2.5,
                                        square_plus=True, attention_norm_idx=1, reweight_attention=True),
    'attention_laplacian_dopri5': dict(block='attention', function='laplacian', method='dopri5', time=3.0,
                                       tol_scale=800.0),
    'constant_transformer_dopri5': dict(block='constant', function='transformer', method='dopri5', time=2.0,
                                        tol_scale=100.0),
    'mixed_laplacian_rk4': dict(block='mixed', function='laplacian', method='rk4', time=2.3),
    'hard_laplacian_euler': dict(block='hard_attention', function='laplacian', method='euler', time=3.0,
                                 att_samp_pct=0.6, use_flux=False),
    'hard_transformer_rk4': dict(block='hard_attention', function='transformer', method='rk4', time=2.0,
                                 att_samp_pct=0.8, use_flux=False),
  }
  for i, (name, over) in enumerate(cases.items()):
    opt = {**BASE, **over}
    fcls = {'laplacian': LaplacianODEFunc, 'transformer': ODEFuncTransformerAtt, 'GAT': ODEFuncAtt}[opt['function']]
    bcls = {'constant': ConstantODEblock, 'attention': AttODEblock, 'mixed': MixedODEblock,
            'hard_attention': HardAttODEblock}[opt['block']]
    t = torch.tensor([0, opt['time']])
    block = bcls(fcls, [], opt, data_of(ei, x), torch.device('cpu'), t=t)
    randomise(block, 400 + i)
    block.eval()
    block.set_x0(x)
    with torch.no_grad():
      z = block(x)
    extra = {}
    if opt['block'] == 'hard_attention':   # also the training-mode forward: quantile edge sampling + renormalisation
      nfe_eval = block.odefunc.nfe
      block.train()
      block.set_x0(x)
      import io, contextlib
      with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        z_train = block(x)
      extra = {'z_train': z_train, 'train_edge_index': block.odefunc.edge_index,
               'train_attention': block.odefunc.attention_weights}
      block.eval()
      block.odefunc.nfe = nfe_eval
    save('block_' + name, opt, dict({'edge_index': ei, 'x': x, 'z': z, 'nfe': np.int64(block.odefunc.nfe)}, **extra), block)


def gen_gnn():
  n, feat, classes = 120, 40, 5
  ei = make_graph(n, 6, 31)
  g = torch.Generator().manual_seed(32)
  xin = torch.randn(n, feat, generator=g)
  for i, (name, over) in enumerate({
      'constant_transformer_rk4': dict(block='constant', function='transformer', method='rk4', time=2.3),
      'attention_laplacian_euler': dict(block='attention', function='laplacian', method='euler', time=3.0)}.items()):
    opt = copy.deepcopy({**BASE, **over})
    data = data_of(ei, xin)
    model = GNN(opt, DummyDataset(data, classes), torch.device('cpu'))
    randomise(model, 500 + i)
    model.eval()
    with torch.no_grad():
      out = model(xin)
    save('gnn_' + name, opt, {'edge_index': ei, 'x': xin, 'out': out, 'num_classes': np.int64(classes),
                              'nfe': np.int64(model.getNFE())}, model)


def gen_gnn_options():
  """GNN.forward's optional branches at test time (GNN.py:17-72): use_mlp + fc_out, batch_norm (running statistics) +
  augment, use_labels (label columns bypass the encoder), beltrami (separate feature / positional encoders)."""
  n, feat, classes = 120, 40, 5
  ei = make_graph(n, 6, 33)
  g = torch.Generator().manual_seed(34)
  cases = {
    'mlp_fcout': (dict(use_mlp=True, fc_out=True, function='transformer', method='rk4', time=2.0), feat, 0),
    'bn_augment': (dict(batch_norm=True, augment=True, function='laplacian', method='euler', time=3.0), feat, 0),
    'labels': (dict(use_labels=True, function='laplacian', block='attention', method='rk4', time=2.0), feat + classes, 0),
    'beltrami': (dict(beltrami=True, function='transformer', attention_type='exp_kernel', feat_hidden_dim=16,
                      pos_enc_hidden_dim=8, pos_enc_dim=6, method='rk4', time=2.0), feat, 6),
  }
  for i, (name, (over, width, pos)) in enumerate(cases.items()):
    opt = copy.deepcopy({**BASE, **over})
    xin = torch.randn(n, width, generator=g)
    if over.get('use_labels'):
      xin[:, -classes:] = torch.nn.functional.one_hot(torch.randint(0, classes, (n,), generator=g), classes).float()
    pe = torch.randn(n, pos, generator=g) if pos else None
    data = data_of(ei, xin[:, :feat])
    opt_in = copy.deepcopy(opt)      # (the constructor rewrites opt['hidden_dim'] for use_labels / beltrami)
    model = GNN(opt, DummyDataset(data, classes), torch.device('cpu'))
    randomise(model, 520 + i)
    if over.get('batch_norm'):
      for bn in (model.bn_in, model.bn_out):
        bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * 0.2)
        bn.running_var.copy_(0.5 + torch.rand(bn.running_var.shape, generator=g))
    model.eval()
    with torch.no_grad():
      out = model(xin, pe) if pe is not None else model(xin)
    rec = {'edge_index': ei, 'x': xin, 'out': out, 'num_classes': np.int64(classes), 'num_features': np.int64(feat),
           'nfe': np.int64(model.getNFE())}
    if pe is not None:
      rec['pos'] = pe
    save('gnnopt_' + name, opt_in, rec, model)


def gen_beltrami():
  """BLEND attention: separate exp kernels on the feature and positional channels, multiplied
  (function_transformer_attention.py:83-101, :133-171); with and without label columns after the positional block."""
  n = 160
  ei = make_graph(n, 8, 61, with_loops=True)
  g = torch.Generator().manual_seed(62)
  for i, (name, feat, pos, lab, over) in enumerate([
      ('beltrami_expkernel', 16, 8, 0, {}),
      ('beltrami_expkernel_labels_n1', 16, 8, 3, {'attention_norm_idx': 1, 'heads': 2, 'attention_dim': 8}),
      ('beltrami_expkernel_sqp', 12, 12, 0, {'square_plus': True})]):
    d = feat + pos + lab
    x = torch.randn(n, d, generator=g)
    x0 = torch.randn(n, d, generator=g)
    opt = {**BASE, 'attention_type': 'exp_kernel', 'beltrami': True, 'feat_hidden_dim': feat, 'pos_enc_hidden_dim': pos,
           'hidden_dim': d, **over}
    func = ODEFuncTransformerAtt(d, d, opt, data_of(ei, x), torch.device('cpu'))
    randomise(func, 800 + i)
    func.x0 = x0
    with torch.no_grad():
      att, (v, prods) = func.multihead_att_layer(x, func.edge_index)
      f = func(0.0, x)
    assert v is None
    save('func_transformer_' + name, opt, {'edge_index': ei, 'x': x, 'x0': x0, 'func_edge_index': func.edge_index,
                                           'attention': att, 'prods': prods, 'f': f}, func)
  # a whole block: attention computed once with the split kernel, GRAND-l diffusion (the BLEND configuration C4)
  d, feat, pos = 24, 16, 8
  x = torch.randn(n, d, generator=g)
  opt = {**BASE, 'attention_type': 'exp_kernel', 'beltrami': True, 'feat_hidden_dim': feat, 'pos_enc_hidden_dim': pos,
         'hidden_dim': d, 'block': 'attention', 'function': 'laplacian', 'method': 'rk4', 'time': 2.5}
  block = AttODEblock(LaplacianODEFunc, [], opt, data_of(ei, x), torch.device('cpu'), t=torch.tensor([0, opt['time']]))
  randomise(block, 850)
  block.eval()
  block.set_x0(x)
  with torch.no_grad():
    z = block(x)
  save('block_attention_laplacian_beltrami_rk4', opt, {'edge_index': ei, 'x': x, 'z': z,
                                                       'nfe': np.int64(block.odefunc.nfe)}, block)


def gen_rewire():
  """RewireAttODEblock (src/block_transformer_rewiring.py): the eval forward, and training forwards whose rewiring
  (two-hop densification or random pairs, quantile threshold, renormalisation) is recorded edge by edge."""
  import io, contextlib
  n, d = 90, 24
  ei = make_graph(n, 4, 71)
  g = torch.Generator().manual_seed(72)
  x = torch.randn(n, d, generator=g)
  base = dict(block='rewire_attention', function='laplacian', method='rk4', time=2.0, att_samp_pct=1.0, use_flux=False,
              rw_addD=0.02, new_edges='k_hop_att', sparsify='S_hat')
  cases = {
    'khop_shat': {},
    'khop_recalc_flux': dict(sparsify='recalc_att', use_flux=True),
    'random': dict(new_edges='random', rw_addD=0.93),
    'khop_transformer': dict(function='transformer', method='euler'),
  }
  for i, (name, over) in enumerate(cases.items()):
    opt = {**BASE, **base, **over}
    fcls = {'laplacian': LaplacianODEFunc, 'transformer': ODEFuncTransformerAtt}[opt['function']]
    block = RewireAttODEblock(fcls, [], opt, data_of(ei, x), torch.device('cpu'), t=torch.tensor([0, opt['time']]))
    randomise(block, 900 + i)
    block.eval()
    block.set_x0(x)
    with torch.no_grad():
      z_eval = block(x)
    nfe_eval = block.odefunc.nfe
    rec = {'edge_index': ei, 'x': x, 'z': z_eval, 'nfe': np.int64(nfe_eval)}
    block.train()
    np.random.seed(1234 + i)
    rounds = 0
    for rnd in (1, 2):                      # the rewiring accumulates from one training forward to the next
      block.set_x0(x)
      try:
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
          z_train = block(x)
      except RuntimeError as err:           # quantile level 1 / (pc_change - rw_addD) outside [0, 1]: the reference stops
        assert 'quantile' in str(err)
        break
      rounds = rnd
      rec['z_train%d' % rnd] = z_train
      rec['train_edge_index%d' % rnd] = block.odefunc.edge_index
      rec['train_weights%d' % rnd] = block.odefunc.edge_weight
    rec['rounds'] = np.int64(rounds)
    save('rewire_' + name, opt, rec, block)
    print('    edges: %d -> %s (%d training forwards before the quantile level leaves [0, 1])' % (
      ei.shape[1] + n, ' -> '.join(str(rec['train_edge_index%d' % r].shape[1]) for r in range(1, rounds + 1)), rounds))


def gen_early():
  """Test-time integrators with early stopping (reference src/early_stop_solver.py), installed on a block the way
  GNNEarly does (src/GNN_early.py:28-36): best (train, val, test, time) and the accuracies after every step."""
  import early_stop_solver as es
  n, d, classes = 150, 24, 5
  ei = make_graph(n, 6, 41)
  g = torch.Generator().manual_seed(42)
  x = torch.randn(n, d, generator=g)
  y = torch.randint(0, classes, (n,), generator=g)
  role = torch.randperm(n, generator=g)
  train_mask, val_mask, test_mask = role < 40, (role >= 40) & (role < 95), role >= 95
  cases = {
    'rk4_transformer': dict(block='constant', function='transformer', method='rk4', time=2.3),
    'rk4_laplacian_arxiv': dict(block='attention', function='laplacian', method='rk4', time=1.5, step_size=0.5,
                                dataset='ogbn-arxiv'),
    'dopri5_laplacian': dict(block='attention', function='laplacian', method='dopri5', time=2.0, tol_scale=800.0),
    'dopri5_transformer_cut': dict(block='constant', function='transformer', method='dopri5', time=1.5,
                                   tol_scale=50.0, max_test_steps=6),
  }
  for i, (name, over) in enumerate(cases.items()):
    opt = {**BASE, 'earlystopxT': 3, 'max_test_steps': 100, **over}
    fcls = {'laplacian': LaplacianODEFunc, 'transformer': ODEFuncTransformerAtt}[opt['function']]
    bcls = {'constant': ConstantODEblock, 'attention': AttODEblock}[opt['block']]
    arxiv = opt['dataset'] == 'ogbn-arxiv'
    data = Data(x=x, edge_index=ei, y=y.view(-1, 1) if arxiv else y, train_mask=train_mask, val_mask=val_mask,
                test_mask=test_mask)
    block = bcls(fcls, [], opt, data, torch.device('cpu'), t=torch.tensor([0, opt['time']]))
    randomise(block, 600 + i)
    gi = torch.Generator().manual_seed(650 + i)
    m2_w = torch.randn(classes, d, generator=gi) * 0.8
    m2_b = torch.randn(classes, generator=gi) * 0.1
    integ = es.EarlyStopInt(opt['time'], opt, torch.device('cpu'))
    integ.data, integ.m2_weight, integ.m2_bias = data, m2_w, m2_b
    block.test_integrator = integ
    log = []
    cls = es.SOLVERS[opt['method']]
    orig = cls.evaluate

    def recorder(self, *a, _orig=orig, _log=log, _m=opt['method']):
      r = _orig(self, *a)
      t1 = float(self.rk_state.t1) if _m == 'dopri5' else float(a[2])
      _log.append([t1] + [float(v) for v in r])
      return r

    cls.evaluate = recorder
    try:
      block.eval()
      block.set_x0(x)
      with torch.no_grad():
        z = block(x)
    finally:
      cls.evaluate = orig
    sol = integ.solver
    best = np.array([sol.best_train, sol.best_val, sol.best_test, sol.best_time], dtype=np.float64)
    save('early_' + name, opt, {'edge_index': ei, 'x': x, 'z': z, 'labels': y, 'train_mask': train_mask,
                                'val_mask': val_mask, 'test_mask': test_mask, 'm2_weight': m2_w, 'm2_bias': m2_b,
                                'best': best, 'steps': np.array(log, dtype=np.float64),
                                'nfe': np.int64(block.odefunc.nfe)}, block)
    print('    best (train, val, test, time) =', best.round(4).tolist(), ' evaluations:', len(log))


def gen_adjoint():
  """Training-mode block forward + backward with opt['adjoint'] (torchdiffeq.odeint_adjoint, restated in
  oracle/shims): output, gradient of <z, c> with respect to the input and to every parameter that receives one."""
  n, d = 120, 24
  ei = make_graph(n, 6, 51)
  g = torch.Generator().manual_seed(52)
  x = torch.randn(n, d, generator=g)
  c = torch.randn(n, d, generator=g)
  cases = {
    'constant_transformer_rk4_rk4': dict(block='constant', function='transformer', method='rk4', time=2.3,
                                         adjoint_method='rk4', adjoint_step_size=1.0),
    'attention_laplacian_dopri5_rk4': dict(block='attention', function='laplacian', method='dopri5', time=2.0, tol_scale=800.0,
                                           adjoint_method='rk4', adjoint_step_size=0.5),
    'constant_laplacian_euler_heun': dict(block='constant', function='laplacian', method='euler', time=2.0, step_size=0.5,
                                          adjoint_method='adaptive_heun', tol_scale_adjoint=3000.0),
    'constant_gat_dopri5_dopri5': dict(block='constant', function='GAT', method='dopri5', time=1.5, tol_scale=20.0,
                                       adjoint_method='dopri5', tol_scale_adjoint=20.0),
    # round 6: the ODE blocks of best_params Pubmed (attention block, cosine scores, squareplus over rows, one head; dopri5 forward,
    # adjoint_method adaptive_heun -- the reference's default) and CoauthorCS (4 heads, squareplus over columns, no source term, no
    # self-loop weight; dopri5 both ways) in miniature, with their tolerances
    'attention_laplacian_dopri5_heun_pubmed': dict(block='attention', function='laplacian', method='dopri5', time=4.0, tol_scale=1991.0688305523001,
                                                   adjoint_method='adaptive_heun', tol_scale_adjoint=16324.368093998313, heads=1, attention_dim=16,
                                                   attention_type='cosine_sim', square_plus=True, attention_norm_idx=0, add_source=True),
    'attention_laplacian_dopri5_dopri5_coauthorcs': dict(block='attention', function='laplacian', method='dopri5', time=3.126400580172773,
                                                         tol_scale=9348.983916372074, adjoint_method='dopri5', tol_scale_adjoint=6599.1250595331385,
                                                         heads=4, attention_dim=8, attention_type='scaled_dot', square_plus=True, attention_norm_idx=1,
                                                         add_source=False, self_loop_weight=0),
    'constant_gat_rk4_rk4': dict(block='constant', function='GAT', method='rk4', time=2.3, adjoint_method='rk4', adjoint_step_size=1.0),
  }
  for i, (name, over) in enumerate(cases.items()):
    opt = {**BASE, 'adjoint': True, 'adjoint_step_size': 1.0, 'tol_scale_adjoint': 1.0, **over}
    fcls = {'laplacian': LaplacianODEFunc, 'transformer': ODEFuncTransformerAtt, 'GAT': ODEFuncAtt}[opt['function']]
    bcls = {'constant': ConstantODEblock, 'attention': AttODEblock}[opt['block']]
    block = bcls(fcls, [], opt, data_of(ei, x), torch.device('cpu'), t=torch.tensor([0, opt['time']]))
    randomise(block, 700 + i)
    block.train()
    xin = x.clone().requires_grad_(True)
    block.set_x0(xin)
    z = block(xin)
    nfe_fwd = block.odefunc.nfe
    (z * c).sum().backward()
    rec = {'edge_index': ei, 'x': x, 'c': c, 'z': z, 'grad_x': xin.grad, 'nfe_forward': np.int64(nfe_fwd),
           'nfe': np.int64(block.odefunc.nfe)}
    for k, p in block.named_parameters():
      if p.grad is not None:
        rec['grad/' + k] = p.grad
    save('adjoint_' + name, opt, rec, block)
    print('    nfe forward %d, total %d; grads: %s' % (nfe_fwd, block.odefunc.nfe,
                                                       ', '.join(k[5:] for k in rec if k.startswith('grad/'))))


def gen_regularised():
  """Training forward with the reference's regularisers (src/regularized_ODE_function.py, registry src/base_classes.py:10-29):
  state + one integral per regulariser, and the gradient of  <z, c> + sum_j coeff_j mean(reg_j)  (run_GNN.py:82-87)."""
  from base_classes import create_regularization_fns
  n, d = 90, 12
  ei = make_graph(n, 6, 61)
  g = torch.Generator().manual_seed(62)
  x = torch.randn(n, d, generator=g)
  c = torch.randn(n, d, generator=g)
  cases = {
    'transformer_rk4_kinetic': dict(function='transformer', method='rk4', time=2.0, kinetic_energy=0.05),
    'laplacian_rk4_all': dict(function='laplacian', method='rk4', time=2.0, kinetic_energy=0.05, jacobian_norm2=0.02,
                              directional_penalty=0.03),
    'transformer_euler_directional': dict(function='transformer', method='euler', time=1.5, step_size=0.5,
                                          directional_penalty=0.1, attention_norm_idx=1, square_plus=True),
    'gat_rk4_kinetic_jacobian': dict(function='GAT', method='rk4', time=1.0, step_size=0.5, kinetic_energy=0.05,
                                     jacobian_norm2=0.02),
  }
  for i, (name, over) in enumerate(cases.items()):
    opt = {**BASE, 'hidden_dim': d, 'attention_dim': 8, 'heads': 2, 'kinetic_energy': None, 'jacobian_norm2': None,
           'total_deriv': None, 'directional_penalty': None, **over}
    fns, coeffs = create_regularization_fns(opt)
    fcls = {'laplacian': LaplacianODEFunc, 'transformer': ODEFuncTransformerAtt, 'GAT': ODEFuncAtt}[opt['function']]
    block = ConstantODEblock(fcls, fns, opt, data_of(ei, x), torch.device('cpu'), t=torch.tensor([0, opt['time']]))
    randomise(block, 800 + i)
    block.train()
    xin = x.clone().requires_grad_(True)
    block.set_x0(xin)
    z, regs = block(xin)
    loss = (z * c).sum() + sum(cf * r.mean() for cf, r in zip(coeffs, regs))
    loss.backward()
    # (the regularised solve integrates reg_odefunc.odefunc -- the block's FIRST function object, base_classes.py:39-42)
    rec = {'edge_index': ei, 'x': x, 'c': c, 'z': z, 'grad_x': xin.grad, 'coeffs': np.asarray(coeffs, dtype=np.float64),
           'nfe': np.int64(block.reg_odefunc.odefunc.nfe)}
    for j, r in enumerate(regs):
      rec['reg%d' % j] = r
    for k, p in block.named_parameters():
      if p.grad is not None:
        rec['grad/' + k] = p.grad
    save('reg_' + name, opt, rec, block)
    print('    regs %s nfe %d' % ([float(r.mean()) for r in regs], block.reg_odefunc.odefunc.nfe))


def gen_training():
  """Training-mode block forward + `loss.backward()` WITHOUT the adjoint method (opt['adjoint'] = False, the reference's default and
  its Cora / Citeseer best_params): torch autograd runs back through every accepted step of the restated torchdiffeq dopri5
  (oracle/shims: rk_common's _UncheckedAssign, the controller under no_grad) or through the fixed grid.  Records the output, the
  gradient of <z, c> with respect to the input and to every parameter that receives one, and the evaluation count."""
  n, d = 140, 24
  ei = make_graph(n, 6, 71)
  g = torch.Generator().manual_seed(72)
  x = torch.randn(n, d, generator=g)
  c = torch.randn(n, d, generator=g)
  cases = {
    # best_params Cora in miniature: attention block, squareplus over columns, 8 heads
    'attention_laplacian_dopri5_cora': dict(block='attention', function='laplacian', method='dopri5', time=4.0, tol_scale=800.0,
                                            square_plus=True, attention_norm_idx=1, heads=8, attention_dim=32),
    'attention_laplacian_dopri5_softmax_rows': dict(block='attention', function='laplacian', method='dopri5', time=2.5, tol_scale=50.0,
                                                    add_source=False),
    'constant_laplacian_dopri5': dict(block='constant', function='laplacian', method='dopri5', time=3.0, tol_scale=200.0),
    'constant_laplacian_midpoint': dict(block='constant', function='laplacian', method='midpoint', time=2.3, step_size=0.5),
    'constant_transformer_midpoint': dict(block='constant', function='transformer', method='midpoint', time=2.0),
    # round 6: the fixed-grid methods of run_GNN.py's default mode (adjoint off) -- rk4 with a short last step, euler, the normaliser of
    # best_params Cora, and the attention block (gradients reach the attention layer through the edge weights of every evaluation)
    'constant_transformer_rk4': dict(block='constant', function='transformer', method='rk4', time=2.3),
    'constant_transformer_euler': dict(block='constant', function='transformer', method='euler', time=2.0, step_size=0.5),
    'constant_transformer_rk4_columns_squareplus': dict(block='constant', function='transformer', method='rk4', time=2.0,
                                                        square_plus=True, attention_norm_idx=1, heads=8, attention_dim=32),
    'constant_laplacian_rk4': dict(block='constant', function='laplacian', method='rk4', time=2.3),
    'attention_laplacian_rk4': dict(block='attention', function='laplacian', method='rk4', time=2.3, step_size=0.5),
    'attention_laplacian_euler_no_source': dict(block='attention', function='laplacian', method='euler', time=3.0, add_source=False),
    'constant_gat_rk4': dict(block='constant', function='GAT', method='rk4', time=2.3),
  }
  for i, (name, over) in enumerate(cases.items()):
    opt = {**BASE, **over}
    fcls = {'laplacian': LaplacianODEFunc, 'transformer': ODEFuncTransformerAtt, 'GAT': ODEFuncAtt}[opt['function']]
    bcls = {'constant': ConstantODEblock, 'attention': AttODEblock}[opt['block']]
    block = bcls(fcls, [], opt, data_of(ei, x), torch.device('cpu'), t=torch.tensor([0, opt['time']]))
    randomise(block, 900 + i)
    block.eval()
    block.set_x0(x)
    with torch.no_grad():
      z_eval = block(x)
    block.odefunc.nfe = 0
    block.train()
    xin = x.clone().requires_grad_(True)
    block.set_x0(xin)
    z = block(xin)
    nfe = block.odefunc.nfe
    (z * c).sum().backward()
    rec = {'edge_index': ei, 'x': x, 'c': c, 'z': z, 'z_eval': z_eval, 'grad_x': xin.grad, 'nfe': np.int64(nfe),
           'nfe_after_backward': np.int64(block.odefunc.nfe)}
    for k, p in block.named_parameters():
      if p.grad is not None:
        rec['grad/' + k] = p.grad
    save('train_' + name, opt, rec, block)
    print('    nfe forward %d (after backward %d); grads: %s' % (nfe, block.odefunc.nfe, ', '.join(k[5:] for k in rec if k.startswith('grad/'))))


if __name__ == '__main__':
  if len(sys.argv) > 1:          # regenerate selected groups only: python oracle/gen_golden.py regularised
    for name in sys.argv[1:]:
      globals()['gen_' + name]()
    sys.exit(0)
  os.makedirs(OUT, exist_ok=True)
  torch.manual_seed(0)
  gen_norms()
  gen_funcs()
  gen_blocks()
  gen_gnn()
  gen_gnn_options()
  gen_beltrami()
  gen_rewire()
  gen_early()
  gen_adjoint()
  gen_regularised()
  gen_training()
