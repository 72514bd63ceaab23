"""Early-stopping test-time integrators: device evaluator (csrc/early_stop.hip) and the EarlyStopInt surface,
against the reference's own runs (tests/golden/early_*.npz) and against the CPU oracle."""
import numpy as np
import pytest
import torch

import gnpde_amd as G
from gnpde_amd import ops
from helpers import Fixture, fixtures, Data, assert_parity
from oracle import restate as R

pytestmark = pytest.mark.gpu

FUNCS = {'laplacian': G.LaplacianODEFunc, 'transformer': G.ODEFuncTransformerAtt}
BLOCKS = {'constant': G.ConstantODEblock, 'attention': G.AttODEblock}


def _hit_bounds(y, w, b, labels, masks, d_dec):
  """[lo, hi] hit counts per mask from float64 logits; nodes whose two best logits are closer than 1e-4 relative
  may legitimately go either way under a different fp32 summation order."""
  z = torch.relu(y[:, :d_dec].double().cpu())
  logits = z @ w.double().cpu().t()
  if b is not None:
    logits = logits + b.double().cpu()
  top = logits.topk(min(2, logits.shape[1]), dim=1)
  pred = top.indices[:, 0]
  if logits.shape[1] > 1:
    amb = (top.values[:, 0] - top.values[:, 1]) <= 1e-4 * logits.abs().max(dim=1).values.clamp_min(1e-30)
  else:
    amb = torch.zeros_like(pred, dtype=torch.bool)
  ok = pred.eq(labels.cpu().long())
  out = []
  for m in masks:
    m = m.cpu()
    sure = int((ok & m & ~amb).sum())
    out.append((sure, sure + int((m & amb).sum())))
  return out


@pytest.mark.parametrize('n,d,d_dec,c,ld', [(1, 16, 16, 3, 16), (15, 24, 24, 7, 24), (16, 24, 24, 16, 24), (1000, 128, 128, 17, 128),
                                            (5003, 128, 128, 40, 128), (777, 162, 162, 40, 162), (2048, 256, 128, 64, 256),
                                            (300, 80, 80, 7, 96), (64, 6, 5, 1, 6)])
@pytest.mark.parametrize('bias', [True, False])
def test_decode_count(dev, n, d, d_dec, c, ld, bias):
  g = torch.Generator().manual_seed(n + 7 * c)
  buf = torch.randn(n, ld, generator=g).to(dev)
  y = buf[:, :d]
  w = (torch.randn(c, d_dec, generator=g) / np.sqrt(d_dec)).to(dev)
  b = (torch.randn(c, generator=g) * 0.1).to(dev) if bias else None
  labels = torch.randint(0, c, (n,), generator=g)
  # make about half of the labels right so that the counts are not trivially small
  with torch.no_grad():
    ref_pred = (torch.relu(y[:, :d_dec]) @ w.t() + (b if bias else 0)).argmax(1).cpu()
  flip = torch.rand(n, generator=g) < 0.5
  labels = torch.where(flip, ref_pred, labels)
  role = torch.randint(0, 4, (n,), generator=g)
  masks = [role == 0, role == 1, role >= 2]    # role 3: also in test (overlap is allowed), some nodes in no split
  masks[2] = masks[2] | (role == 1) & (torch.arange(n) % 5 == 0)
  ev = ops.EarlyStopEvaluator(w, b, labels.to(dev), masks[0].to(dev), masks[1].to(dev), masks[2].to(dev), max_trace=4)
  ev.reset()
  ev.evaluate(y, 3)
  ev.evaluate(y, 4)          # same state again: strict '>' keeps the first
  res = ev.read()
  bounds = _hit_bounds(y, w, b, labels, masks, d_dec)
  assert res['evals'] == 2 and len(res['trace']) == 2
  for k in range(3):
    lo, hi = bounds[k]
    assert lo <= res['trace'][0]['hits'][k] <= hi, (k, res['trace'][0]['hits'], bounds)
  assert res['trace'][0]['hits'] == res['trace'][1]['hits'], 'evaluation is not deterministic'
  assert res['trace'][0]['step'] == 3 and res['trace'][1]['step'] == 4
  if res['trace'][0]['hits'][1] > 0:
    assert res['step'] == 3 and res['best_hits'] == res['trace'][0]['hits']
  else:
    assert res['step'] == 0 and res['best_hits'] == [0, 0, 0]
  assert ev.sizes == [int(m.sum()) for m in masks]


def test_decode_first_maximum_wins(dev):
  """torch.max semantics on exact ties: the lowest class index (classes 2 and 19 share a decoder row)."""
  g = torch.Generator().manual_seed(5)
  n, d, c = 500, 32, 24
  y = torch.randn(n, d, generator=g).to(dev)
  w = torch.randn(c, d, generator=g)
  w[19] = w[2]
  w[[2, 19]] *= 50.0           # the duplicated pair dominates wherever its logit is positive
  b = torch.zeros(c)
  logits = torch.relu(y.cpu()) @ w.t()
  tie_wins = logits[:, 2] >= logits.max(1).values
  assert tie_wins.sum() > 50
  labels = torch.full((n,), 2, dtype=torch.long)
  mask = tie_wins
  ev = ops.EarlyStopEvaluator(w.to(dev), b.to(dev), labels.to(dev), mask.to(dev), mask.to(dev), mask.to(dev))
  ev.reset()
  ev.evaluate(y, 1)
  assert ev.read()['best_hits'] == [int(mask.sum())] * 3


def test_best_keeps_strict_improvements_only(dev):
  """Sequence of states with validation hits 3, 5, 5, 4, 6 -> best moves at evaluations 1, 2 and 5 only."""
  n, c = 64, 4
  w = torch.eye(c, 8).to(dev)
  labels = torch.zeros(n, dtype=torch.long)
  val = torch.zeros(n, dtype=torch.bool)
  val[:10] = True
  ev = ops.EarlyStopEvaluator(w, None, labels.to(dev), val.to(dev), val.to(dev), ~val.to(dev), max_trace=8)
  ev.reset()
  seen = []
  for step, hits in enumerate([3, 5, 5, 4, 6], start=1):
    y = torch.zeros(n, 8)
    y[:, 1] = 1.0            # everybody predicts class 1 (wrong) ...
    y[:hits, 0] = 2.0        # ... except the first `hits` validation nodes
    ev.evaluate(y.to(dev), step)
    seen.append(ev.read()['step'])
  assert seen == [1, 2, 2, 2, 5]
  assert [r['hits'][1] for r in ev.read()['trace']] == [3, 5, 5, 4, 6]


def test_decoder_shape_errors(dev):
  w = torch.randn(65, 8).to(dev)
  lab = torch.zeros(4, dtype=torch.long).to(dev)
  m = torch.ones(4, dtype=torch.bool).to(dev)
  ev = ops.EarlyStopEvaluator(w, None, lab, m, m, m)
  with pytest.raises(G.GnpdeError):
    ev.evaluate(torch.zeros(4, 8, device=dev), 1)        # 65 classes
  ev = ops.EarlyStopEvaluator(torch.randn(3, 16).to(dev), None, lab, m, m, m)
  with pytest.raises(G.GnpdeError):
    ev.evaluate(torch.zeros(4, 8, device=dev), 1)        # decoder wider than the state
  with pytest.raises(G.GnpdeError):
    ev.evaluate(torch.zeros(5, 16, device=dev), 1)       # row count


def _install(fx, dev, keep_trace=True):
  x = fx.t('x', dev)
  data = Data(x, fx.t('edge_index', dev))
  data.y = fx.t('labels', dev).view(-1, 1) if fx.opt['dataset'] == 'ogbn-arxiv' else fx.t('labels', dev)
  for k in ('train_mask', 'val_mask', 'test_mask'):
    setattr(data, k, fx.t(k, dev).bool())
  block = BLOCKS[fx.opt['block']](FUNCS[fx.opt['function']], [], fx.opt, data, dev,
                                  t=torch.tensor([0, fx.opt['time']])).to(dev)
  block.load_state_dict(fx.params, strict=True)
  integ = G.EarlyStopInt(fx.opt['time'], fx.opt, dev)     # as GNNEarly.__init__ does (reference GNN_early.py:28-36)
  integ.keep_trace = keep_trace
  integ.data = data
  integ.m2_weight = fx.t('m2_weight', dev)
  integ.m2_bias = fx.t('m2_bias', dev)
  block.test_integrator = integ
  block.eval()
  return block, integ, x


def _accepted(steps):
  """The reference logs one evaluation per trial step; a rejected trial repeats the previous line."""
  out = []
  for row in steps:
    if not out or row[0] != out[-1][0]:
      out.append(row)
  return np.array(out)


@pytest.mark.parametrize('name', fixtures('early_'))
def test_early_stop_block(dev, name):
  fx = Fixture(name)
  block, integ, x = _install(fx, dev)
  block.set_x0(x)
  with torch.no_grad():
    z = block(x)
  adaptive = fx.opt['method'] == 'dopri5'
  cut = name.endswith('_cut')
  # Fixed end time: the state is insensitive to the step sequence.  `_cut` stops after max_test_steps trial steps,
  # i.e. at an ADAPTIVE time; its first step (dt = 0.04) has an error estimate at fp32 rounding level, so the second
  # step size -- and every later time -- legitimately differs by ~2 % between two correct implementations, and the
  # state by |f| * |dt difference|.  There the check is the step count (nfe), the times within 5 % and the state
  # within 5e-3; accuracies may move by a node or two.
  t_rtol = 1e-6 if not adaptive else (5e-2 if cut else 1e-4)
  tol = 1e-5 if not adaptive else (5e-3 if cut else max(1e-5, 20 * fx.opt['tol_scale'] * 1e-7))
  assert_parity(z, fx.t('z'), tol=tol, what=name)
  sol = integ.solver
  ref_steps = _accepted(fx.arr['steps'])
  got = np.array([[r['time']] + r['acc'] for r in sol.trace])
  assert got.shape == ref_steps.shape, (got.shape, ref_steps.shape)
  assert np.allclose(got[:, 0], ref_steps[:, 0], rtol=t_rtol, atol=0), (got[:, 0], ref_steps[:, 0])
  a_tol = 2.0 / 40 + 1e-9 if cut else 1e-9
  assert np.allclose(got[:, 1:], ref_steps[:, 1:], rtol=0, atol=a_tol), (got, ref_steps)
  best = fx.arr['best']
  assert np.allclose([sol.best_train, sol.best_val, sol.best_test], best[:3], rtol=0, atol=a_tol)
  if not cut:
    assert np.isclose(sol.best_time, best[3], rtol=t_rtol)
  assert block.odefunc.nfe == int(fx.arr['nfe']), 'nfe %d vs reference %d' % (block.odefunc.nfe, int(fx.arr['nfe']))
  # second forward with a changed decoder: same captured graph, new weights picked up
  if not adaptive:
    integ.m2_weight = -fx.t('m2_weight', dev)
    block.odefunc.nfe = 0
    block.set_x0(x)
    with torch.no_grad():
      z2 = block(x)
    assert torch.equal(z, z2)
    masks = [fx.t(k).bool() for k in ('train_mask', 'val_mask', 'test_mask')]
    acc = R.early_stop_accuracies(z2.cpu(), -fx.t('m2_weight'), fx.t('m2_bias'), fx.t('labels'), masks)
    assert np.allclose(integ.solver.trace[-1]['acc'], acc, atol=1e-9)


def test_early_stop_vs_oracle_larger(dev):
  """A graph with hub rows and 40 classes (ogbn-arxiv's count), GRAND-l rk4, 12 steps, against the CPU oracle."""
  from helpers import random_graph
  n, d, c = 3000, 64, 40
  ei = random_graph(n, 8, 3, hubs=2, hub_deg=1500)
  g = torch.Generator().manual_seed(9)
  x = torch.randn(n, d, generator=g)
  labels = torch.randint(0, c, (n,), generator=g)
  role = torch.randperm(n, generator=g)
  masks = [role < 600, (role >= 600) & (role < 1500), role >= 1500]
  fx = Fixture('early_rk4_laplacian_arxiv')
  opt = dict(fx.opt, block='constant', function='laplacian', time=4.0, step_size=1.0, hidden_dim=d, dataset='Cora')
  data = Data(x.to(dev), ei.to(dev))
  data.y = labels.to(dev)
  data.train_mask, data.val_mask, data.test_mask = [m.to(dev) for m in masks]
  block = G.ConstantODEblock(G.LaplacianODEFunc, [], opt, data, dev, t=torch.tensor([0, opt['time']])).to(dev)
  with torch.no_grad():
    block.odefunc.alpha_train.fill_(0.3)
    block.odefunc.beta_train.fill_(0.2)
  w = torch.randn(c, d, generator=g)
  b = torch.randn(c, generator=g) * 0.1
  integ = G.EarlyStopInt(opt['time'], opt, dev)
  integ.keep_trace = True
  integ.data, integ.m2_weight, integ.m2_bias = data, w.to(dev), b.to(dev)
  block.test_integrator = integ
  block.eval()
  block.set_x0(x.to(dev))
  with torch.no_grad():
    z = block(x.to(dev))
  e_n, w_n = R.get_rw_adj(ei, None, 1, opt['self_loop_weight'], n)
  rhs = lambda t, y: R.rhs_laplacian(y, e_n, w_n, torch.tensor(0.3), torch.tensor(0.2), x, opt['no_alpha_sigmoid'],  # noqa: E731
                                     opt['add_source'])
  z_ref, best, steps = R.odeint_rk4_early_stop(rhs, x, opt['earlystopxT'] * opt['time'], 1.0, w, b, labels, masks)
  assert_parity(z, z_ref, what='state')
  got = np.array([[r['time']] + r['acc'] for r in integ.solver.trace])
  ref = np.array(steps)
  assert got.shape == ref.shape
  # one node flipping on a near-tie moves an accuracy by 1 / |mask|
  assert np.all(np.abs(got[:, 1:] - ref[:, 1:]) <= 2.0 / 600 + 1e-12), np.abs(got - ref).max()
  assert abs(integ.solver.best_val - best[1]) <= 2.0 / 900 + 1e-12


@pytest.mark.parametrize('function', ['laplacian', 'transformer'])
def test_early_stop_on_the_relabelled_graph(dev, function):
  """rk4 with the in-graph evaluator on the relabelled graph (graph.LocalityView; EarlyStopEvaluator.relabelled permutes labels
  and split masks, counters and trace are shared): state bit-identical, every per-step hit count and the best step equal."""
  from gnpde_amd import synthetic
  n, d, c = 5000, 64, 40
  ei = torch.as_tensor(synthetic.community_powerlaw_graph(n, 30000, seed=6, n_comm=10)[0])
  g = torch.Generator().manual_seed(9)
  x = torch.randn(n, d, generator=g)
  labels = torch.randint(0, c, (n,), generator=g)
  role = torch.randperm(n, generator=g)
  masks = [role < 1000, (role >= 1000) & (role < 2500), role >= 2500]
  fx = Fixture('early_rk4_laplacian_arxiv')
  opt = dict(fx.opt, block='constant', function=function, time=4.0, step_size=1.0, hidden_dim=d, dataset='Cora',
             heads=4, attention_dim=16, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False, reweight_attention=False,
             beltrami=False, mix_features=False)
  data = Data(x.to(dev), ei.to(dev))
  data.y = labels.to(dev)
  data.train_mask, data.val_mask, data.test_mask = [m.to(dev) for m in masks]
  fcls = G.LaplacianODEFunc if function == 'laplacian' else G.ODEFuncTransformerAtt
  block = G.ConstantODEblock(fcls, [], opt, data, dev, t=torch.tensor([0, opt['time']])).to(dev)
  with torch.no_grad():
    block.odefunc.alpha_train.fill_(0.3)
    block.odefunc.beta_train.fill_(0.2)
  integ = G.EarlyStopInt(opt['time'], opt, dev)
  integ.keep_trace = True
  integ.data, integ.m2_weight, integ.m2_bias = data, torch.randn(c, d, generator=g).to(dev), (torch.randn(c, generator=g) * 0.1).to(dev)
  block.test_integrator = integ
  block.eval()
  block.set_x0(x.to(dev))
  res = {}
  for mode in ('0', 'parts', 'degree'):
    block.odefunc.opt['gnpde_reorder'] = mode
    with torch.no_grad():
      z = block(x.to(dev)).clone()
    sol = integ.solver
    res[mode] = (z, [(r['time'], tuple(r['hits'])) for r in sol.trace], (sol.best_val, sol.best_test, sol.best_time))
    assert len(sol.trace) == int(opt['earlystopxT'] * opt['time']) and all(sum(r['hits']) > 0 for r in sol.trace)
  for mode in ('parts', 'degree'):
    assert torch.equal(res['0'][0], res[mode][0])
    assert res['0'][1] == res[mode][1] and res['0'][2] == res[mode][2], (mode, res['0'][1:], res[mode][1:])


def test_euler_is_refused(dev):
  fx = Fixture('early_rk4_transformer')
  fx.opt['method'] = 'euler'
  block, integ, x = _install(fx, dev)
  block.set_x0(x)
  with pytest.raises(AssertionError):
    with torch.no_grad():
      block(x)


def test_index_valued_split_masks(dev):
  """The reference's ogbn-arxiv Data carries NODE-INDEX tensors as masks (train_mask = split_idx['train'], reference
  src/data.py:90); `logits[mask]` indexes either way.  Same counts as the equivalent boolean masks."""
  n, d, c = 3000, 32, 9
  g = torch.Generator().manual_seed(91)
  y = torch.randn(n, d, generator=g).to(dev)
  w = (torch.randn(c, d, generator=g) / np.sqrt(d)).to(dev)
  labels = torch.randint(0, c, (n,), generator=g)
  role = torch.randint(0, 3, (n,), generator=g)
  bools = [role == 0, role == 1, role == 2]
  idx = [torch.nonzero(m).flatten()[torch.randperm(int(m.sum()), generator=g)] for m in bools]   # shuffled index lists
  a = ops.EarlyStopEvaluator(w, None, labels.to(dev), *[m.to(dev) for m in bools])
  b = ops.EarlyStopEvaluator(w, None, labels.to(dev), *[i.to(dev) for i in idx])
  assert a.sizes == b.sizes == [int(m.sum()) for m in bools]
  for ev in (a, b):
    ev.reset()
    ev.evaluate(y, 1)
  assert a.read()['best_hits'] == b.read()['best_hits']
  with pytest.raises(G.GnpdeError):
    ops.EarlyStopEvaluator(w, None, labels.to(dev), torch.tensor([0, n]).to(dev), idx[1].to(dev), idx[2].to(dev))


def test_dopri5_rejections_before_the_first_accept_evaluate_the_initial_state(dev):
  """Reference src/early_stop_solver.py:82-90 evaluates rk_state.y1 after EVERY trial step; after a rejection that is the
  unchanged previous state, so trials rejected before the first accept evaluate y0 at t0 -- and y0 can be the best."""
  import importlib
  O = importlib.import_module('gnpde_amd.odeint')
  n, d, c = 400, 16, 5
  g = torch.Generator().manual_seed(92)
  y0 = torch.randn(n, d, generator=g).to(dev)
  w = (torch.randn(c, d, generator=g) / np.sqrt(d)).to(dev)
  labels = (torch.relu(y0) @ w.t()).argmax(1)                # y0 classifies every node correctly ...
  role = torch.randint(0, 3, (n,), generator=g)
  masks = [(role == i).to(dev) for i in range(3)]
  calls = []

  def stiff(t, y):          # ... and the dynamics scramble it; the rate jumps right after t0, so the first trial step (whose
    calls.append(float(t))  # first stage was evaluated AT t0) has a large error estimate and is rejected
    rate = 1.0 if float(t) < 1e-6 else 500.0
    return -rate * y + 0.8 * rate * torch.roll(y, 1, dims=1)

  class Opt(dict):
    pass
  opt = Opt(method='dopri5', dataset='Cora', max_test_steps=100, earlystopxT=1)
  integ = G.EarlyStopInt(1.0, opt, dev)

  class D(object):
    pass
  data = D()
  data.y, data.train_mask, data.val_mask, data.test_mask = labels, masks[0], masks[1], masks[2]
  integ.data, integ.m2_weight, integ.m2_bias = data, w, None
  rejected = []
  orig = O._solve_dopri5

  def spy(func, y0_, t, rtol, atol, **kw):
    inner = kw.get('on_reject')
    kw['on_reject'] = lambda y, tc: (rejected.append(tc), inner(y, tc))[1]
    return orig(func, y0_, t, rtol, atol, **kw)
  ES = importlib.import_module('gnpde_amd.early_stop_solver')
  ES._solve_dopri5 = spy
  try:
    with torch.no_grad():
      integ(stiff, y0, torch.tensor([0.0, 1.0], device=dev), method='dopri5', rtol=1e-6, atol=1e-8)
  finally:
    ES._solve_dopri5 = orig
  assert rejected and rejected[0] == 0.0, 'the first trial step must have been rejected for this test to bite'
  sol = integ.solver
  assert sol.best_time == 0.0 and sol.best_val == 1.0 and sol.best_train == 1.0, (sol.best_time, sol.best_val)


@pytest.mark.parametrize('name', [n for n in fixtures('early_') if 'dopri5' in n])
def test_early_stop_dopri5_device_controller_equals_host_controller(dev, name):
  """The reference's default evaluation path (dopri5 + early stopping, src/early_stop_solver.py:82-128) on the DEVICE
  controller -- evaluator kernels inside the captured trial step, gated by the controller record, trial budget counted on the
  device -- against the host-controlled loop (options={'eager_stages': True}: one scalar read per trial step): same accepted
  times, same accuracies step by step, same best, same NFE, same state; and the host read the record once per BATCH of trial
  steps, not once per step."""
  fx = Fixture(name)
  res = {}
  for mode in ('device', 'host'):
    block, integ, x = _install(fx, dev)
    if mode == 'host':
      inner = integ

      class _Host(torch.nn.Module):      # the block passes its own options; add the switch on the way through
        def forward(self, func, y0, t, **kw):
          kw['options'] = dict(kw.get('options') or {}, eager_stages=True)
          return inner(func, y0, t, **kw)
      block.test_integrator = _Host()
    block.set_x0(x)
    with torch.no_grad():
      z = block(x)
    sol = integ.solver
    res[mode] = dict(z=z.clone(), trace=[(r['time'], tuple(r['hits']), r['step']) for r in sol.trace],
                     best=(sol.best_train, sol.best_val, sol.best_test, sol.best_time), nfe=block.odefunc.nfe,
                     stats=getattr(block.odefunc, '_dopri5_stats', None))
  a, b = res['device'], res['host']
  assert a['nfe'] == b['nfe'] == int(fx.arr['nfe'])
  assert len(a['trace']) == len(b['trace'])
  for (ta, ha, sa), (tb, hb, sb) in zip(a['trace'], b['trace']):
    assert sa == sb and ha == hb and abs(ta - tb) <= 1e-6 * max(abs(tb), 1e-3), ((ta, ha, sa), (tb, hb, sb))
  assert a['best'][:3] == b['best'][:3] and abs(a['best'][3] - b['best'][3]) <= 1e-6 * max(abs(b['best'][3]), 1e-3)
  assert_parity(a['z'], b['z'], tol=2e-6, what=name + ': device vs host controller')
  st = a['stats']
  trials = st['accepted'] + st['rejected']
  assert st['launches'] == trials, 'no trial step may be replayed past the end of the solve'
  assert st['syncs'] <= trials and (trials < 6 or st['syncs'] < trials), st      # batches, not one read per trial step
