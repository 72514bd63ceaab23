"""C4-shaped timing: ogbn-arxiv shape, d = 162, hard_attention block (eval), Laplacian function, dopri5
(tol_scale 11353, T = 3.676): native stage launches vs the host loop of elementwise ops (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
dev = torch.device('cuda:0')
ei, n = G.synthetic.make_graph('arxiv')
d = 162
x = (torch.randn(n, d, generator=torch.Generator().manual_seed(0)) * 0.5).to(dev)
opt = dict(heads=2, attention_dim=32, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False,
           reweight_attention=False, beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9,
           add_source=False, no_alpha_sigmoid=False, mix_features=False, hidden_dim=d, augment=False, adjoint=False,
           tol_scale=11353.558848254957, data_norm='rw', method='dopri5', step_size=1.0, max_iters=100,
           block='hard_attention', function='laplacian', time=3.6760155951687636, att_samp_pct=0.81, use_flux=False)
class D: pass
data = D(); data.x, data.edge_index, data.edge_attr, data.num_nodes = x, ei.to(dev), None, n
block = G.HardAttODEblock(G.LaplacianODEFunc, [], opt, data, dev, t=torch.tensor([0, opt['time']])).to(dev)
g = torch.Generator().manual_seed(1)
with torch.no_grad():
  for name, p in block.named_parameters():
    if p.dim() >= 2 and 'multihead_att_layer' in name:
      p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
block.eval(); block.set_x0(x)
def solve_with(options):
  def integ(func, y0, t, **kw):
    kw['options'] = dict(kw.get('options') or {}, **options)
    return G.odeint(func, y0, t, **kw)
  return integ


modes = [('device controller', {}), ('host controller', {'eager_stages': True}), ('host loop', {'host_controller': True})]
if len(sys.argv) > 1 and sys.argv[1] == 'native':
  modes = modes[:2]
for label, options in modes:
  block.test_integrator = solve_with(options)
  with torch.no_grad():
    block(x); torch.cuda.synchronize()
    block.odefunc.nfe = 0
    t0 = time.perf_counter()
    for _ in range(5): z = block(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
  print('%-18s forward %.2f ms, %d evaluations of f per forward -> %.0f evaluations/s  %s' % (
    label, dt * 1e3, block.odefunc.nfe // 5, block.odefunc.nfe / 5 / dt,
    getattr(block.odefunc, '_dopri5_stats', '') if label == 'device controller' else ''), flush=True)

# the launch-bound end: Cora shape (2 708 nodes, 10 556 edges + self loops), GRAND-nl, d = 80, dopri5 as run_GNN.py's default
ei, n = G.synthetic.make_graph('cora')
d = 80
x = (torch.randn(n, d, generator=torch.Generator().manual_seed(0)) * 0.5).to(dev)
opt = dict(opt, block='constant', function='transformer', hidden_dim=d, heads=8, attention_dim=128, time=18.2948, tol_scale=821.9773,
           add_source=True, square_plus=True, attention_norm_idx=1)
data = D(); data.x, data.edge_index, data.edge_attr, data.num_nodes = x, ei.to(dev), None, n
block = G.ConstantODEblock(G.ODEFuncTransformerAtt, [], opt, data, dev, t=torch.tensor([0, opt['time']])).to(dev)
with torch.no_grad():
  for name, p in block.named_parameters():
    if p.dim() >= 2 and 'multihead_att_layer' in name:
      p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
block.eval(); block.set_x0(x)
for label, options in modes[:2]:
  block.test_integrator = solve_with(options)
  with torch.no_grad():
    block(x); torch.cuda.synchronize()
    block.odefunc.nfe = 0
    t0 = time.perf_counter()
    for _ in range(10): z = block(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
  print('cora %-18s forward %.3f ms, %d evaluations of f per forward -> %.0f evaluations/s  %s' % (
    label, dt * 1e3, block.odefunc.nfe // 10, block.odefunc.nfe / 10 / dt,
    getattr(block.odefunc, '_dopri5_stats', '') if label == 'device controller' else ''), flush=True)
