import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
mode = sys.argv[1]
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
block.eval(); block.set_x0(x)
options = {'eager_stages': True} if mode == 'eager' else {}
def integ(func, y0, t, **kw):
  kw['options'] = dict(kw.get('options') or {}, **options)
  return G.odeint(func, y0, t, **kw)
block.test_integrator = integ
with torch.no_grad():
  for _ in range(6): z = block(x)
torch.cuda.synchronize()
