import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gnpde_amd as G
class A: pass
args = A(); args.seed = 0; args.scale = 1.0; args.config = 'arxiv-adjoint'; args.replays = 3
dev = torch.device('cuda:0')
ei_cpu, n = G.synthetic.make_graph('arxiv', seed=0, scale=1.0)
d = 162
ei = ei_cpu.to(dev)
x = (torch.randn(n, d, generator=torch.Generator().manual_seed(12)) * 0.5).to(dev)
opt = dict(heads=2, attention_dim=32, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False, reweight_attention=False, beltrami=False,
           leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=500, add_source=False, no_alpha_sigmoid=False, mix_features=False, hidden_dim=d,
           augment=False, adjoint=True, adjoint_method='rk4', adjoint_step_size=1, tol_scale=11353.558848254957,
           tol_scale_adjoint=1.0, data_norm='rw', method='dopri5', step_size=1, max_iters=100, block='hard_attention',
           function='laplacian', time=3.6760155951687636, att_samp_pct=0.8105268910037231, use_flux=False)
data = bench._Data()
data.x, data.edge_index, data.edge_attr, data.num_nodes = x, ei, None, n
block = G.HardAttODEblock(G.LaplacianODEFunc, [], opt, data, dev, t=torch.tensor([0, opt['time']])).to(dev)
block.train()
def it():
  xin = x.clone().requires_grad_(True)
  block.set_x0(xin)
  block.odefunc.nfe = 0
  torch.cuda.synchronize(); t0 = time.perf_counter()
  z = block(xin)
  torch.cuda.synchronize(); t1 = time.perf_counter()
  z.sum().backward()
  torch.cuda.synchronize(); t2 = time.perf_counter()
  return t1 - t0, t2 - t1
for _ in range(4): print(it())
pr = cProfile.Profile()
pr.enable()
for _ in range(3): it()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
