"""Training-step timing at the ogbn-arxiv shape (GPU box only).

  A  GRAND-nl (constant block, transformer function, scaled dot), d = 128, rk4 step 1, T = 10, adjoint rk4 step 1
  B  the reference's ogbn-arxiv best_params shape: hard_attention block, Laplacian function, d = 162, dopri5
     (tol_scale 11353, T = 3.676), adjoint rk4 step 1, att_samp_pct 0.81

For each: forward (tape-free native solver), backward (adjoint ODE through native f + VJP kernels), evaluations of f.
"""
import sys, os, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G

dev = torch.device('cuda:0')
ei, n = G.synthetic.make_graph('arxiv')
ei = ei.to(dev)


class D:
  pass


def run(label, d, over, reps=3):
  x = (torch.randn(n, d, generator=torch.Generator().manual_seed(0)) * 0.5).to(dev)
  opt = dict(heads=4, attention_dim=16, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False,
             reweight_attention=False, beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9,
             add_source=True, no_alpha_sigmoid=False, mix_features=False, hidden_dim=d, augment=False, adjoint=True,
             adjoint_method='rk4', adjoint_step_size=1.0, tol_scale=1.0, tol_scale_adjoint=1.0, data_norm='rw',
             method='rk4', step_size=1.0, max_iters=100, block='constant', function='transformer', time=10.0,
             att_samp_pct=1.0, use_flux=False)
  opt.update(over)
  data = D()
  data.x, data.edge_index, data.edge_attr, data.num_nodes = x, ei, None, n
  block = G.set_block(opt)(G.set_function(opt), [], opt, data, dev, t=torch.tensor([0, opt['time']])).to(dev)
  g = torch.Generator().manual_seed(1)
  with torch.no_grad():
    for name, p in block.named_parameters():
      if p.dim() >= 2 and 'multihead_att_layer' in name:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
  block.train()
  tf = tb = 0.0
  nf = nb = 0
  for r in range(reps + 1):
    xin = x.clone().requires_grad_(True)
    block.set_x0(xin)
    block.odefunc.nfe = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
      z = block(xin)
    loss = z.pow(2).mean()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n_fwd = block.odefunc.nfe
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if r > 0:          # first round: graph capture, allocator warm-up
      tf += t1 - t0
      tb += t2 - t1
      nf, nb = n_fwd, block.odefunc.nfe - n_fwd
  print('%s: forward %.1f ms (%d evaluations of f), backward %.1f ms (%d evaluations of f + VJP), peak memory %.1f GB'
        % (label, tf / reps * 1e3, nf, tb / reps * 1e3, nb, torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else 'AB'
if 'A' in which:
  run('A  GRAND-nl d=128 rk4 T=10, adjoint rk4', 128, {})
if 'B' in which:
  run('B  hard_attention + GRAND-l d=162 dopri5 T=3.676, adjoint rk4', 162,
      dict(heads=2, attention_dim=32, add_source=False, tol_scale=11353.558848254957, method='dopri5', block='hard_attention',
           function='laplacian', time=3.6760155951687636, att_samp_pct=0.8105268910037231))
