# diagnostic: trial-step traces of the adaptive adjoint backward, native stages vs flat host loop (Pubmed-shaped block)
import importlib, sys, time, torch
sys.path.insert(0, '.')
import gnpde_amd as G
from tests.helpers import Data, random_graph
O = importlib.import_module('gnpde_amd.odeint')
dev = torch.device('cuda:0')
n, d = 19717, 128
ei = random_graph(n, 4, seed=5).to(dev)
x = (torch.randn(n, d, generator=torch.Generator().manual_seed(6)) * 0.5).to(dev)
base = dict(heads=1, attention_dim=16, attention_type='cosine_sim', attention_norm_idx=0, square_plus=True, reweight_attention=False, beltrami=False,
            leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=5000, add_source=True, no_alpha_sigmoid=False, mix_features=False, hidden_dim=d,
            augment=False, adjoint=True, adjoint_method='adaptive_heun', adjoint_step_size=1, tol_scale=1991.07, tol_scale_adjoint=16324.37,
            data_norm='rw', method='dopri5', step_size=1, max_iters=100, block='attention', function='laplacian', time=12.94)
traces = {}
_orig_mixed = O._mixed_norm


def recording_mixed(shapes):
  sizes = [int(torch.Size(sh).numel()) for sh in shapes]

  def norm(v):
    parts = [O._rms(p) for p in torch.split(v, sizes)]
    if O._TRIAL_TRACE is not None:
      O._TRIAL_TRACE.append(('parts', [float(p) for p in parts]))
    return max(parts)
  return norm


O._mixed_norm = recording_mixed
for host in (False, True):
  opt = dict(base, gnpde_host_adjoint=host)
  block = G.AttODEblock(G.LaplacianODEFunc, [], opt, Data(x, ei), dev, t=torch.tensor([0, opt['time']])).to(dev)
  g = torch.Generator().manual_seed(1)
  with torch.no_grad():
    for p in block.parameters():
      if p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
  block.train()
  xin = x.clone().requires_grad_(True)
  block.set_x0(xin)
  z = block(xin)
  O._TRIAL_TRACE = []
  z.sum().backward()
  traces[host] = O._TRIAL_TRACE
  O._TRIAL_TRACE = None
  print('host' if host else 'native', 'trials', len(traces[host]), 'grad alpha', float(block.odefunc.alpha_train.grad), 'beta', float(block.odefunc.beta_train.grad), 'gx norm', float(xin.grad.norm()))
for host in (False, True):
  print('host' if host else 'native')
  shown = 0
  for rec in traces[host]:
    if rec[0] == 'parts':
      print('   parts', ['%.6g' % v for v in rec[1]])
    else:
      print('   trial t %.5f dt %.6g ratio %.6g' % rec)
    shown += 1
    if shown > 14:
      break
