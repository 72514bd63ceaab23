"""Functions / score types outside the BASELINE configurations at the benchmark shape: rk4 steps/s of ConstantODEblock.forward for the
transformer function with each score type and for the GAT function (ODEFuncAtt), and the kernel profile when run under rocprofv3
(a look at rows a4 / a10 of SURVEY section 8)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnpde_amd as G  # noqa: E402


class D(object):
  pass


def main():
  dev = torch.device('cuda:0')
  K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
  ei, n = G.synthetic.make_graph('arxiv', seed=0)
  d = 128
  x = torch.randn(n, d, generator=torch.Generator().manual_seed(1)).to(dev)
  for function, att_type, fcls in (('transformer', 'scaled_dot', G.ODEFuncTransformerAtt), ('transformer', 'cosine_sim', G.ODEFuncTransformerAtt),
                                   ('transformer', 'pearson', G.ODEFuncTransformerAtt), ('transformer', 'exp_kernel', G.ODEFuncTransformerAtt),
                                   ('GAT', 'scaled_dot', G.ODEFuncAtt)):
    opt = dict(heads=4, attention_dim=16, attention_type=att_type, attention_norm_idx=0, square_plus=False, reweight_attention=False,
               beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=100000, add_source=True, no_alpha_sigmoid=False,
               mix_features=False, hidden_dim=d, augment=False, adjoint=False, tol_scale=1.0, data_norm='rw', method='rk4', step_size=1.0,
               max_iters=100, block='constant', function=function, time=float(K))
    data = D()
    data.x, data.edge_index, data.edge_attr, data.num_nodes = x, ei.to(dev), None, n
    block = fcls and G.ConstantODEblock(fcls, [], opt, data, dev, t=torch.tensor([0, float(K)])).to(dev)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
      for name, p in block.named_parameters():
        if p.dim() >= 2 and 'multihead_att_layer' in name:
          p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      block.odefunc.beta_train.fill_(0.1)
    block.eval()
    times = []
    with torch.no_grad():
      for rep in range(5):
        block.set_x0(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = block(x)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = sorted(times[1:])[len(times[1:]) // 2]
    print('%-12s %-11s %d rk4 steps: %.3f ms per forward, %.1f steps/s (finite %s)' % (function, att_type, K, 1e3 * best, K / best, bool(torch.isfinite(z).all())))


if __name__ == '__main__':
  main()
