"""A/B timing of aggregation-kernel variants on the bench graph (GPU box only)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
ei, n = G.synthetic.make_graph(name)
d = G.synthetic.CONFIGS[name]['d']
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
graph = G.CSRGraph(ei2.to(dev), n)
x = torch.randn(n, d, device=dev)
bufs = [torch.randn_like(x) for _ in range(6)]
y, k1, k2, k3, ua, x0 = bufs
w = torch.rand(graph.e, device=dev) / 16
alpha, beta = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
E = graph.e
bytes_alg = E * (8 + 4 * d) + n * (4 + 8 * d) + 4 * d * n
def run(stage_kw, reps=20):
  for _ in range(3):
    ops.spmm_rhs(graph, w, stage_kw.pop('u_') if False else x, alpha, beta, x0, True, **stage_kw)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, **stage_kw)
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps
res = {}
for v in [0, 7, 8, 9, 10, 11, 12, 13, 14, 15, 7, 0]:
  _lib.check(_lib.lib().gnpde_tune(0, v))
  t_rhs = run(dict(stage=_lib.STAGE_RHS, out_k=k1))
  t_rk3 = run(dict(stage=_lib.STAGE_RK3, dt=1.0, y=y, k1=k1, k2=k2, out_k=k3, out_y=ua))
  res[v] = (round(t_rhs, 1), round(t_rk3, 1), round(bytes_alg / t_rhs / 1e3, 0))
  print('variant', v, 'rhs %.1f us  rk3 %.1f us  -> %.0f GB/s algorithmic' % (t_rhs, t_rk3, bytes_alg / t_rhs / 1e3), flush=True)
