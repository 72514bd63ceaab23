"""Upper bound of what row ordering buys the aggregation kernel: the community-structured graph with
shuffled ids vs ids sorted by (true) community vs sorted by degree (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gnpde_amd as G
from gnpde_amd import ops, _lib
dev = torch.device('cuda:0')
n, d = 169343, 128
ei, relabel, comm = G.synthetic.community_powerlaw_graph(n, 1260000, seed=0)
true = torch.empty(n, dtype=torch.long); true[torch.from_numpy(relabel)] = torch.from_numpy(comm)
deg = torch.bincount(ei[0], minlength=n)
def run(ei_, tag):
  ei2, _ = G.add_remaining_self_loops(ei_, None, 1.0, n)
  graph = G.CSRGraph(ei2.to(dev), n)
  x = torch.randn(n, d, device=dev); x0 = torch.randn_like(x); out = torch.empty_like(x)
  w = torch.rand(graph.e, device=dev) / 16
  alpha, beta = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
  for _ in range(3): ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, out=out)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, out=out)
  e1.record(); torch.cuda.synchronize()
  print('%-28s E=%d  %.1f us' % (tag, graph.e, e0.elapsed_time(e1) * 1e3 / 20), flush=True)
run(ei, 'shuffled ids')
for name, key in (('sorted by community', true * n + torch.arange(n)), ('sorted by degree desc', -deg * n + torch.arange(n)),
                  ('community, then degree', true * (n * 20000) + (-deg + 15000) * n + torch.arange(n))):
  order = torch.argsort(key); newid = torch.empty(n, dtype=torch.long); newid[order] = torch.arange(n)
  run(newid[ei], name)
