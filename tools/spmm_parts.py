"""Where does the aggregation's time go on a hub-heavy graph?  Times the hub chunks and the rows separately (GPU box only).
  python tools/spmm_parts.py [rmat|arxiv]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'rmat'
ei, n = G.synthetic.make_graph(name)
d = G.synthetic.CONFIGS[name]['d']
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
graph = G.CSRGraph(ei2.to(dev), n)
deg = torch.bincount(ei2[0], minlength=n)
x = torch.randn(n, d, device=dev)
out, x0 = torch.empty_like(x), torch.randn_like(x)
w = torch.rand(graph.e, device=dev) / 16
alpha, beta = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
def run(reps=5):
  for _ in range(2): ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, out=out)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, out=out)
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps
e_long = int(deg[deg > 512].sum()); e_short = graph.e - e_long
rows_short = int((deg <= 512).sum())
res = {}
for part, label in ((0, 'all'), (1, 'hub chunks only'), (2, 'rows only')):
  _lib.check(_lib.lib().gnpde_tune(9, part))
  res[label] = round(run(), 1)
_lib.check(_lib.lib().gnpde_tune(9, 0))
b_chunks = e_long * (8 + 4 * d)
b_rows = e_short * (8 + 4 * d) + n * (4 + 12 * d)
print(json.dumps({'graph': name, 'us': res, 'entries_in_hub_chunks': e_long, 'entries_in_rows': e_short, 'rows': rows_short,
                  'hub_chunks_gbs': round(b_chunks / res['hub chunks only'] / 1e3), 'rows_gbs': round(b_rows / res['rows only'] / 1e3),
                  'rows_le4': int((deg <= 4).sum()), 'rows_5_64': int(((deg > 4) & (deg <= 64)).sum()),
                  'rows_65_512': int(((deg > 64) & (deg <= 512)).sum())}))
