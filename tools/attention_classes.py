"""Where does the row attention spend its time?  The arxiv-shaped graph restricted to one degree class of rows at a time
(all columns kept: the k-row gathers stay as scattered as in the whole graph).  GPU box only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib
import bench

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
cfg = G.synthetic.CONFIGS[name]
ei, n = G.synthetic.make_graph(name)
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
ei2 = ei2.to(dev)
d, A, h = cfg['d'], 16, 4
gen = torch.Generator().manual_seed(3)
x = torch.randn(n, d, generator=gen).to(dev)
wqk = (torch.randn(2 * A, d, generator=gen) / d ** 0.5).to(dev)
qk = ops.linear(x, wqk, torch.zeros(2 * A, device=dev))
st = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False, q=qk, k=qk[:, A:], ldqk=2 * A)
deg = torch.bincount(ei2[0], minlength=n)
classes = [('all rows', 0, 1 << 30), ('rows of <= 8 entries', 0, 8), ('rows of 9..16', 9, 16), ('rows of <= 16', 0, 16), ('rows of 17..64', 17, 64),
           ('rows of 65..128', 65, 128), ('rows of 129..512', 129, 512), ('rows of 17..512', 17, 512), ('rows of > 512 (hub phases)', 513, 1 << 30)]
for label, lo, hi in classes:
  keep = (deg >= lo) & (deg <= hi)
  m = keep[ei2[0]]
  if int(m.sum()) == 0:
    continue
  g = G.CSRGraph(ei2[:, m], n)
  t = bench.timed_replay(lambda: ops.edge_attention(g, st, True, False, False, like=x), 16)
  print(json.dumps({'class': label, 'rows': int(keep.sum()), 'entries': int(m.sum()), 'attention_us': round(t * 1e6, 2)}), flush=True)
