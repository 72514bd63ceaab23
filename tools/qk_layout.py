"""Row attention with q||k interleaved in one [n, 2A] array (a 128-byte line holds one node's q AND k: every k-row gather drags
the node's q along) against separate [n, A] arrays (two nodes' k rows per line, the k table half the size).  GPU box only."""
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
graph = G.CSRGraph(ei2.to(dev), n)
d, A, h = cfg['d'], 16, 4
gen = torch.Generator().manual_seed(3)
x = torch.randn(n, d, generator=gen).to(dev)
wqk = (torch.randn(2 * A, d, generator=gen) / d ** 0.5).to(dev)
qk = ops.linear(x, wqk, torch.zeros(2 * A, device=dev))
qs, ks = qk[:, :A].contiguous(), qk[:, A:].contiguous()
st_i = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False, q=qk, k=qk[:, A:], ldqk=2 * A)
st_s = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False, q=qs, k=ks, ldqk=A)
w_i = ops.edge_attention(graph, st_i, True, False, False, like=x)[0]
w_s = ops.edge_attention(graph, st_s, True, False, False, like=x)[0]
print(json.dumps({'equal': bool(torch.equal(w_i, w_s)), 'max_abs_diff': float((w_i - w_s).abs().max())}), flush=True)
for label, st in (('interleaved [n, 2A]', st_i), ('separate [n, A] + [n, A]', st_s), ('interleaved again', st_i)):
  t = bench.timed_replay(lambda: ops.edge_attention(graph, st, True, False, False, like=x), 16)
  print(json.dumps({'layout': label, 'attention_us': round(t * 1e6, 2)}), flush=True)
