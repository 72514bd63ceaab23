"""A/B of the hub-row attention's phase 1 (GPU box only): chunk partials folded straight from memory (gnpde_tune(11, 2)) against the
LDS-staged fold (the default since round 3, csrc/attention.hip::hub_normalise_body).  Same values combined in the same order, so the
head-mean weights must be BIT-identical; prints the time of the attention launches (projection excluded) for both.

  python tools/hub_fold_ab.py [arxiv|rmat|...]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
cfg = G.synthetic.CONFIGS[name]
ei, n = G.synthetic.make_graph(name)
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
graph = G.CSRGraph(ei2.to(dev), n)
d, A, h = cfg['d'], cfg['att_dim'], cfg['heads']
gen = torch.Generator().manual_seed(0)
x = torch.randn(n, d, generator=gen).to(dev)
wqk = (torch.randn(2 * A, d, generator=gen) / d ** 0.5).to(dev)
qk = ops.linear(x, wqk, torch.zeros(2 * A, device=dev))
st = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False, q=qk, k=qk[:, A:], ldqk=2 * A)
L = _lib.lib()


def attend():
  return ops.edge_attention(graph, st, True, False, False, like=x)[0]


def timed(reps=50):
  for _ in range(3):
    attend()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    attend()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


res = {'graph': name, 'hub_rows': graph.n_long_rows, 'hub_chunks': int(graph.struct.n_long_chunks), 'max_row_len': graph.max_row_len,
       'from_memory_us': [], 'lds_staged_us': []}
_lib.check(L.gnpde_tune(_lib.TUNE_HUB_FOLD, 2))
w0 = attend().clone()
_lib.check(L.gnpde_tune(_lib.TUNE_HUB_FOLD, 0))
w1 = attend().clone()
_lib.check(L.gnpde_tune(_lib.TUNE_HUB_FOLD, 0))
res['bit_equal'] = bool(torch.equal(w0, w1))
for _ in range(3):
  _lib.check(L.gnpde_tune(_lib.TUNE_HUB_FOLD, 2))
  res['from_memory_us'].append(round(timed(), 1))
  _lib.check(L.gnpde_tune(_lib.TUNE_HUB_FOLD, 0))
  res['lds_staged_us'].append(round(timed(), 1))
_lib.check(L.gnpde_tune(_lib.TUNE_HUB_FOLD, 0))
print(json.dumps(res))
sys.exit(0 if res['bit_equal'] else 1)
