"""A/B timing of aggregation-kernel variants (GPU box only).

  python tools/spmm_ab.py GRAPH [variant ...]      GRAPH = arxiv | rmat | rmat:SCALE | uniform:LOG2N:DEG[:D]

Every variant's output is compared with variant 99's (the round-1 row kernel; 0 = the shipped default) (same inputs, same summation order inside a row => equal up to
the order in which the G neighbour slots are combined), then timed over the RHS epilogue and the RK4C stage epilogue.
Prints one JSON line per variant and a final summary line.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib

dev = torch.device('cuda:0')
spec = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
variants = [int(v) for v in sys.argv[2:]] or [99, 50, 100, 101, 102, 103, 110, 111, 112, 113, 120, 121, 122, 123, 0]
t0 = time.time()
parts = spec.split(':')
if parts[0] == 'uniform':
  n = 1 << int(parts[1])
  ei = G.synthetic.uniform_graph(n, n * int(parts[2]) // 2, seed=0)
  d = int(parts[3]) if len(parts) > 3 else 256
elif parts[0] == 'rmat' and len(parts) > 1:
  ei, n = G.synthetic.make_graph('rmat', scale=float(parts[1]))
  d = 256
elif parts[0] in ('arxiv_sorted', 'arxiv_part'):
  # locality experiments on the arxiv shape: ids in community order (upper bound), or ordered by the native partitioner
  import numpy as np
  cfg = G.synthetic.CONFIGS['arxiv']
  n, d = cfg['n'], cfg['d']
  ei, relabel, comm = G.synthetic.community_powerlaw_graph(n, cfg['pairs'], 0)
  if parts[0] == 'arxiv_sorted':
    inv = np.empty(n, dtype=np.int64)
    inv[relabel] = np.arange(n)
    ei = torch.from_numpy(inv)[ei]
  else:
    k = int(parts[1]) if len(parts) > 1 else 64
    tp = time.time()
    g0 = G.CSRGraph(ei, n)
    part = G.partition_rows(g0, k).long()
    order = torch.argsort(part, stable=True)          # new position -> old id
    inv = torch.empty(n, dtype=torch.long)
    inv[order] = torch.arange(n)
    ei = inv[ei]
    cut = float((part[g0.t['rowidx'][:g0.e].long().cpu()] != part[g0.t['colidx'][:g0.e].long().cpu()]).float().mean())
    print(json.dumps({'partition_s': round(time.time() - tp, 2), 'parts': k, 'edge_cut': round(cut, 3)}), flush=True)
else:
  ei, n = G.synthetic.make_graph(parts[0])
  d = G.synthetic.CONFIGS[parts[0]]['d']
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
graph = G.CSRGraph(ei2.to(dev), n)
deg = torch.bincount(ei2[0], minlength=n)
print(json.dumps({'graph': spec, 'n': n, 'e': graph.e, 'd': d, 'long_rows': graph.n_long_rows,
                  'long_chunks': int(graph.struct.n_long_chunks), 'max_deg': int(deg.max()),
                  'median_deg': float(deg.float().median()), 'rows_le4': int((deg <= 4).sum()),
                  'edges_in_long_rows': int(deg[deg > 512].sum()), 'prep_s': round(time.time() - t0, 1)}), flush=True)
del ei, ei2, deg
x = torch.randn(n, d, device=dev)
y, ub, k1, x0 = (torch.randn_like(x) for _ in range(4))
w = torch.rand(graph.e, device=dev) / 16
alpha, beta = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
E = graph.e
bytes_alg = E * (8 + 4 * d) + n * (4 + 8 * d) + 4 * d * n
STAGES = {'rhs': dict(stage=_lib.STAGE_RHS, out_k=k1),
          'rk4c': dict(stage=_lib.STAGE_RK4C, dt=1.0, y=y, k1=ub, out_y=y)}


def run(kw, reps):
  for _ in range(2):
    ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, **kw)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, **kw)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


_lib.check(_lib.lib().gnpde_tune(0, 99))   # reference: the round-1 row kernel
ref = ops.spmm_rhs(graph, w, x, alpha, beta, x0, True).clone()
_lib.check(_lib.lib().gnpde_tune(0, 0))
shipped = ops.spmm_rhs(graph, w, x, alpha, beta, x0, True).clone()   # the shipped default: variants that keep its summation order are bit-equal
reps = 20 if graph.e < 10_000_000 else 6
best = {}
for v in variants:
  _lib.check(_lib.lib().gnpde_tune(0, v))
  out = ops.spmm_rhs(graph, w, x, alpha, beta, x0, True)
  err = float((out - ref).abs().max() / ref.abs().max())
  res = {'variant': v, 'rel_max_vs_v0': err, 'bit_equal_to_default': bool(torch.equal(out, shipped))}
  for name, kw in STAGES.items():
    t = run(dict(kw), reps)
    res[name + '_us'] = round(t, 1)
    res[name + '_gbs'] = round(bytes_alg / t / 1e3, 0)
  best[v] = res['rhs_us']
  print(json.dumps(res), flush=True)
  assert err < 2e-6, 'variant %d disagrees with variant 0' % v
_lib.check(_lib.lib().gnpde_tune(0, 0))
bv = min(best, key=best.get)
print(json.dumps({'summary': spec, 'best_variant': bv, 'best_us': best[bv], 'v0_us': best.get(0), 'algorithmic_bytes': bytes_alg}))
