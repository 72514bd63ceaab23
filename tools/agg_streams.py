"""What do the epilogue streams cost the aggregation?  (GPU box only.)  The arxiv-shaped graph: plain A u (gnpde_spmm: gathers + one
row written), f = alpha (A u - u) + beta x0 (+ u_i, x0_i read), and the rk4 stage variants (+ y / k1 read); and the pure gather of
the SAME column ids in CSR order, k per output row (gnpde_gather_ceiling)."""
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
d = cfg['d']
x, x0, y, k1, out = [torch.randn(n, d, device=dev) for _ in range(5)]
w = torch.rand(graph.e, device=dev) / 16
alpha, beta = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
E = graph.e


def t_of(fn, label, rows):
  t = bench.timed_replay(fn, 8)
  print(json.dumps({'what': label, 'us': round(t * 1e6, 1), 'row_gather_gbs': round(rows * 4 * d / t / 1e9, 1)}), flush=True)


t_of(lambda: ops.spmm(graph, w, x, out=out), 'plain aggregation A u (no epilogue operands)', E)
t_of(lambda: ops.spmm_rhs(graph, w, x, alpha, None, None, True, out=out), 'alpha (A u - u)  (reads u_i)', E)
t_of(lambda: ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, out=out), 'alpha (A u - u) + beta x0  (reads u_i, x0_i)', E)
t_of(lambda: ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, stage=_lib.STAGE_RK2C, dt=1.0, y=y, out_y=out), 'stage RK2C (+ y)', E)
t_of(lambda: ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, stage=_lib.STAGE_RK4C, dt=1.0, y=y, k1=k1, out_y=out), 'stage RK4C (+ y, k1)', E)
k = max(1, E // n)
col = graph.t['colidx'][: n * k].contiguous()
L = _lib.lib()
for variant in (0, 1):
  t_of(lambda: _lib.check(L.gnpde_gather_ceiling(_lib.ptr(x), n, d, d, _lib.ptr(col), k, _lib.ptr(out), n, variant, _lib.stream_of(x))),
       'pure gather of the graph\'s own column ids in CSR order, %d per output row (variant %d)' % (k, variant), n * k)
uni = torch.randint(0, n, (n * k,), device=dev, dtype=torch.int32)
t_of(lambda: _lib.check(L.gnpde_gather_ceiling(_lib.ptr(x), n, d, d, _lib.ptr(uni), k, _lib.ptr(out), n, 0, _lib.stream_of(x))),
     'pure gather, uniformly random ids', n * k)
