"""One short GPU check of the rows -> XCD deal of the aggregation kernels (hashed blocks, csrc/spmm.hip) against the contiguous
eighths (gnpde_tune(10, 2) / (10, 1) force one or the other): which XCD takes a row must not enter the arithmetic, so every kernel family has to
give BIT-identical results under both deals -- with hub rows, with a row_begin (the boundary pass of a partitioned graph),
with the rk4 stage epilogues -- and agree with a dense-free PyTorch evaluation of the same sum; then the launch time of
both at the ogbn-arxiv shape.  Prints JSON lines; exits non-zero on any mismatch.

  python tools/xcd_check.py [--no-timing]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib
from helpers import random_graph

dev = torch.device('cuda:0')
L = _lib.lib()
t_start = time.time()


def deal(contiguous):   # None: as the graph says (gnpde_graph_t.xcd_deal)
  _lib.check(L.gnpde_tune(_lib.TUNE_XCD_ROWS, 0 if contiguous is None else (1 if contiguous else 2)))


def both(fn):
  deal(False)
  a = fn()
  deal(True)
  b = fn()
  deal(None)
  return a, b


failures = 0
n = 30011
ei = random_graph(n, 8, seed=5, hubs=3, hub_deg=1700, isolated=9, dup=50)
graph = G.CSRGraph(ei.to(dev), n)
E = graph.e
gen = torch.Generator().manual_seed(1)
w_edge = torch.rand(ei.shape[1], generator=gen) / 8
w = ops.edge_to_csr_mean(graph, w_edge.to(dev))
alpha, beta = torch.tensor([0.3], device=dev), torch.tensor([0.2], device=dev)
for d in (128, 256, 96, 200, 64, 6, 33):
  u = torch.randn(n, d, generator=gen).to(dev)
  x0, y, k1 = (torch.randn(n, d, generator=gen).to(dev) for _ in range(3))
  for name, kw in (('rhs', {}), ('rk2c', dict(stage=_lib.STAGE_RK2C, dt=0.7, y=y)), ('rk4c', dict(stage=_lib.STAGE_RK4C, dt=0.7, y=y, k1=k1))):
    def f():
      if not kw:
        return ops.spmm_rhs(graph, w, u, alpha, beta, x0, True).clone()
      out = torch.empty_like(u)
      ops.spmm_rhs(graph, w, u, alpha, beta, x0, True, out_y=out, **kw)
      return out
    a, b = both(f)
    ok = bool(torch.equal(a, b)) and bool(torch.isfinite(a).all())
    failures += 0 if ok else 1
    rec = {'case': 'd=%d %s' % (d, name), 'bit_equal': ok}
    if not kw:   # against index_add on the device: f = sigmoid(alpha) (A u - u) + beta x0
      ax = torch.zeros_like(u).index_add_(0, ei[0].to(dev), u[ei[1].to(dev)] * w_edge.to(dev).unsqueeze(1))
      ref = torch.sigmoid(alpha) * (ax - u) + beta * x0
      rec['rel_max_vs_index_add'] = float((a - ref).abs().max() / ref.abs().max())
      if not rec['rel_max_vs_index_add'] < 1e-5:
        failures += 1
    print(json.dumps(rec), flush=True)
# boundary pass of a partitioned graph: rows [row_begin, n) only
for d in (128, 256):
  u = torch.randn(n, d, generator=gen).to(dev)
  x0 = torch.randn(n, d, generator=gen).to(dev)
  for rb in (1, 12345):
    def f():
      graph.struct.row_begin = rb
      try:
        out = torch.zeros_like(u)
        ops.spmm_rhs(graph, w, u, alpha, beta, x0, True, out=out)
        return out
      finally:
        graph.struct.row_begin = 0
    a, b = both(f)
    full = ops.spmm_rhs(graph, w, u, alpha, beta, x0, True)
    # hub rows below row_begin are still written by the fold of their chunks: compare the rows of the pass
    ok = bool(torch.equal(a[rb:], b[rb:])) and bool(torch.equal(a[rb:], full[rb:]))
    failures += 0 if ok else 1
    print(json.dumps({'case': 'd=%d row_begin=%d' % (d, rb), 'bit_equal': ok}), flush=True)
print(json.dumps({'checks_s': round(time.time() - t_start, 1), 'failures': failures}), flush=True)

if '--no-timing' not in sys.argv:
  ei, n = G.synthetic.make_graph('arxiv')
  ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
  graph = G.CSRGraph(ei2.to(dev), n)
  d = 128
  x = torch.randn(n, d, device=dev)
  x0, k1 = torch.randn_like(x), torch.randn_like(x)
  w = torch.rand(graph.e, device=dev) / 16

  def timed(reps=40):
    for _ in range(3):
      ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, out=k1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      ops.spmm_rhs(graph, w, x, alpha, beta, x0, True, out=k1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
  res = {'hashed_blocks_us': [], 'contiguous_eighths_us': []}
  for _ in range(3):
    deal(False)
    res['hashed_blocks_us'].append(round(timed(), 1))
    deal(True)
    res['contiguous_eighths_us'].append(round(timed(), 1))
  deal(None)
  res['graph'] = 'arxiv d=128 (spmm_pair_kernel + spmm_long_reduce_kernel, RHS epilogue)'
  print(json.dumps(res), flush=True)
print(json.dumps({'total_s': round(time.time() - t_start, 1)}))
sys.exit(1 if failures else 0)
