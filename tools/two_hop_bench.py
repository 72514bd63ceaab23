"""Time the two-hop densification (csrc/twohop.hip) against the expanded-list torch composite on the ogbn-arxiv shape.
   python tools/two_hop_bench.py [scale]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnpde_amd as G  # noqa: E402
from gnpde_amd import ops, synthetic  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device('cuda:0')
ei, n = synthetic.make_graph('arxiv', seed=0, scale=scale)[:2]
ei = ei.to(dev)
w = torch.rand(ei.shape[1], device=dev)
graph = G.CSRGraph(ei, n)
for rep in range(3):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  out_ei, out_w = ops.two_hop(graph, w)
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  print('native two_hop: n %d, nnz(A) %d -> nnz(S) %d, %.2f ms' % (n, ei.shape[1], out_ei.shape[1], (t1 - t0) * 1e3), flush=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import sparse_composite as B
try:
  for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    new_ei, new_w = B._spspmm(ei, w, ei, w, n)
    keep = new_ei[0] != new_ei[1]
    ref_ei, ref_w = B._coalesce(torch.cat([ei, new_ei[:, keep]], dim=1), torch.cat([w, new_w[keep]]) / 2, n)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('expanded-list composite: %.2f ms, peak memory %.2f GB' % ((t1 - t0) * 1e3, torch.cuda.max_memory_allocated() / 1e9))
  print('same entries:', torch.equal(ref_ei, out_ei), ' max rel diff %.2e' % float(((ref_w - out_w).abs() / ref_w.abs().clamp_min(1e-20)).max()))
except RuntimeError as exc:
  print('composite failed:', str(exc)[:200])
