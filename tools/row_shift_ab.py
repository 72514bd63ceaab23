"""Block size of the hashed rows -> XCDs deal on the relabelled ogbn-arxiv stand-in (gnpde_tune(12, s): blocks of 2^s rows; default
s = 4 at this size).  GPU box only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops

dev = torch.device('cuda:0')
cfg = G.synthetic.CONFIGS['arxiv']
ei, n = G.synthetic.make_graph('arxiv')
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
base = G.CSRGraph(ei2.to(dev), n)
view = base.locality_view(4 * cfg['d'], 'parts')
for s in (0, 4, 5, 6, 7, 8, 9, 10, 0):
  ops.tune(12, s)
  t = min(view.graph._aggregation_time(cfg['d'], reps=10) for _ in range(3))
  print(json.dumps({'row_shift': s or 'default', 'plain_aggregation_us': round(t * 1e6, 1)}), flush=True)
ops.tune(12, 0)
