"""Block size of the hashed rows -> XCDs deal on the relabelled ogbn-arxiv stand-in (gnpde_tune(12, s): blocks of 2^s rows; default
s = 4 at this size).  GPU box only.

The knob this script drives was taken out again after the measurement (no effect: profiles/r03_row_shift_ab.txt) so that
csrc/spmm.hip stays byte for byte the file the PMC records of profiles/hbm_traffic.json were taken with.  To repeat the run, add
to choose_row_shift (csrc/spmm.hip), before `int s = 4;`:
    if (g_tune[12] >= 4 && g_tune[12] <= 10) return g_tune[12];"""
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
