"""How much would locality buy the aggregation?  (GPU box only.)  gnpde_gather_ceiling on the ogbn-arxiv-sized table with index
distributions from 'no locality' to 'all neighbours within a window of the output row' (XCD x owns the x-th eighth of the output
rows, so a window is an XCD-local working set).  Prints us per launch and the row-gather rate."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import _lib
import bench

dev = torch.device('cuda:0')
n, d, k = 169343, 128, 15
x = torch.randn(n, d, device=dev)
out = torch.empty_like(x)
L = _lib.lib()
gen = torch.Generator(device=dev).manual_seed(1)
rows = torch.arange(n, device=dev).repeat_interleave(k)


def run(idx, label):
  idx = idx.to(torch.int32).contiguous()
  best = None
  for variant in (0, 1):
    def call():
      _lib.check(L.gnpde_gather_ceiling(_lib.ptr(x), n, d, d, _lib.ptr(idx), k, _lib.ptr(out), n, variant, _lib.stream_of(x)))
    t = bench.timed_replay(call, 8)
    best = t if best is None or t < best else best
  print(json.dumps({'indices': label, 'us': round(best * 1e6, 1), 'row_gather_gbs': round(n * k * 4 * d / best / 1e9, 1)}), flush=True)


uni = torch.randint(0, n, (n * k,), device=dev, generator=gen)
run(uni, 'uniform over the table')
for win in (256, 2048, 8192, 32768):
  loc = (rows + torch.randint(-win, win + 1, (n * k,), device=dev, generator=gen)).clamp_(0, n - 1)
  run(loc, 'within +-%d rows of the output row (%.1f MB window)' % (win, 2 * win * d * 4 / 1e6))
  mix = torch.where(torch.rand(n * k, device=dev, generator=gen) < 2.0 / 3.0, loc, uni)
  run(mix, '2/3 within +-%d rows, 1/3 uniform' % win)
