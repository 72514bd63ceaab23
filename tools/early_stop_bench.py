"""Time of one early-stopping evaluation (decode + arg-max + counts) at the ogbn-arxiv shape, and its read rate."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops

dev = torch.device('cuda:0')
for n, d, c in [(169343, 128, 40), (169343, 128, 7), (2097152, 256, 40), (2708, 80, 7)]:
  g = torch.Generator().manual_seed(0)
  y = torch.randn(n, d, generator=g).to(dev)
  w = (torch.randn(c, d, generator=g) / d ** 0.5).to(dev)
  lab = torch.randint(0, c, (n,), generator=g).to(dev)
  role = torch.rand(n, generator=g).to(dev)
  ev = ops.EarlyStopEvaluator(w, torch.zeros(c, device=dev), lab, role < 0.5, (role >= 0.5) & (role < 0.7), role >= 0.7)
  ev.reset()
  for _ in range(3):
    ev.evaluate(y, 1)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  reps = 50
  a.record()
  for i in range(reps):
    ev.evaluate(y, i)
  b.record()
  torch.cuda.synchronize()
  us = a.elapsed_time(b) * 1e3 / reps
  print('n=%d d=%d classes=%d: %.1f us per evaluation (two launches), state read at %.0f GB/s' % (n, d, c, us, n * d * 4 / us / 1e3))
