"""A/B of the q||k projection kernels at the ogbn-arxiv shape ([169343, 128] x [128, 32]): us per launch over rotating L2-cold inputs.

  python tools/linear_ab.py [--n 169343] [--d 128] [--m 32]

gnpde_tune(8, k): 0 = linear_staged_kernel (round 3), 6 / 7 / 8 = linear_staged2_kernel paired columns / two tiles in flight / both,
9 / 10 / 11 = the same three on a grid of 4 / 4 / 2 workgroups per CU; gnpde_tune(13, 1 | 2) with knob 6: no stores / loads alone."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnpde_amd as G   # noqa: E402
from gnpde_amd import ops, _lib   # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=169343)
  ap.add_argument('--d', type=int, default=128)
  ap.add_argument('--m', type=int, default=32)
  ap.add_argument('--reps', type=int, default=200)
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  g = torch.Generator().manual_seed(0)
  xs = [torch.randn(args.n, args.d, generator=g).to(dev) for _ in range(5)]      # 5 x 87 MB > the 256-MiB Infinity Cache
  w = (torch.randn(args.m, args.d, generator=g) / args.d ** 0.5).to(dev)
  b = torch.randn(args.m, generator=g).to(dev)
  outs = [torch.empty(args.n, args.m, device=dev) for _ in range(5)]
  ref = None
  res = {}
  for knob, diag in [(12, 0), (0, 0), (6, 0), (7, 0), (8, 0), (9, 0), (10, 0), (11, 0), (6, 1), (6, 2), (2, 0), (12, 0), (0, 0)]:
    ops.tune(_lib.TUNE_LINEAR_STREAMING, knob)
    ops.tune(13, diag)
    for i in range(5):
      ops.linear(xs[i], w, b, out=outs[i])
    torch.cuda.synchronize()
    if diag == 0:
      if ref is None:
        ref = outs[0].clone()
      same = bool(torch.equal(ref, outs[0]))
    else:
      same = None
    times = []
    for _ in range(3):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for r in range(args.reps):
        ops.linear(xs[r % 5], w, b, out=outs[r % 5])
      e1.record()
      torch.cuda.synchronize()
      times.append(e0.elapsed_time(e1) * 1e3 / args.reps)
    us = sorted(times)[1]
    nbytes = args.n * (args.d + args.m) * 4
    res['knob%d_diag%d' % (knob, diag)] = {'us': round(us, 2), 'gbs': round(nbytes / us / 1e3, 1), 'bit_identical_to_knob0': same}
    print(json.dumps({'knob': knob, 'diag': diag, 'us': round(us, 2), 'tbs': round(nbytes / us / 1e6, 2), 'same': same}), flush=True)
  ops.tune(_lib.TUNE_LINEAR_STREAMING, 0)
  ops.tune(13, 0)


if __name__ == '__main__':
  main()
