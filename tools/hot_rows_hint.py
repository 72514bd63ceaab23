"""Experiment (GPU box only): does a nontemporal hint on the rows that are referenced rarely keep the often-referenced rows in
the XCD's L2?  gnpde_gather_ceiling with the reference distribution of the ogbn-arxiv stand-in (propensity ~ rank^-0.75,
shuffled ids, 15 references per output row): plain loads vs the hint on every id outside the hottest `frac` of the nodes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gnpde_amd import _lib
import bench

dev = torch.device('cuda:0')
n, d, k = 169343, 128, 15
x = torch.randn(n, d, device=dev)
out = torch.empty_like(x)
L = _lib.lib()
rng = np.random.default_rng(0)
w = np.arange(1, n + 1, dtype=np.float64) ** (-0.75)
perm = rng.permutation(n)                       # rank -> node id
cdf = np.cumsum(w)
ranks = np.minimum(np.searchsorted(cdf, rng.random(n * k) * cdf[-1]), n - 1)
ids = torch.from_numpy(perm[ranks].astype(np.int64))
rank_of = torch.from_numpy(ranks)


def run(idx, variant, label):
  idx = idx.to(torch.int32).contiguous().to(dev)

  def call():
    _lib.check(L.gnpde_gather_ceiling(_lib.ptr(x), n, d, d, _lib.ptr(idx), k, _lib.ptr(out), n, variant, _lib.stream_of(x)))
  t = bench.timed_replay(call, 8)
  print(json.dumps({'what': label, 'us': round(t * 1e6, 1), 'row_gather_gbs': round(n * k * 4 * d / t / 1e9, 1)}), flush=True)
  return out.clone()


ref = run(ids, 0, 'power-law references, plain loads')
for frac in (0.005, 0.01, 0.02, 0.03, 0.05, 0.10):
  hot = rank_of < int(frac * n)
  enc = torch.where(hot, ids, ids | (1 << 31)) if False else torch.where(hot, ids, ids - (1 << 31))   # top bit set = cold (as int32: negative)
  got = run(enc, 2, 'nontemporal hint outside the hottest %.1f %% of the nodes (%.0f %% of the references inside, %.2f MB)'
            % (100 * frac, 100 * float(hot.float().mean()), frac * n * d * 4 / 1e6))
  assert torch.equal(got, ref)
run(ids - (1 << 31), 2, 'nontemporal hint on every reference')
