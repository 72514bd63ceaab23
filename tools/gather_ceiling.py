"""What does the memory system give for random row gathers?  (GPU box only.)  Rows of d floats gathered by a
random index list from a table that does / does not fit the 256 MiB Infinity Cache."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import _lib
dev = torch.device('cuda:0')
L = _lib.lib()
def bench(n_rows, d, n_idx, reps=5):
  src = torch.randn(n_rows, d, device=dev)
  idx = torch.randint(0, n_rows, (n_idx,), device=dev, dtype=torch.int32)
  dst = torch.empty(n_idx, d, device=dev)
  def run():
    _lib.check(L.gnpde_gather_rows(_lib.ptr(src), d, _lib.ptr(idx), n_idx, d, _lib.ptr(dst), d, _lib.stream_of(src)))
  run(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): run()
  e1.record(); torch.cuda.synchronize()
  t = e0.elapsed_time(e1) * 1e-3 / reps
  idl = idx.long()
  torch.index_select(src, 0, idl, out=dst); torch.cuda.synchronize()
  e0.record()
  for _ in range(reps): torch.index_select(src, 0, idl, out=dst)
  e1.record(); torch.cuda.synchronize()
  t2 = e0.elapsed_time(e1) * 1e-3 / reps
  gb = n_idx * d * 4 / 1e9
  print('table %6.0f MB, rows of %4d B, %5.1f GB gathered: gather_rows %.0f GB/s read (+ same written), index_select %.0f GB/s'
        % (n_rows * d * 4 / 1e6, d * 4, gb, gb / t, gb / t2), flush=True)
bench(169343, 128, 2_480_000 * 4)
bench(2_097_152, 128, 20_000_000)
bench(2_097_152, 256, 20_000_000)
bench(8_000_000, 256, 20_000_000)
