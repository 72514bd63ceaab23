"""Host-side profile (cProfile) of one adjoint training step at the ogbn-arxiv shape: where the Python time of the
backward goes (the backward is launch-bound).  GPU box only."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], 'none']
import train_bench as TB   # noqa: E402  (defines run(); runs nothing with the 'none' selector)
import torch

pr = cProfile.Profile()
orig_backward = torch.Tensor.backward


def profiled_backward(self, *a, **k):
  pr.enable()
  try:
    return orig_backward(self, *a, **k)
  finally:
    torch.cuda.synchronize()
    pr.disable()


torch.Tensor.backward = profiled_backward
TB.run('A', 128, {}, reps=2)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
print(s.getvalue()[:9000])
