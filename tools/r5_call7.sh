#!/bin/bash
set -u
OUT=gpurun_out/r5c7
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -30 | tee $OUT/adjoint_tests.log
python - <<'PY' 2>&1 | tee $OUT/adaptive_adjoint_timing.txt
# backward time of a Pubmed-shaped block (19 717 nodes, 88 648 entries + loops, d = 128; attention block, Laplacian, dopri5 forward,
# adjoint adaptive_heun as best_params Pubmed): native stages vs the flat host loop
import time, torch, sys
sys.path.insert(0, '.')
import gnpde_amd as G
from tests.helpers import Data, random_graph
dev = torch.device('cuda:0')
n, d = 19717, 128
ei = random_graph(n, 4, seed=5).to(dev)
x = (torch.randn(n, d, generator=torch.Generator().manual_seed(6)) * 0.5).to(dev)
base = dict(heads=1, attention_dim=16, attention_type='cosine_sim', attention_norm_idx=0, square_plus=True, reweight_attention=False, beltrami=False,
            leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=5000, add_source=True, no_alpha_sigmoid=False, mix_features=False, hidden_dim=d,
            augment=False, adjoint=True, adjoint_method='adaptive_heun', adjoint_step_size=1, tol_scale=1991.07, tol_scale_adjoint=16324.37,
            data_norm='rw', method='dopri5', step_size=1, max_iters=100, block='attention', function='laplacian', time=12.94)
for host in (False, True):
  opt = dict(base, gnpde_host_adjoint=host)
  block = G.AttODEblock(G.LaplacianODEFunc, [], opt, Data(x, ei), dev, t=torch.tensor([0, opt['time']])).to(dev)
  g = torch.Generator().manual_seed(1)
  with torch.no_grad():
    for name, p in block.named_parameters():        # (every parameter seeded: nn.Linear's default bias init would make the two blocks differ)
      if p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      elif name.endswith('.bias'):
        p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
  block.train()
  ts = []
  for it in range(4):
    xin = x.clone().requires_grad_(True)
    block.set_x0(xin)
    block.odefunc.nfe = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z = block(xin)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    nf = block.odefunc.nfe
    z.sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, nf, block.odefunc.nfe - nf))
  f, b, nf, nb = ts[-1]
  print('host flat loop' if host else 'native stages ', 'forward %.2f ms (%d evals)  backward %.2f ms (%d augmented evals)' % (f * 1e3, nf, b * 1e3, nb))
PY
