"""A/B of the fixed-step solver's projection tracking (GPU box only): the BASELINE C3 solve (or another bench graph) with a fresh
q||k projection in every evaluation (refresh 0, merged and two-launch row attention) against tracking with a refresh every
1 / 2 / 4 / never steps.  Prints ms per step and the difference of the final state to the refresh-0 solve.

  python tools/tracking_ab.py [arxiv|rmat|cora] [STEPS]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
import bench

name = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
cfg = G.synthetic.CONFIGS[name]
ei_cpu, n = G.synthetic.make_graph(name)
x = torch.randn(n, cfg['d'], generator=torch.Generator().manual_seed(0)).to(dev)
ei = ei_cpu.to(dev)


class A(object):
  heads = None; att_dim = None; norm_idx = 0; square_plus = False; function = 'transformer'; steps = K


def run(refresh, tune=None, replays=3):
  if tune:
    G.ops.tune(*tune)
  opt = dict(bench.build_opt(cfg, A), gnpde_projection_refresh=refresh)
  block = bench.make_block(G, opt, ei, n, x, dev, float(K), 0)
  block.set_x0(x)
  with torch.no_grad():
    z = block(x)
    ts = []
    for _ in range(replays):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      z = block(x)
      torch.cuda.synchronize()
      ts.append(time.perf_counter() - t0)
  solver = next(iter(block.odefunc._solver_state.values()))['solver']
  if tune:
    G.ops.tune(tune[0], 0)
  return z.clone(), 1e3 * sorted(ts)[len(ts) // 2] / K, solver.projection_refresh


z0, ms0, r0 = run(0)
print(json.dumps({'graph': name, 'steps': K, 'refresh': 0, 'ms_per_step': round(ms0, 4), 'tracking': r0}), flush=True)
z0b, ms0b, _ = run(0, tune=(12, 1))
print(json.dumps({'refresh': 0, 'row_attention': 'two launches (round 2)', 'ms_per_step': round(ms0b, 4),
                  'bit_equal_to_merged': bool(torch.equal(z0, z0b))}), flush=True)
for refresh in (1, 2, 4, 1000000):
  z, ms, r = run(refresh)
  d = (z.double() - z0.double())
  print(json.dumps({'refresh': refresh, 'tracking': r, 'ms_per_step': round(ms, 4),
                    'rel_max_vs_refresh0': float(d.abs().max() / z0.double().abs().max()),
                    'rel_l2_vs_refresh0': float(d.norm() / z0.double().norm())}), flush=True)
