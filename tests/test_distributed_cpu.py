"""World-size-2 (and 3) gloo runs of the sharded solve on CPU: partition plan, halo index maps, the
all-to-all exchange and the sharded rk4 / euler driver against the unpartitioned oracle."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import gnpde_amd as G
from gnpde_amd import distributed as D
from helpers import random_graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_solve_gloo(world):
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py')]
  env = dict(os.environ, OMP_NUM_THREADS='2')
  res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
  assert res.returncode == 0 and 'DIST_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_plan_is_a_permutation_and_local_graphs_cover_all_edges():
  n = 500
  ei = random_graph(n, 6, seed=9, hubs=1, hub_deg=300)
  plan = D.PartitionPlan(ei, n, 4)
  assert torch.equal(torch.sort(plan.order)[0], torch.arange(n))
  seen = torch.zeros(ei.shape[1], dtype=torch.long)
  for r in range(4):
    sh = plan.shard(r)
    seen[sh.edge_ids] += 1
    # local column ids: owned first, halo after; every halo id maps back to the right global node
    glob_cols = ei[1][sh.edge_ids]
    loc = sh.edge_index[1]
    own = loc < sh.n_own
    assert torch.equal(sh.own_old_ids[loc[own]], glob_cols[own])
    assert torch.equal(plan.order[sh.halo_new[loc[~own] - sh.n_own]], glob_cols[~own])
    assert torch.equal(sh.own_old_ids[sh.edge_index[0]], ei[0][sh.edge_ids])
    # interior rows come first and never reference a halo column
    rows_with_halo = torch.unique(sh.edge_index[0][~own])
    assert rows_with_halo.numel() == sh.n_own - sh.n_interior
    assert rows_with_halo.numel() == 0 or int(rows_with_halo.min()) >= sh.n_interior
  assert torch.all(seen == 1)


def test_native_backend_refuses_cpu():
  n = 50
  ei = random_graph(n, 3, seed=1)
  plan = D.PartitionPlan(ei, n, 2)
  with pytest.raises(G.GnpdeError):
    D.NativeBackend(plan.shard(0), 8, 'cpu', 'laplacian', dict(edge_weight=torch.ones(1)), torch.tensor(0.), torch.tensor(0.))
