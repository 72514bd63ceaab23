"""The native row-partitioned solver (csrc/sharded.hip) on ONE GPU.

* P2P transport (the product default): REAL 2- and 4-way partitions, one process per rank, all ranks on cuda:0.  The ranks
  map each other's stage buffers through IPC handles exactly as they would across GPUs; boundary rows are pushed into the
  peers' halo regions and awaited through epoch flags inside each rank's hipGraph (tests/dist_gpu_worker.py).
* RCCL transport (eager launches): a single process cannot own two RCCL ranks on one device, so the exchange is exercised
  as a SELF exchange: the shard below mirrors a third of its own rows into a halo region, the boundary rows read those
  mirrored copies instead of the originals, and every evaluation has to refresh them through ncclSend / ncclRecv to self.
  If the exchange were skipped, stale or mis-ordered, the boundary rows would integrate garbage (the halo starts as NaN).
Index maps of the partitions are also covered on CPU by the gloo tests (tests/test_distributed_cpu.py).
"""
import json
import os
import subprocess
import sys
import pytest
import torch

import gnpde_amd as G
from gnpde_amd import distributed as D, _lib
from oracle import restate as R
from helpers import assert_parity, random_graph

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method='thread')]


class SelfShard(object):
  """World-1 shard whose boundary rows reference MIRRORED copies of owned rows (see module docstring)."""

  def __init__(self, ei, n, frac_interior=0.6, frac_mirror=0.35, seed=0):
    g = torch.Generator().manual_seed(seed)
    self.rank, self.world = 0, 1
    self.n_own = n
    self.n_interior = int(n * frac_interior)
    mirrored = torch.sort(torch.randperm(n, generator=g)[:int(n * frac_mirror)]).values
    pos = torch.full((n,), -1, dtype=torch.long)
    pos[mirrored] = torch.arange(mirrored.numel())
    row, col = ei
    redirect = (row >= self.n_interior) & (pos[col] >= 0)
    col_local = torch.where(redirect, n + pos[col], col)
    self.edge_index = torch.stack([row, col_local])
    self.n_halo = int(mirrored.numel())
    self.send_idx = mirrored
    self.send_counts = [self.n_halo]
    self.recv_counts = [self.n_halo]
    self.n_redirected = int(redirect.sum())

  @property
  def n_local(self):
    return self.n_own + self.n_halo


def _problem(kind, seed=3):
  n, d, A, h = 6000, 128, 16, 4
  ei = random_graph(n, 8, seed=seed, hubs=2, hub_deg=1500)
  g = torch.Generator().manual_seed(seed)
  x = torch.randn(n, d, generator=g)
  params = dict(Wq=torch.randn(A, d, generator=g) / d ** 0.5, Wk=torch.randn(A, d, generator=g) / d ** 0.5,
                bq=torch.randn(A, generator=g) * 0.1, bk=torch.randn(A, generator=g) * 0.1, heads=h)
  alpha, beta = torch.tensor(0.3), torch.tensor(0.2)
  if kind == 'laplacian':
    _, w = G.get_rw_adj(ei, None, norm_dim=0, fill_value=0.0, num_nodes=n, dtype=torch.float32)
    rhs = lambda t, y: R.rhs_laplacian(y, ei, w, alpha, beta, x, False, True)   # noqa: E731
    p = dict(edge_weight=w)
  else:
    rhs = lambda t, y: R.rhs_transformer(y, ei, params['Wq'], params['bq'], params['Wk'], params['bk'], h, alpha, beta,  # noqa: E731
                                         x, False, True)
    p = params
  return n, d, ei, x, p, alpha, beta, rhs


@pytest.mark.parametrize('kind', ['transformer', 'laplacian'])
@pytest.mark.parametrize('method,T', [('rk4', 3.0), ('euler', 2.5)])
def test_rccl_self_exchange_eager(dev, kind, method, T):
  n, d, ei, x, p, alpha, beta, rhs = _problem(kind)
  sh = SelfShard(ei, n)
  assert sh.n_redirected > 1000 and 0 < sh.n_interior < sh.n_own
  be = D.NativeBackend(sh, d, dev, kind, p, alpha, beta, True)
  ref = R.odeint_fixed(rhs, x, T, 1.0, method)
  xd = x.to(dev)
  solver = D.NativeShardedSolver(sh, be, T, 1.0, method, transport='rccl')
  solver.y.fill_(float('nan'))                   # halo rows (and everything else) start poisoned
  with torch.no_grad():
    z = solver.integrate(xd, xd, use_graph=False).clone()
    z2 = solver.integrate(xd, xd, use_graph=False).clone()
  assert torch.equal(z, z2)
  assert solver.n_rhs_evals == (4 if method == 'rk4' else 1) * len(R.time_grid(T, 1.0)[1:])
  assert_parity(z, ref, what='RCCL self-exchange %s %s' % (kind, method))
  with pytest.raises(_lib.GnpdeError):           # capture of the RCCL transport is refused, not attempted (HIP runtime bug)
    solver.integrate(xd, xd, use_graph=True)
  solver.close()


@pytest.mark.parametrize('world', [2, 4])
@pytest.mark.parametrize('kind,method,T', [('transformer', 'rk4', 3.0), ('laplacian', 'euler', 2.5)])
def test_ranks_sharing_one_gpu(dev, tmp_path, world, kind, method, T):
  """P2P transport, real partitions: `world` processes on this one GPU (module docstring)."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / 'result.json')
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='4')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
         '--master-port', str(29600 + world), os.path.join(root, 'tests', 'dist_gpu_worker.py'), out, kind, method, str(T)]
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
  r = json.load(open(out))
  assert r['world'] == world and r['halo_rows'] > 0 and 0 < r['interior_rows'] < r['own_rows']
  assert r['rel_max'] < 1e-5 and r['rel_l2'] < 1e-5, r


@pytest.mark.parametrize('world', [2, 3])
def test_blocks_run_sharded_from_the_operator_surface(dev, tmp_path, world):
  """opt['gnpde_shard'] + an initialised process group: ODEblock.forward (the call the reference makes at
  src/block_constant.py:57-62) partitions the graph over the ranks and runs the native sharded solver -- against the
  reference-recorded fixtures, with the reference's NFE, every rank holding the whole result (tests/dist_block_worker.py)."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / 'blocks.json')
  names = ('block_constant_transformer_rk4,block_constant_laplacian_euler,block_attention_laplacian_euler,block_constant_transformer_sqp_n1_rk4,'
           'block_attention_laplacian_dopri5,block_constant_transformer_dopri5,block_constant_gat_rk4,block_attention_laplacian_beltrami_rk4,'
           # BLEND's split feature x positional kernel as a per-evaluation attention (no recorded block solve: against the one-GPU solve)
           'selfcheck:func_transformer_beltrami_expkernel,selfcheck:func_transformer_beltrami_expkernel_sqp,'
           # re-weighted attention (edge_attr data): row softmax in the in-graph solver, column softmax in the exchange loop
           'selfcheck_rw:func_transformer_sd_softmax_n0,selfcheck_rw:func_transformer_sd_softmax_n1,'
           # dopri5 of a function normalised over COLUMNS: the device controller over the engine's general mode (round 6)
           'selfcheck_dopri5:func_transformer_sd_softmax_n1')
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='4')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
         '--master-port', str(29640 + world), os.path.join(root, 'tests', 'dist_block_worker.py'), out, names]
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
  r = json.load(open(out))
  for name in names.split(','):
    e = r[name]
    assert e['world'] == world and e['halo_rows'] > 0, e
    assert e['rel_max'] < 1e-5 and e['rel_l2'] < 1e-5, (name, e)
    assert e['nfe'] == e['ref_nfe'] and e['replay_equal'] and e['ranks_agree'], (name, e)
    assert e.get('moved', 1.0) > 1e-3, (name, e)       # (self-checks: the solve did something)
    if name == 'block_constant_transformer_sqp_n1_rk4':
      # squareplus over columns: the exchanges between the attention passes ride in the per-rank graph (gnpde_sharded_solver_set_general)
      assert e['solvers'] == ['NativeShardedSolver'], (name, e)
    if name.startswith('selfcheck_dopri5:'):
      assert e['solvers'] == ['NativeShardedDopri5'], (name, e)
    if name.endswith('_dopri5'):
      # the adaptive blocks take the device controller over the partition (gnpde_dopri5_create_sharded): fewer reads of the
      # controller record than trial steps, none per trial step
      assert e['solvers'] == ['NativeShardedDopri5'] and 0 < e['syncs'] < e['trials'], (name, e)


@pytest.mark.parametrize('kind,method,T', [('laplacian', 'dopri5', 2.5), ('transformer', 'dopri5', 1.5), ('laplacian', 'adaptive_heun', 1.0)])
def test_adaptive_methods_run_partitioned(dev, tmp_path, kind, method, T):
  """dopri5 / adaptive_heun over the row partition (3 ranks sharing the GPU): every stage one halo exchange + one fused HIP
  evaluation, the error norm all-reduced, torchdiffeq 0.2.1's controller on every rank -- the same number of evaluations (= the
  same accept / reject sequence) and the same state as the restated torchdiffeq over the CPU oracle on the whole graph."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / 'result.json')
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='4')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3', '--master-addr', '127.0.0.1',
         '--master-port', '29670', os.path.join(root, 'tests', 'dist_gpu_worker.py'), out, kind, method, str(T)]
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
  r = json.load(open(out))
  assert r['world'] == 3 and r['halo_rows'] > 0 and r['evals'] == r['ref_evals'], r
  assert r['rel_max'] < 1e-5 and r['rel_l2'] < 1e-5, r
  # the controller on every rank's device (gnpde_dopri5_create_sharded [+ gnpde_dopri5_set_pair], the path the operator surface takes):
  # the same evaluation count and state as the reference's controller, the same decisions on every rank (asserted in the worker), and
  # FEWER host reads than trial steps -- none per trial step: one per batch, batches as long as the end point allows
  per_trial = 6 if method == 'dopri5' else 1
  assert r['native_evals'] == r['ref_evals'] and r['native_trials'] * per_trial + 2 == r['native_evals'], r
  assert r['native_rel_max'] < 1e-5 and r['native_rel_l2'] < 1e-5 and r['native_vs_host_controller'] < 1e-5, r
  assert r['native_syncs'] < r['native_trials'] and r['native_syncs_per_trial_batch1'] == r['native_trials'], r
  assert 8 < r['native_budget_evals'] < r['native_evals'], r      # (the budget is looked at once per batch of trial steps)


@pytest.mark.parametrize('kind,method', [('gat', 'rk4'), ('gat_n1', 'rk4'), ('gat', 'dopri5')])
def test_gat_function_runs_partitioned(dev, tmp_path, kind, method):
  """ODEFuncAtt's right-hand side (reference src/function_GAT_attention.py:45-65, :105-115) over the row partition, 2 ranks sharing
  the GPU: the projections of the halo rows are recomputed locally like the transformer's keys.  Softmax over rows: P2P transport
  inside each rank's hipGraph; over columns (attention_norm_idx = 1): the loop with exchanges between the attention passes;
  dopri5: the all-reduced error norm.  Against the unpartitioned CPU oracle."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / 'result.json')
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='4')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29693', os.path.join(root, 'tests', 'dist_gpu_worker.py'), out, kind, method, '2.0']
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
  r = json.load(open(out))
  assert r['world'] == 2 and r['halo_rows'] > 0
  assert r['rel_max'] < 1e-5 and r['rel_l2'] < 1e-5, r
  if method == 'dopri5':
    assert r['evals'] == r['ref_evals'], r


def test_sharding_request_without_a_supported_configuration_fails_loudly(dev):
  """gnpde_shard on a configuration the partitioned solver does not cover (mix_features)
  raises instead of silently running on one GPU; without a process group the request is ignored (single-GPU solve)."""
  import torch.distributed as dist
  from helpers import Fixture, Data
  fx = Fixture('block_constant_transformer_rk4')
  x = fx.t('x', dev)
  opt = dict(fx.opt, gnpde_shard=1)
  block = G.ConstantODEblock(G.ODEFuncTransformerAtt, [], opt, Data(x, fx.t('edge_index', dev)), dev,
                             t=torch.tensor([0, opt['time']])).to(dev)
  block.load_state_dict(fx.params, strict=True)
  block.eval()
  block.set_x0(x)
  assert not dist.is_initialized()
  with torch.no_grad():
    z = block(x)                                   # no process group: the request is void
  assert_parity(z, fx.t('z'), what='unsharded')
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29677')
  dist.init_process_group('gloo', rank=0, world_size=1)
  try:
    block.odefunc.opt = dict(block.odefunc.opt, mix_features=True)               # not covered by the partitioned solver
    block.set_x0(x)
    with torch.no_grad(), pytest.raises(_lib.GnpdeError):
      block(x)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('kind', ['transformer_n1', 'transformer_sqp', 'transformer_sqp_n1'])
def test_normalisers_that_are_not_row_local(dev, tmp_path, world, kind):
  """attention_norm_idx = 1 needs the column sums (partial statistics of the halo columns go to their owners, are
  merged there and come back), squareplus the global maximum (a scalar MAX all-reduce) -- real partitions, `world` processes on
  this one GPU, against the unpartitioned CPU oracle (reference src/function_transformer_attention.py:210-213, src/utils.py:196)."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / 'result.json')
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='4')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
         '--master-port', str(29660 + world), os.path.join(root, 'tests', 'dist_gpu_worker.py'), out, kind, 'rk4', '2.5']
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
  r = json.load(open(out))
  assert r['world'] == world and r['halo_rows'] > 0
  assert r['rel_max'] < 1e-5 and r['rel_l2'] < 1e-5, r
  # the same exchanges INSIDE the per-rank graph (gnpde_sharded_solver_set_general: the partial column statistics pushed to their owners,
  # merged in rank order, pushed back; the squareplus maximum through the flag blocks): eager = replayed graph, and the same kernels in
  # the same order as the Python-driven loop
  assert r['native_replay_equal'] and r['native_vs_loop'] < 1e-6, r
  assert r['native_rel_max'] < 1e-5 and r['native_rel_l2'] < 1e-5, r


@pytest.mark.parametrize('kind', ['cosine_sim', 'pearson', 'exp_kernel'])
def test_other_score_functions_run_partitioned(dev, tmp_path, kind):
  """The row-local score functions besides the scaled dot product (reference src/function_transformer_attention.py:193-206): the
  partitioned evaluation only differs in the per-edge formula -- 2 processes on this one GPU, P2P transport inside each rank's
  hipGraph, against the unpartitioned CPU oracle."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / 'result.json')
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='4')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29691', os.path.join(root, 'tests', 'dist_gpu_worker.py'), out, kind, 'rk4', '3.0']
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
  r = json.load(open(out))
  assert r['world'] == 2 and r['halo_rows'] > 0 and 0 < r['interior_rows'] < r['own_rows']
  assert r['rel_max'] < 1e-5 and r['rel_l2'] < 1e-5, r


def test_no_exchange_world1_matches_single_gpu_solver(dev):
  """A world-1 shard without halo (comm NULL): interior = all rows; must equal the single-GPU solver bit for bit."""
  n, d, ei, x, p, alpha, beta, rhs = _problem('transformer', seed=5)
  plan = D.PartitionPlan(ei, n, 1)
  sh = plan.shard(0)
  assert sh.n_halo == 0 and sh.n_interior == sh.n_own
  be = D.NativeBackend(sh, d, dev, 'transformer', p, alpha, beta, True)
  solver = D.NativeShardedSolver(sh, be, 2.0, 1.0, 'rk4')     # P2P transport, world 1: no peers, nothing to exchange
  x_own = D.scatter_rows(x, sh).to(dev)
  with torch.no_grad():
    z = solver.integrate(x_own, x_own)
  ref = R.odeint_fixed(rhs, x, 2.0, 1.0, 'rk4')
  assert_parity(z, ref[sh.own_old_ids], what='world-1 native sharded solve')


@pytest.mark.timeout(600, method='thread')
def test_bench_gpus_2_launches_its_own_ranks_and_prints_a_parseable_headline():
  """`python bench.py --gpus 2` without a launcher (how the driver runs `--gpus 1`): the script spawns its ranks itself
  (bench.launch_ranks); on this one-GPU box the two ranks share the device (a functional run).  The LAST stdout line is the compact
  headline with n_gpus = 2."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, GNPDE_RANKS_SHARE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--scale', '0.25'],
                       env=env, capture_output=True, text=True, timeout=540)
  assert res.returncode == 0, res.stderr[-2000:]
  lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
  head = json.loads(lines[-1])
  assert len(lines[-1]) < 4096
  assert head['n_gpus'] == 2 and head['steps'] == 4 and head['unit'] == 'steps/s' and head['value'] > 0
  assert head['config']['ranks_seen'] == 2 and head['config']['ranks_share_one_device'] is True
  assert head['config']['transport'] in ('p2p', 'rccl', 'torch') and head['config']['finite'] is True
  assert head['config']['sharded_vs_unpartitioned_timed_solve_rel_max'] is None or head['config']['sharded_vs_unpartitioned_timed_solve_rel_max'] < 1e-4
  assert any(ln.startswith('{"bench_detail"') for ln in lines[:-1])
