"""Worker for tests/test_sharded_gpu.py::test_blocks_run_sharded_from_the_operator_surface (torch.distributed.run, gloo
bootstrap, every rank on cuda:0): the reference-recorded block fixtures through ODEblock.forward with opt['gnpde_shard'] set --
the block partitions its graph over the ranks, every rank integrates its rows with the native sharded solver and gets the
whole state back.  Must reproduce the fixture (reference output) like the single-GPU block does."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gnpde_amd as G  # noqa: E402
from helpers import Fixture, Data, parity  # noqa: E402

BLOCKS = {'constant': G.ConstantODEblock, 'attention': G.AttODEblock}
FUNCS = {'transformer': G.ODEFuncTransformerAtt, 'laplacian': G.LaplacianODEFunc, 'GAT': G.ODEFuncAtt}


def main():
  out_path, names = sys.argv[1], sys.argv[2].split(',')
  dist.init_process_group('gloo')
  rank, world = dist.get_rank(), dist.get_world_size()
  dev = torch.device('cuda:0')
  torch.cuda.set_device(dev)
  res = {}
  for name in names:
    if name.startswith('selfcheck:') or name.startswith('selfcheck_rw:') or name.startswith('selfcheck_dopri5:'):
      # a FUNCTION fixture (reference parameters, state and graph) integrated by the constant block: the partitioned solve against the
      # unpartitioned solve of this package (which the reference-recorded fixtures pin) -- for functions without a recorded block solve
      fx = Fixture(name.split(':', 1)[1])
      x = fx.t('x', dev)
      opt = dict(fx.opt, block='constant', method='rk4', time=2.3, step_size=1.0)
      if name.startswith('selfcheck_dopri5:'):
        # an adaptive solve of a function whose normaliser is NOT row-local: the device controller over the exchange engine in its
        # general mode (gnpde_dopri5_create_sharded over gnpde_sharded_solver_set_general)
        opt.update(method='dopri5', tol_scale=200.0, time=1.7)
      edge_attr = None
      if name.startswith('selfcheck_rw:'):
        # opt['reweight_attention'] with a weighted edge list (reference src/function_transformer_attention.py:208-209: the scores
        # times the edge weights; the one-GPU path is pinned by the reference-recorded `layer_reweight` fixture)
        opt['reweight_attention'] = True
        E = fx.t('edge_index').shape[1]
        edge_attr = (0.5 + torch.rand(E, generator=torch.Generator().manual_seed(17))).to(dev)
      block = G.ConstantODEblock(FUNCS[opt['function']], [], opt, Data(x, fx.t('edge_index', dev), edge_attr), dev,
                                 t=torch.tensor([0, opt['time']])).to(dev)
      missing, unexpected = block.load_state_dict({'odefunc.' + k: v for k, v in fx.params.items()}, strict=False)
      assert not unexpected and all(k.startswith('reg_odefunc.') for k in missing), (missing, unexpected)
      block.eval()
      with torch.no_grad():
        block.set_x0(x)
        z_one = block(x)                                     # no request: one GPU
        nfe_one = block.odefunc.nfe
        block.odefunc.opt = dict(block.odefunc.opt, gnpde_shard=1)
        block.odefunc.nfe = 0
        block.set_x0(x)
        z = block(x)
        nfe = block.odefunc.nfe
        block.set_x0(x)
        z2 = block(x)
      e_inf, e_2 = parity(z, z_one)
      ent = next(iter(block.odefunc._shard_state.values()))
      sh = ent['shard']
      mine = z.cpu()
      allz = [torch.empty_like(mine) for _ in range(world)]
      dist.all_gather(allz, mine)
      res[name] = dict(rel_max=e_inf, rel_l2=e_2, nfe=nfe, ref_nfe=nfe_one, replay_equal=bool(torch.equal(z, z2)),
                       ranks_agree=all(torch.equal(a, mine) for a in allz), own_rows=sh.n_own, halo_rows=sh.n_halo, world=world,
                       moved=float((z_one - x).abs().max()), solvers=sorted(type(v).__name__ for v in ent['solvers'].values()))
      dist.barrier()
      for e in block.odefunc._shard_state.values():
        e['close']()
      block.odefunc._shard_state.clear()
      continue
    fx = Fixture(name)
    x = fx.t('x', dev)
    opt = dict(fx.opt, gnpde_shard=1)
    block = BLOCKS[opt['block']](FUNCS[opt['function']], [], opt, Data(x, fx.t('edge_index', dev)), dev,
                                 t=torch.tensor([0, opt['time']])).to(dev)
    block.load_state_dict(fx.params, strict=True)
    block.eval()
    block.set_x0(x)
    with torch.no_grad():
      z = block(x)
      nfe = block.odefunc.nfe
      block.set_x0(x)
      z2 = block(x)                      # cached partition, solver and graph
    e_inf, e_2 = parity(z, fx.t('z'))
    ent = next(iter(block.odefunc._shard_state.values()))
    sh = ent['shard']
    # every rank holds the same full result
    mine = z.cpu()
    allz = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allz, mine)
    res[name] = dict(rel_max=e_inf, rel_l2=e_2, nfe=nfe, ref_nfe=int(fx.arr['nfe']), replay_equal=bool(torch.equal(z, z2)),
                     ranks_agree=all(torch.equal(a, mine) for a in allz), own_rows=sh.n_own, halo_rows=sh.n_halo,
                     world=world, solvers=sorted(type(v).__name__ for v in ent['solvers'].values()),
                     syncs=(getattr(block.odefunc, '_dopri5_stats', None) or {}).get('syncs'),
                     trials=sum((getattr(block.odefunc, '_dopri5_stats', None) or {}).get(k, 0) for k in ('accepted', 'rejected')))
    dist.barrier()
    for e in block.odefunc._shard_state.values():
      e['close']()
    block.odefunc._shard_state.clear()
  if rank == 0:
    json.dump(res, open(out_path, 'w'))
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
