"""Worker for tests/test_sharded_gpu.py::test_ranks_sharing_one_gpu (launched by torch.distributed.run, gloo for the
host-side bootstrap, EVERY rank on cuda:0).

Each rank builds its shard of a real P-way partition, maps the other ranks' stage buffers through the IPC handles
(distributed.P2PContext) and runs the native row-partitioned solver with the P2P transport: boundary rows are pushed
into the peers' halo regions and awaited through epoch flags inside each rank's hipGraph.  Rank 0 gathers the owned rows,
undoes the partition permutation and compares with the unpartitioned CPU oracle solve."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gnpde_amd as G  # noqa: E402
from gnpde_amd import distributed as D  # noqa: E402
from oracle import restate as R  # noqa: E402
from helpers import random_graph, parity  # noqa: E402


def main():
  out_path, kind, method, T = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
  dist.init_process_group('gloo')
  rank, world = dist.get_rank(), dist.get_world_size()
  dev = torch.device('cuda:0')
  torch.cuda.set_device(dev)
  n, d, A, h = 5000, 128, 16, 4
  # (column normalisation gives a hub ROW the weight sum of all its columns -- 1200 / 8 here: the explicit solve would diverge, in
  #  the oracle as well; the attention_norm_idx = 1 cases run on the graph without hubs)
  ei = random_graph(n, 8, seed=21, hubs=0 if 'n1' in kind else 2, hub_deg=1200)
  g = torch.Generator().manual_seed(22)
  x = torch.randn(n, d, generator=g)
  params = dict(Wq=torch.randn(A, d, generator=g) / d ** 0.5, Wk=torch.randn(A, d, generator=g) / d ** 0.5,
                bq=torch.randn(A, generator=g) * 0.1, bk=torch.randn(A, generator=g) * 0.1, heads=h)
  alpha, beta = torch.tensor(0.3), torch.tensor(0.2)
  plan = D.PartitionPlan(ei, n, world)
  sh = plan.shard(rank)
  general = kind.startswith('transformer_')          # transformer_n1 / transformer_sqp / transformer_sqp_n1
  norm_idx, square_plus = int('n1' in kind), 'sqp' in kind
  # score functions other than the scaled dot product (reference src/function_transformer_attention.py:193-206), row softmax
  att_type = kind if kind in ('cosine_sim', 'pearson', 'exp_kernel') else 'scaled_dot'
  att_kw = dict(attention_type=att_type)
  if att_type == 'exp_kernel':
    params.update(output_var=torch.tensor([1.3]), lengthscale=torch.tensor([2.1]))
    att_kw.update(output_var=params['output_var'], lengthscale=params['lengthscale'])
  if att_type != 'scaled_dot':
    params['att_type'] = att_type
  if kind == 'laplacian':
    _, w = G.get_rw_adj(ei, None, norm_dim=0, fill_value=0.0, num_nodes=n, dtype=torch.float32)
    p = dict(edge_weight=w[sh.edge_ids])
  else:
    p = dict(params, norm_idx=norm_idx, square_plus=square_plus)
  gat = kind.startswith('gat')                        # gat (softmax over rows, in-graph P2P solver) / gat_n1 (over columns: general path)
  if gat:
    # reference src/function_GAT_attention.py: W [d, A], a [2 d_k, 1, 1]; the backend takes the row-major [A, d] copy the layer keeps
    Wg = torch.randn(d, A, generator=g) / d ** 0.5
    ag = torch.randn(2 * (A // h), 1, 1, generator=g) * 0.5
    p = dict(W=Wg.t().contiguous(), a=ag.reshape(-1), heads=h, leaky_slope=0.2, norm_idx=norm_idx)
    general = norm_idx != 0
  be = D.NativeBackend(sh, d, dev, 'gat' if gat else ('transformer' if (general or att_type != 'scaled_dot') else kind), p, alpha, beta, True)
  x_own = D.scatter_rows(x, sh).to(dev)
  res = {}
  if general:
    # normalisers that are not row-local: the Python-driven loop with exchanges between the attention passes
    assert be.general
    with torch.no_grad():
      solver = D.ShardedSolver(sh, be)
      z1 = solver.integrate(x_own, x_own, T, 1.0, method).clone()
      z2 = solver.integrate(x_own, x_own, T, 1.0, method).clone()
    same = torch.equal(z1, z2)
    if not same:
      print('rank %d: repeated solve differs: max |d| %g, finite %s / %s' % (rank, float((z1 - z2).abs().max()), bool(torch.isfinite(z1).all()), bool(torch.isfinite(z2).all())), flush=True)
    full = D.gather_rows_all(z1.cpu(), plan, sh)
    # the same solve with the exchanges between the attention passes INSIDE the per-rank graph (gnpde_sharded_solver_set_general): the
    # same kernels in the same order, the merges peer by peer in rank order -- eager and as a replayed hipGraph
    ctx = D.P2PContext(sh, d, 5)
    native = {}
    with torch.no_grad():
      nat = D.NativeShardedSolver(sh, be, T, 1.0, method, ctx=ctx)
      nat.set_spin_limit(1 << 22)
      n_eager = nat.integrate(x_own, x_own, use_graph=False).clone()
      nat.check()
      n_g1 = nat.integrate(x_own, x_own, use_graph=True).clone()
      n_g2 = nat.integrate(x_own, x_own, use_graph=True).clone()
      nat.check()
    native['native_replay_equal'] = bool(torch.equal(n_g1, n_g2)) and bool(torch.equal(n_eager, n_g1))
    native['native_vs_loop_bitwise'] = bool(torch.equal(n_g1, z1))
    native['native_vs_loop'] = float((n_g1 - z1).abs().max() / z1.abs().max())
    flags = torch.tensor([int(native['native_replay_equal']), int(native['native_vs_loop_bitwise'])])
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    native['native_replay_equal'], native['native_vs_loop_bitwise'] = bool(flags[0]), bool(flags[1])
    worst = torch.tensor([native['native_vs_loop']], dtype=torch.float64)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    native['native_vs_loop'] = float(worst[0])
    nfull = D.gather_rows_all(n_g1.cpu(), plan, sh)
    nat.close()
    ctx.close()
    if rank == 0:
      rhs = lambda t, y: R.rhs_transformer(y, ei, params['Wq'], params['bq'], params['Wk'], params['bk'], h, alpha, beta,  # noqa: E731
                                           x, False, True, norm_idx=norm_idx, square_plus=square_plus)
      if gat:
        rhs = lambda t, y: R.rhs_gat(y, ei, Wg, ag, h, alpha, beta, x, False, True, 0.2, norm_idx)   # noqa: E731
      ref = R.odeint_fixed(rhs, x, T, 1.0, method)
      e_inf, e_2 = parity(full, ref)
      print('general %s: rel_max %g rel_l2 %g' % (kind, e_inf, e_2), flush=True)
      n_inf, n_2 = parity(nfull, ref)
      json.dump(dict({'rel_max': e_inf, 'rel_l2': e_2, 'world': world, 'edge_cut': plan.edge_cut(), 'halo_rows': sh.n_halo,
                      'interior_rows': sh.n_interior, 'own_rows': sh.n_own, 'native_rel_max': n_inf, 'native_rel_l2': n_2}, **native),
                open(out_path, 'w'))
    dist.barrier()
    dist.destroy_process_group()
    assert same, 'rank %d: repeated solve differs' % rank
    return
  if method in ('dopri5', 'adaptive_heun'):
    # adaptive pair on the partitioned graph: host controller over the sharded HIP evaluations, error norm all-reduced; against
    # the restated torchdiffeq 0.2.1 of oracle/shims over the CPU oracle on the whole graph (same evaluations, same state)
    from oracle.shims import install as REF_TORCHDIFFEQ
    rtol, atol = 1e-4, 1e-6
    tt = torch.tensor([0.0, T])
    with torch.no_grad():
      solver = D.ShardedSolver(sh, be)
      z1 = solver.integrate_adaptive(x_own, x_own, tt.to(dev), rtol, atol, n, method=method).clone()
      evals = solver.n_evals
      z2 = solver.integrate_adaptive(x_own, x_own, tt.to(dev), rtol, atol, n, method=method).clone()
    assert torch.equal(z1, z2), 'rank %d: repeated adaptive solve differs' % rank
    full = D.gather_rows_all(z1.cpu(), plan, sh)
    native = {}
    if method in ('dopri5', 'adaptive_heun'):
      # the same solve with the controller on every rank's DEVICE (gnpde_dopri5_create_sharded): trial steps as per-rank hipGraphs, the
      # error norm summed over the ranks inside the stream; the host reads the controller record once per batch of trial steps
      ctx = D.P2PContext(sh, d, 4)
      with torch.no_grad():
        nat = D.NativeShardedDopri5(sh, be, rtol, atol, n, with_source=True, ctx=ctx, pair=method)
        nat.engine.set_spin_limit(1 << 22)
        n1, fin1 = nat.integrate(x_own, x_own, 0.0, T, trials_per_sync=8)
        n1 = n1.clone()
        st1 = nat.stats()
        n2, fin2 = nat.integrate(x_own, x_own, 0.0, T, trials_per_sync=1)
        n2 = n2.clone()
        st2 = nat.stats()
        n3, fin3 = nat.integrate(x_own, x_own, 0.0, T, trials_per_sync=8, max_evals=8)     # the evaluation budget (opt['max_nfe'])
        st3 = nat.stats()
      assert fin1 and fin2 and not fin3, (fin1, fin2, fin3)
      assert torch.equal(n1, n2), 'rank %d: the batch size of the record reads changed the solve' % rank
      mine = torch.tensor([st1['evals'], st1['accepted'], st1['rejected'], st1['launches']], dtype=torch.int64)
      allst = [torch.empty_like(mine) for _ in range(world)]
      dist.all_gather(allst, mine)
      assert all(torch.equal(a, mine) for a in allst), 'rank %d: the ranks took different decisions: %r' % (rank, allst)
      nfull = D.gather_rows_all(n1.cpu(), plan, sh)
      native = dict(native_evals=st1['evals'], native_syncs=st1['syncs'], native_trials=st1['accepted'] + st1['rejected'],
                    native_syncs_per_trial_batch1=st2['syncs'], native_budget_evals=st3['evals'],
                    native_vs_host_controller=float((nfull - full).abs().max() / full.abs().max()))
      nat.close()
      ctx.close()
    if rank == 0:
      calls = [0]

      def rhs(t, y):
        calls[0] += 1
        if kind == 'laplacian':
          return R.rhs_laplacian(y, ei, w, alpha, beta, x, False, True)
        if gat:
          return R.rhs_gat(y, ei, Wg, ag, h, alpha, beta, x, False, True, 0.2, norm_idx)
        return R.rhs_transformer(y, ei, params['Wq'], params['bq'], params['Wk'], params['bk'], h, alpha, beta, x, False, True, **att_kw)
      ref = REF_TORCHDIFFEQ.odeint(rhs, x, tt, method=method, options={}, rtol=rtol, atol=atol)[1]
      e_inf, e_2 = parity(full, ref)
      json.dump({'rel_max': e_inf, 'rel_l2': e_2, 'world': world, 'edge_cut': plan.edge_cut(), 'halo_rows': sh.n_halo,
                 'interior_rows': sh.n_interior, 'own_rows': sh.n_own, 'evals': evals, 'ref_evals': calls[0]}, open(out_path, 'w'))
      if native:
        n_inf, n_2 = parity(nfull, ref)
        r = json.load(open(out_path))
        r.update(native, native_rel_max=n_inf, native_rel_l2=n_2)
        json.dump(r, open(out_path, 'w'))
    dist.barrier()
    dist.destroy_process_group()
    return
  ctx = D.P2PContext(sh, d, 4)
  with torch.no_grad():
    for use_graph in (False, True):
      solver = D.NativeShardedSolver(sh, be, T, 1.0, method, ctx=ctx)
      solver.set_spin_limit(1 << 22)
      z1 = solver.integrate(x_own, x_own, use_graph=use_graph).clone()
      z2 = solver.integrate(x_own, x_own, use_graph=use_graph).clone()   # second solve: epochs keep counting, graph replays
      timed_out, epochs = solver.status()
      assert not timed_out, 'rank %d: a peer never published its epoch' % rank
      assert torch.equal(z1, z2), 'rank %d: repeated solve differs' % rank
      res[use_graph] = z1
      dist.barrier()
      solver.close()
    assert torch.equal(res[False], res[True]), 'rank %d: hipGraph replay differs from eager launches' % rank
    # boundary rows in 3 row ranges, each range's rows of the next stage input pushed right behind it: the same arithmetic in
    # the same order, so bit-identical to the single boundary pass.  A NEW input, chunked solver first: the halo regions
    # still hold the rows of the solves above, so a send slot that is not pushed (or pushed too late) changes the result.
    xin = {False: x_own * 0.7 + 0.1, True: x_own * 0.4 - 0.2}
    chunked = {}
    for use_graph in (False, True):
      solver = D.NativeShardedSolver(sh, be, T, 1.0, method, ctx=ctx, boundary_chunks=3)
      assert solver.boundary_chunks == 3, solver.boundary_chunks
      ranges = [(c[0], c[1]) for c in solver.chunks]
      assert ranges[0][0] == sh.n_interior and ranges[-1][1] == sh.n_own and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
      solver.set_spin_limit(1 << 22)
      c1 = solver.integrate(xin[use_graph], x_own, use_graph=use_graph).clone()
      c2 = solver.integrate(xin[use_graph], x_own, use_graph=use_graph).clone()
      timed_out, _ = solver.status()
      assert not timed_out, 'rank %d: a peer never published its epoch (chunked pushes)' % rank
      assert torch.equal(c1, c2), 'rank %d: repeated chunked solve differs' % rank
      chunked[use_graph] = c1
      dist.barrier()
      solver.close()
    solver = D.NativeShardedSolver(sh, be, T, 1.0, method, ctx=ctx)
    for use_graph in (False, True):
      plain = solver.integrate(xin[use_graph], x_own).clone()
      solver.check()
      assert torch.equal(chunked[use_graph], plain), 'rank %d: chunked boundary pass (graph=%s) differs: max |d| %g' % (
        rank, use_graph, float((chunked[use_graph] - plain).abs().max()))
      assert not torch.equal(plain, res[True])
    dist.barrier()
    solver.close()
  full = D.gather_rows_all(res[True].cpu(), plan, sh)
  if rank == 0:
    if kind == 'laplacian':
      rhs = lambda t, y: R.rhs_laplacian(y, ei, w, alpha, beta, x, False, True)   # noqa: E731
    elif gat:
      rhs = lambda t, y: R.rhs_gat(y, ei, Wg, ag, h, alpha, beta, x, False, True, 0.2, norm_idx)   # noqa: E731
    else:
      rhs = lambda t, y: R.rhs_transformer(y, ei, params['Wq'], params['bq'], params['Wk'], params['bk'], h, alpha, beta,  # noqa: E731
                                           x, False, True, **att_kw)
    ref = R.odeint_fixed(rhs, x, T, 1.0, method)
    e_inf, e_2 = parity(full, ref)
    json.dump({'rel_max': e_inf, 'rel_l2': e_2, 'world': world, 'edge_cut': plan.edge_cut(), 'halo_rows': sh.n_halo,
               'interior_rows': sh.n_interior, 'own_rows': sh.n_own}, open(out_path, 'w'))
  dist.barrier()
  ctx.close()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
