#!/usr/bin/env python
"""ODE steps/sec of the full-graph GRAND-nl diffusion solve (BASELINE.json metric).

One "step" = one rk4 (3/8 rule) solver step = 4 evaluations of f(t,x) = alpha (A(x) x - x) + beta x0 on
the synthetic ogbn-arxiv-shaped graph (169,343 nodes, ~2.48 M edges incl. self-loops, d = 128,
attention_dim 16 / 4 heads, softmax over rows, add_source) -- BASELINE.json configs[2].  The K timed
steps are ONE launch of the hipGraph-captured native solver (T = K, step_size = 1), state resident in HBM;
the launch is repeated `--replays` times (each bracketed by a device synchronisation) and the MEDIAN is reported.

  python bench.py --gpus 1 --steps 100 --warmup 10
  python bench.py --graph rmat --steps 8 --warmup 1        (BASELINE configs[4] shape on ONE GPU: 2 M nodes, d = 256)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (row-partitioned, RCCL halo)

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the CSR aggregation
spmm_rows_kernel / spmm_pair_kernel): algorithmic bytes of DESIGN.md section "Kernels" divided by its average launch duration,
measured here with HIP events around a captured graph of the four rk4-stage variants the solver runs.  `roofline.frac` is
against HBM's 8 TB/s when the gathered table exceeds the Infinity Cache (R-MAT) and against the rate of a perfectly balanced
row gather from the same table measured in the same run (`roofline.ceiling`, gnpde_gather_ceiling) when it is cache-resident
(ogbn-arxiv); `roofline.secondary` times the projection and the row attention.
`cpu_baseline` times the CPU oracle (the reference's op sequence) on this host's cores on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6300.0  # what a streaming copy reaches on this part (same guide, chip-level parameters): SURVEY 8d asks for both


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--graph', default='arxiv', choices=['arxiv', 'cora', 'rmat'])
  ap.add_argument('--scale', type=float, default=1.0, help='shrink the graph (debug only; invalidates the metric)')
  ap.add_argument('--att-dim', type=int, default=None)
  ap.add_argument('--heads', type=int, default=None)
  ap.add_argument('--function', default='transformer', choices=['transformer', 'laplacian'])
  ap.add_argument('--no-graph', action='store_true', help='launch the solver eagerly instead of via hipGraph')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-roofline-probe', action='store_true',
                  help='skip the separate timing launches of the dominant kernel (profiling runs: the rocprof summary then '
                       'holds the solver\'s own launches only); `roofline` is null in the line')
  ap.add_argument('--cpu-evals', type=int, default=None,
                  help='full-size evaluations of f timed on the host (default 6; the rmat shape needs ~100 GB of '
                       'host temporaries per evaluation and is skipped unless a count is given)')
  ap.add_argument('--replays', type=int, default=5, help='timed launches of the K-step solve (median reported)')
  ap.add_argument('--seed', type=int, default=0)
  ap.add_argument('--norm-idx', type=int, default=0, choices=[0, 1], help='attention_norm_idx (1: softmax over columns, general 3-pass path)')
  ap.add_argument('--square-plus', action='store_true', help='squareplus normalisation (Cora best_params)')
  ap.add_argument('--early-stop', action='store_true',
                  help='run the test-time early-stopping evaluator (40-class decoder, arg-max, split accuracies) after '
                       'every step inside the hipGraph, as the reference does at evaluation time (not the headline metric)')
  return ap.parse_args()


def build_opt(cfg, args):
  return dict(heads=args.heads or cfg['heads'], attention_dim=args.att_dim or cfg['att_dim'],
              attention_type='scaled_dot', attention_norm_idx=args.norm_idx, square_plus=args.square_plus,
              reweight_attention=False,
              beltrami=False, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9, add_source=True,
              no_alpha_sigmoid=False, mix_features=False, hidden_dim=cfg['d'], augment=False, adjoint=False,
              tol_scale=1.0, data_norm='rw', method='rk4', step_size=1.0, max_iters=100, block='constant',
              function=args.function, time=float(args.steps))


class _Data(object):
  pass


GRAPH_NAMES = {'arxiv': 'ogbn-arxiv', 'cora': 'Cora', 'rmat': 'RMAT-2M'}


def metric_name(graph, d, world=1):
  """BASELINE.json's metric with the graph that was ACTUALLY run."""
  return 'ODE steps/sec (full-graph diffusion), %s d=%d rk4' % (GRAPH_NAMES.get(graph, graph), d)


def workload_name(graph, function, K):
  shape = {'arxiv': 'synthetic ogbn-arxiv-shaped graph (power-law degrees, 40 communities, shuffled ids)',
           'cora': 'synthetic Cora-shaped graph (uniform random)',
           'rmat': 'synthetic R-MAT graph (Graph500 parameters, 2^21 nodes, 40 M generated edges, symmetrised)'}[graph]
  return '%s, GRAND-%s add_source, rk4 3/8-rule, step_size 1, T=%d, hipGraph-captured solver' % (
    shape, 'nl scaled_dot softmax attention' if function == 'transformer' else 'l', K)


def host_info():
  model = ''
  try:
    for line in open('/proc/cpuinfo'):
      if line.startswith('model name'):
        model = line.split(':', 1)[1].strip()
        break
  except OSError:
    pass
  return {'host_cores': os.cpu_count() or 1, 'cpu_model': model}


def make_block(G, opt, ei, n, x, dev, T, seed):
  data = _Data()
  data.x, data.edge_index, data.edge_attr, data.num_nodes = x, ei, None, n
  fcls = G.ODEFuncTransformerAtt if opt['function'] == 'transformer' else G.LaplacianODEFunc
  block = G.ConstantODEblock(fcls, [], dict(opt, time=T), data, dev, t=torch.tensor([0, T])).to(dev)
  g = torch.Generator().manual_seed(seed + 1)
  with torch.no_grad():
    for name, p in block.named_parameters():
      if p.dim() >= 2 and 'multihead_att_layer' in name:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))  # Q/K ~ N(0, 1/d)
      elif name.endswith('.bias'):
        p.zero_()
    for f in (block.odefunc, block.reg_odefunc.odefunc):
      f.alpha_train.fill_(0.0)   # sigmoid -> 0.5
      f.beta_train.fill_(0.1)
  block.eval()
  return block


def solver_graph(f, x):
  """The CSR the fused solves of `f` launch on: the locality view of its graph when the solver uses one (same operator, nodes
  relabelled part by part), else the graph as given."""
  view = f._locality_view(x) if hasattr(f, '_locality_view') else None
  return (view.graph if view is not None else f._graph(x)), view


def dominant_kernel_time(G, block, x, reps=10):
  """Average duration of one launch of the dominant kernel over the 4 rk4 stage epilogues the solver runs,
  HIP events on the launch stream (torch's current stream, which is the stream the C ABI is handed).
  GRAND-nl: the one-pass attention + aggregation kernel; GRAND-l: the CSR aggregation kernel."""
  import ctypes
  from gnpde_amd import ops, _lib
  f = block.odefunc
  graph, _ = solver_graph(f, x)
  dev = x.device
  bufs = [torch.randn_like(x) for _ in range(7)]
  y, k1, k2, k3, ua, ub, x0 = bufs
  alpha = ops._scalar_dev(f.alpha_train, x)
  beta = ops._scalar_dev(f.beta_train, x)
  # the four stage variants the solver actually runs (compact rk4: stage states from stage inputs)
  stages = [dict(stage=_lib.STAGE_RK1C, out_y=ua, u=y),
            dict(stage=_lib.STAGE_RK2C, y=y, out_y=ub, u=ua),
            dict(stage=_lib.STAGE_RK3C, k1=ua, out_y=k1, u=ub),
            dict(stage=_lib.STAGE_RK4C, y=y, k1=ub, out_y=y, u=k1)]
  fused = False
  if hasattr(f, 'multihead_att_layer') and os.environ.get('GNPDE_ONE_PASS', '0') == '1':
    desc = f._descriptor(x)
    fused = _lib.lib().gnpde_attn_rhs_fused_supported(ctypes.byref(desc.struct.att), x.shape[1], x.stride(0)) == 1
  if fused:
    wqk, bqk = f.multihead_att_layer.qk_weights()
    att = desc.struct.att
    name = 'attn_rhs_fused_kernel (one-pass projection + edge softmax + aggregation + rk4 stage)'

    def launch(u, kw):
      ops.attn_rhs_fused(graph, att, wqk, bqk, u, alpha, beta, x0, True, dt=1.0, **kw)
  else:
    w = torch.rand(max(graph.e, 1), device=dev) / 16
    name = ('CSR aggregation + fused epilogue / rk4 stage: spmm_pair_kernel (d = 68..128, mostly short rows: two rows per '
            'wavefront) or spmm_wide_kernel (16-byte lanes, d = 68..256; spmm_rows_kernel otherwise) + spmm_long_reduce_kernel '
            '(hub-row fold)')

    def launch(u, kw):
      ops.spmm_rhs(graph, w, u, alpha, beta, x0, True, dt=1.0, **kw)

  def once():
    for st in stages:
      kw = dict(st)
      u = kw.pop('u')
      launch(u, kw)
  return timed_replay(once, reps) / 4, graph, name, fused


def timed_replay(fn, reps, replays=3):
  """Average device time of one call of `fn`: `reps` calls captured into one graph on torch's capture stream (the stream the
  C ABI is handed), replayed `replays` times, HIP events around each replay, best replay reported -- kernel time back to back,
  no host launch gaps.  Falls back to eager launches between two events if the capture fails."""
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      for _ in range(reps):
        fn()
    best = None
    for _ in range(replays):
      torch.cuda.synchronize()
      e0.record()
      g.replay()
      e1.record()
      torch.cuda.synchronize()
      t = e0.elapsed_time(e1)
      best = t if best is None or t < best else best
    return best * 1e-3 / reps
  except Exception:
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def gather_ceiling(G, x, n, E, graph=None):
  """Rate of a perfectly balanced gather of whole state rows from the SAME table (gnpde_gather_ceiling), no weights, no
  epilogue streams, no degree skew.  With `graph`: the graph's OWN column ids in CSR order, k = floor(E / n) consecutive ids per
  output row -- the same references in the same order as the aggregation makes them, so the L2 hits that the graph's hot
  columns earn are in the ceiling too; without: k uniformly random rows per output row (no reuse at all).  Bytes by the same
  gather model as the aggregation's (ids + gathered rows + one row written)."""
  from gnpde_amd import _lib
  d, ld = x.shape[1], x.stride(0)
  if d % 4 != 0 or d > 256 or ld % 4 != 0:
    return None
  L = _lib.lib()
  out = torch.empty_like(x)
  if graph is not None and E >= n:
    k = max(1, E // n)
    idx = graph.t['colidx'][: n * k].contiguous()
    what = ('gnpde_gather_ceiling: out[i] = sum of rows colidx[i k .. i k + k) of the same [n, d] table -- the graph\'s own column '
            'ids in CSR order, k = floor(E / n) per output row (same references, same order, perfectly balanced)')
  else:
    k = max(1, int(round(E / float(n))))
    gen = torch.Generator(device=x.device).manual_seed(1234)
    idx = torch.randint(0, n, (n * k,), device=x.device, dtype=torch.int32, generator=gen)
    what = 'gnpde_gather_ceiling: out[i] = sum of k uniformly random rows of the same [n, d] table'
  best = None
  for variant in (0, 1):      # ids loaded per lane / one coalesced id load + shuffle: the faster one is the ceiling
    def call():
      _lib.check(L.gnpde_gather_ceiling(_lib.ptr(x), n, d, ld, _lib.ptr(idx), k, _lib.ptr(out), n, variant, _lib.stream_of(x)))
    t = timed_replay(call, 8)
    if best is None or t < best[0]:
      best = (t, variant)
  t, variant = best
  nbytes = n * k * (4 + 4 * d) + n * 4 * d
  return {'row_gather_gbs': round(n * k * 4 * d / t / 1e9, 1), 'gather_model_gbs': round(nbytes / t / 1e9, 1),
          'avg_launch_us': round(t * 1e6, 2), 'rows_gathered_per_output_row': k, 'variant': variant,
          'gathered_row_bytes_per_launch': n * k * 4 * d, 'bytes_per_launch': nbytes,
          'what': what + '; 16-byte lanes, one wavefront per workgroup, no weights / epilogue operands / degree skew / hub rows; '
                         'measured in this run; row_gather_gbs = bytes of the gathered rows alone / time'}


def secondary_kernels(G, block, x, E, n, ceiling):
  """The other launches of one evaluation of f (GRAND-nl): the q||k projection and the row attention (two launches), timed in
  this run on the solver's own operands; bytes by the model of DESIGN.md section 4."""
  from gnpde_amd import ops, _lib
  f = block.odefunc
  if not hasattr(f, 'multihead_att_layer'):
    return []
  lay = f.multihead_att_layer
  graph, _ = solver_graph(f, x)
  wqk, bqk = lay.qk_weights()
  A, h = lay.attention_dim, lay.h
  d = x.shape[1]
  qk = ops.linear(x, wqk, bqk)
  t_lin = timed_replay(lambda: ops.linear(x, wqk, bqk, out=qk), 16)
  st = ops.attention_struct(_lib.ATT_TYPES[f.opt['attention_type']], h, A, f.opt['attention_norm_idx'], f.opt['square_plus'],
                            q=qk, k=qk[:, A:], ldqk=2 * A)
  t_att = timed_replay(lambda: ops.edge_attention(graph, st, True, False, False, like=x), 16)
  b_lin = n * (4 * d + 4 * 2 * A)
  b_att = E * (4 + 4 * A + 4) + n * (16 + 4 * A)
  out = [{'kernel': 'row attention: scores + softmax over the row + head mean (row_attention_sd_kernel, one launch per degree '
                    'class with the hub phases riding)', 'bytes': b_att, 'avg_us': round(t_att * 1e6, 2),
          'gbs': round(b_att / t_att / 1e9, 1)},
         {'kernel': 'q||k projection [n,d] x [d,2A] on the fp32 MFMA (linear_persistent_kernel / linear_lds_kernel)', 'bytes': b_lin,
          'avg_us': round(t_lin * 1e6, 2), 'gbs': round(b_lin / t_lin / 1e9, 1)}]
  return out


def source_sha16(rel):
  import hashlib
  try:
    return hashlib.sha256(open(os.path.join(ROOT, rel), 'rb').read()).hexdigest()[:16]
  except OSError:
    return None


def cpu_baseline(block, x_cpu, evals):
  """Reference op sequence (oracle) on the host cores: a bounded number of full-size evaluations of f."""
  from oracle import restate as R
  f = block.odefunc
  cpu = lambda t: t.detach().cpu()
  edge = cpu(f.edge_index)
  if hasattr(f, 'multihead_att_layer'):
    lay = f.multihead_att_layer
    args = (cpu(lay.Q.weight), cpu(lay.Q.bias), cpu(lay.K.weight), cpu(lay.K.bias), lay.h)
    rhs = lambda y: R.rhs_transformer(y, edge, *args, cpu(f.alpha_train), cpu(f.beta_train), x_cpu, False, True)
  else:
    w = cpu(f.edge_weight)
    rhs = lambda y: R.rhs_laplacian(y, edge, w, cpu(f.alpha_train), cpu(f.beta_train), x_cpu, False, True)
  # torch's CPU scatter/gather ops do not scale to every core of a large host: try a few thread counts
  # (one evaluation each) and time the sample at the best one, so the baseline is not handicapped
  ncpu = os.cpu_count() or 1
  cands = sorted({c for c in (ncpu, 64, 32, 16, 8) if c <= ncpu}, reverse=True)
  best_t, best_c = None, ncpu
  with torch.no_grad():
    torch.set_num_threads(cands[0])
    out = rhs(x_cpu)  # warm-up, also the parity reference
    for c in cands:
      torch.set_num_threads(c)
      t0 = time.perf_counter()
      rhs(x_cpu)
      t = time.perf_counter() - t0
      if best_t is None or t < best_t:
        best_t, best_c = t, c
    torch.set_num_threads(best_c)
    t0 = time.perf_counter()
    for _ in range(evals):
      rhs(x_cpu)
    dt = (time.perf_counter() - t0) / evals
  return dt, out


def main():
  args = parse()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus != world:
    if world == 1 and args.gpus > 1:
      raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
  import gnpde_amd as G
  if os.environ.get('GNPDE_ONE_PASS', '0') == '1':
    G.ops.tune(G._lib.TUNE_ONE_PASS, 1)
  xcd_knob = 0
  for kv in filter(None, os.environ.get('GNPDE_TUNE', '').split(',')):     # A/B knobs, e.g. GNPDE_TUNE=6=2 (separate kernels)
    key, val = kv.split('=')
    G.ops.tune(int(key), int(val))
    if int(key) == G._lib.TUNE_XCD_ROWS:
      xcd_knob = int(val)
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs a HIP device: there is no CPU fallback for the measured path')
  dev = torch.device('cuda', 0 if os.environ.get('GNPDE_RANKS_SHARE_DEVICE', '0') == '1' else local_rank)
  torch.cuda.set_device(dev)
  if world > 1 or os.environ.get('GNPDE_FORCE_SHARDED', '0') == '1':   # (the env switch exercises the sharded driver on one GPU)
    from gnpde_amd import distributed as D
    return D.bench_main(args, rank, world, dev)

  cfg = G.synthetic.CONFIGS[args.graph]
  ei_cpu, n = G.synthetic.make_graph(args.graph, seed=args.seed, scale=args.scale)
  opt = build_opt(cfg, args)
  d = cfg['d']
  x_cpu = torch.randn(n, d, generator=torch.Generator().manual_seed(args.seed))
  x = x_cpu.to(dev)
  ei = ei_cpu.to(dev)
  K, W = args.steps, args.warmup
  use_graph = not args.no_graph

  main_block = make_block(G, opt, ei, n, x, dev, float(K), args.seed)
  main_block.set_x0(x)
  early = None
  if args.early_stop:
    gen = torch.Generator().manual_seed(args.seed + 1)
    role = torch.rand(n, generator=gen)

    class _Split(object):
      pass
    split = _Split()
    split.y = torch.randint(0, 40, (n,), generator=gen).to(dev)
    split.train_mask, split.val_mask, split.test_mask = (role < 0.54).to(dev), ((role >= 0.54) & (role < 0.71)).to(dev), (role >= 0.71).to(dev)
    early = G.EarlyStopInt(float(K), dict(opt, earlystopxT=1, max_test_steps=10 ** 6, dataset='ogbn-arxiv'), dev)
    early.data = split
    early.m2_weight = (torch.randn(40, d, generator=gen) / d ** 0.5).to(dev)
    early.m2_bias = torch.zeros(40, device=dev)
    main_block.test_integrator = early
  with torch.no_grad():
    if W > 0:
      warm_block = make_block(G, opt, ei, n, x, dev, float(W), args.seed)
      warm_block.set_x0(x)
      if not use_graph:
        import functools
        warm_block.test_integrator = functools.partial(G.odeint, use_graph=False)
      warm_block(x)                       # W untimed warm-up steps
      del warm_block
    if not use_graph:
      import functools
      main_block.test_integrator = functools.partial(G.odeint, use_graph=False)
    main_block(x)                          # untimed: instantiates the K-step hipGraph
    times = []
    for _ in range(max(args.replays, 1)):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      z = main_block(x)                    # EXACTLY K steps
      torch.cuda.synchronize()
      times.append(time.perf_counter() - t0)
  assert torch.isfinite(z).all()
  elapsed = sorted(times)[len(times) // 2]  # median replay
  steps_per_s = K / elapsed
  f = main_block.odefunc
  E = int(f.edge_index.shape[1])
  A, h = opt['attention_dim'], opt['heads']

  # roofline of the dominant kernel (B_spmm = E (4 + 4 + 4d) + N (4 + 8d) + 4dN with add_source)
  if args.no_roofline_probe:
    print(json.dumps({'metric': metric_name(args.graph, d), 'value': round(steps_per_s, 3), 'unit': 'steps/s', 'n_gpus': 1,
                      'steps': K, 'warmup': W, 'ms_per_step': round(1e3 * elapsed / K, 4), 'higher_is_better': True,
                      'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                      'config': {'workload': workload_name(args.graph, args.function, K), 'graph': args.graph, 'nodes': n,
                                 'edges_with_self_loops': E, 'd': d, 'hipgraph': use_graph},
                      'roofline': None, 'cpu_baseline': None, 'note': 'profiling run: --no-roofline-probe'}))
    return
  t_spmm, graph, kname, fused = dominant_kernel_time(G, main_block, x)
  bytes_l = E * (8 + 4 * d) + n * (4 + 8 * d) + 4 * d * n                      # SURVEY.md 8d, B_l + source
  bytes_nl = E * (4 + 4 * A + 4 * d) + n * (4 + 12 * A + 12 * d) + 4 * d * n   # SURVEY.md 8d, B_nl + source
  bytes_spmm = bytes_nl if fused else bytes_l
  achieved = bytes_spmm / t_spmm / 1e9
  ceiling = gather_ceiling(G, x, n, E, graph)          # the graph's own references, balanced
  ceiling_uniform = gather_ceiling(G, x, n, E)       # no reuse at all (what round 3's first lines were quoted against)
  try:
    secondary = [] if fused else secondary_kernels(G, main_block, x, E, n, ceiling)
  except Exception as exc:   # (a probe that cannot run must not cost the line)
    secondary = [{'error': repr(exc)[:200]}]
  try:
    # counter traffic of the same launches from the PMC record of this shape (profiles/hbm_traffic.json, `detail`): what the
    # kernels really moved next to what the model says they need
    det = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))['detail']['%s_d%d_spmm' % (args.graph, d)]
    for ent, pat in zip(secondary, ('row_attention_sd_kernel', 'linear_')):
      recs = [v for k, v in det['kernels'].items() if pat in k and 'bytes_per_launch' in v]
      if recs and 'avg_us' in ent:
        tb = sum(r['bytes_per_launch'] for r in recs) if pat.startswith('row') else max(r['bytes_per_launch'] for r in recs)
        ent['traffic'] = round(tb)
        ent['traffic_gbs'] = round(tb / (ent['avg_us'] * 1e-6) / 1e9, 1)
        ent['l2_hit_rate'] = [r.get('l2_hit_rate') for r in recs]
        ent['traffic_commit'] = det.get('commit')
  except Exception:   # noqa: BLE001 -- no record for this shape
    pass
  traffic = None
  tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
  if os.path.exists(tpath):
    try:
      traffic = json.load(open(tpath)).get('%s_d%d_%s' % (args.graph, d, 'fused' if fused else 'spmm'))
    except Exception:
      traffic = None
  bytes_eval = bytes_nl if args.function == 'transformer' else bytes_l
  # compulsory DRAM bytes of one aggregation launch with perfect reuse of gathered rows (SURVEY 8d B_min without the
  # projection): colidx + w + one read of u + the per-row streams (x0 in, out)
  dram_floor = E * 8 + n * (4 + 12 * d)
  state_mb = n * d * 4 / 2 ** 20
  resident = bool(state_mb < 256)
  # What `achieved`, `peak` and `frac` are.
  #  * Table >> Infinity Cache (R-MAT): every gathered row comes from DRAM.  achieved = gather-model (algorithmic) bytes of one
  #    launch / its duration, peak = HBM's 8 TB/s (the task's definition).
  #  * Table inside the 256-MiB Infinity Cache (ogbn-arxiv, 83 MiB): the gathered rows come from L2 / MALL and HBM's rate does
  #    not bound them (a fraction of it can exceed 1 and means nothing).  The bound is the rate at which the memory system
  #    delivers randomly addressed rows of this table: achieved = bytes of the E gathered rows alone / launch duration,
  #    peak = the same quantity of a perfectly balanced gather of THE SAME column ids in CSR order measured in this run
  #    (gnpde_gather_ceiling: no weights, no epilogue streams, no skew) -- the aggregation also streams 3 N d-float operands on
  #    average on top of its gathers (x0 in and one row out in every stage, y and k1 in two of four: 11 us per 87-MB stream at
  #    this shape, tools/agg_streams.py), so it cannot reach that rate; the gap IS mostly those streams.
  row_gather = E * 4 * d / t_spmm / 1e9
  if resident and ceiling is not None:
    bound, ach, peak = 'l2-miss/MALL', row_gather, ceiling['row_gather_gbs']
    peak_src = 'measured in this run: row-gather rate of gnpde_gather_ceiling on the same table (see `ceiling`)'
    ach_is = ('bytes of the gathered neighbour rows alone (E * 4 d) of one aggregation launch / its average duration over the four '
              'rk4 stage variants, timed in this run (HIP events around a captured graph of the launches); the gather-model rate '
              'with every stream counted is `gather_model_gbs`')
  else:
    bound, ach, peak, peak_src = 'hbm', achieved, HBM_PEAK_GBS, 'HBM3E peak, MI355X_MICROARCH.md'
    ach_is = ('gather-model (algorithmic) bytes of one aggregation launch / its average duration over the four rk4 stage variants, '
              'timed in this run (HIP events around a captured graph of the launches)')
  out = {
    'metric': metric_name(args.graph, d),
    'value': round(steps_per_s, 3), 'unit': 'steps/s', 'n_gpus': 1, 'steps': K, 'warmup': W,
    'ms_per_step': round(1e3 * elapsed / K, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
    'dtype': 'f32', 'data': 'synthetic',
    'timing': {'replays': len(times), 'statistic': 'median', 'ms_per_step_min': round(1e3 * min(times) / K, 4),
               'ms_per_step_max': round(1e3 * max(times) / K, 4)},
    'config': {'workload': workload_name(args.graph, args.function, K),
               'graph': args.graph, 'nodes': n, 'edges_with_self_loops': E, 'd': d, 'attention_dim': A, 'heads': h,
               'rhs_evals_per_step': 4, 'hipgraph': use_graph, 'scale': args.scale,
               'attention_norm_idx': args.norm_idx, 'square_plus': args.square_plus,
               'early_stop_evaluator': bool(args.early_stop),
               'long_rows': graph.n_long_rows,
               # how the aggregation launches deal the rows to the 8 XCDs (gnpde_graph_t.xcd_deal, chosen per graph from the
               # measured imbalance of contiguous eighths; GNPDE_TUNE=10=1 / 10=2 force one or the other for A/B runs)
               'xcd_row_deal': {0: 'hashed_blocks' if graph.struct.xcd_deal == 1 else 'contiguous_eighths',
                                1: 'contiguous_eighths (forced)', 2: 'hashed_blocks (forced)'}.get(xcd_knob, '?'),
               'xcd_contiguous_imbalance': round(graph.xcd_imbalance_contiguous, 4),
               'algorithmic_bytes_per_rhs_eval': bytes_eval,
               'eval_gbs_vs_gather_model': round(bytes_eval * 4 * steps_per_s / 1e9, 1)},
    'roofline': {'kernel': kname, 'bound': bound,
                 'achieved': round(ach, 1), 'peak': peak, 'unit': 'GB/s',
                 'frac': round(ach / peak, 4), 'peak_source': peak_src, 'achieved_is': ach_is,
                 'gather_model_gbs': round(achieved, 1), 'row_gather_gbs': round(row_gather, 1),
                 'frac_of_row_gather_ceiling': None if ceiling is None else round(row_gather / ceiling['row_gather_gbs'], 4),
                 'hbm_peak': HBM_PEAK_GBS,
                 'frac_of_hbm_peak': None if resident else round(achieved / HBM_PEAK_GBS, 4),
                 'hbm_copy_rate': HBM_COPY_GBS,
                 'ceiling': ceiling, 'ceiling_uniform_random_ids': ceiling_uniform,
                 'epilogue_stream_bytes_per_launch_avg': 4 * d * n * 3,   # x0 + out every stage, y and k1 in two of four: 12 per step
                 'traffic': None, 'traffic_gbs': None,
                 'algorithmic_bytes_per_launch': bytes_spmm, 'avg_launch_us': round(t_spmm * 1e6, 2),
                 'compulsory_gather_bytes_per_launch': E * (4 + 4 * d) + n * (4 + 12 * d),
                 'dram_floor_bytes': dram_floor,
                 'gathered_table_mib': round(state_mb, 1),
                 'table_fits_infinity_cache': resident,
                 'secondary': secondary,
                 'note': ('the gathered state (%.0f MiB) fits the 256 MiB Infinity Cache: the gather-model bytes are served by '
                          'L2 / MALL, so HBM bandwidth is not their ceiling; `frac` = row-gather rate of the aggregation / row-gather '
                          'rate of a perfectly balanced gather from the same table measured in this run' % state_mb) if resident else
                         ('the gathered state (%.0f MiB) exceeds the 256 MiB Infinity Cache: gathers are DRAM traffic '
                          'except for hub columns; `frac` is against the HBM peak' % state_mb)},
  }
  if isinstance(traffic, dict):   # measured PMC record (tools/pmc_traffic.py): bytes + provenance
    tb = traffic.get('bytes_per_launch')
    out['roofline']['traffic'] = tb
    out['roofline']['traffic_gbs'] = round(tb / t_spmm / 1e9, 1) if tb else None
    if tb and not resident:
      out['roofline']['traffic_frac_of_hbm_peak'] = round(tb / t_spmm / 1e9 / HBM_PEAK_GBS, 4)
    src = {k: traffic.get(k) for k in ('kernel', 'commit', 'fetch_bytes', 'write_bytes', 'l2_hit_rate', 'method', 'source_sha16')}
    now = source_sha16('graph-neural-pde_amd/csrc/spmm.hip')
    # the record is stale when csrc/spmm.hip is no longer the file it was measured with (hash stored by tools/pmc_traffic.py)
    src['stale'] = bool(traffic.get('source_sha16') is None or now is None or traffic.get('source_sha16') != now)
    src['spmm_hip_sha16_now'] = now
    out['roofline']['traffic_source'] = src
  rf = out['roofline']
  if not resident and rf['frac'] > 1.0:
    # The gather model charges every neighbour row to HBM; when the hot rows of a power-law graph hit in L2 the model's bytes
    # exceed what crosses the memory interface and "algorithmic bytes / HBM peak" is no longer a fraction of anything.  Then
    # `frac` = bytes that DID cross it (PMC record of the same command, `traffic`) / launch duration / HBM peak; without a
    # fresh record, the row-gather rate against the measured gather ceiling.  The contract's ratio stays in `frac_algorithmic`.
    rf['frac_algorithmic'] = rf['frac']
    fresh = isinstance(traffic, dict) and traffic.get('bytes_per_launch') and not rf.get('traffic_source', {}).get('stale', True)
    if fresh:
      rf['frac'] = rf['traffic_frac_of_hbm_peak']
      rf['frac_is'] = ('measured HBM-side bytes of one launch (rocprofv3 PMC record of the same command, `traffic`) / its duration / the '
                       '8 TB/s HBM peak: the gather model (`achieved`, `frac_algorithmic`) exceeds the peak because %.0f %% of the gathered '
                       'lines hit in L2' % (100.0 * (traffic.get('l2_hit_rate') or 0.0)))
    elif rf.get('frac_of_row_gather_ceiling') is not None:
      rf['frac'] = rf['frac_of_row_gather_ceiling']
      rf['frac_is'] = 'row-gather rate of the launch / row-gather rate of the balanced gather of the same column ids measured in this run (`ceiling`)'
  _, view = solver_graph(f, x)
  if view is not None and early is None:
    # the timed solve ran on the relabelled graph: the same K steps on the graph as given, outside the timed region, must agree
    # bit for bit (the entries of a row keep their order, so every row sum is the same sum)
    out['config']['node_relabelling'] = dict(view.stats, what='graph.LocalityView: nodes relabelled (part by part of the native label-propagation '
                                             'partitioner, or by descending row length: the faster one by a timed aggregation), one-time graph '
                                             'preparation; the state is permuted on entry / exit of the solve, inside the timed region')
    try:
      with torch.no_grad():
        f.opt['gnpde_reorder'] = '0'
        z_plain = main_block(x)
        torch.cuda.synchronize()
      out['config']['node_relabelling']['solve_equal_to_unrelabelled_bitwise'] = bool(torch.equal(z, z_plain))
      del z_plain
    except Exception as exc:   # noqa: BLE001
      out['config']['node_relabelling']['check_error'] = repr(exc)[:200]
    finally:
      f.opt.pop('gnpde_reorder', None)
  else:
    out['config']['node_relabelling'] = None
  if early is not None:
    sol = early.solver
    out['early_stop'] = {'best_val': sol.best_val, 'best_test': sol.best_test, 'best_time': sol.best_time,
                         'classes': 40, 'evaluations': K}
  cpu_evals = args.cpu_evals
  if cpu_evals is None:
    cpu_evals = 6 if args.graph != 'rmat' else 0
  if not args.no_cpu_baseline and cpu_evals == 0:
    out['cpu_baseline'] = dict(host_info(), value=None, unit='steps/s', cores=None, kind='port',
                               sample='skipped: one evaluation of the reference op sequence at this shape materialises '
                                      '[E,d] temporaries of %.0f GB each (SURVEY 8d: "reference path OOM"); pass '
                                      '--cpu-evals N to time it on a host that has the memory' % (E * d * 4 / 1e9))
  elif not args.no_cpu_baseline:
    t_eval, ref = cpu_baseline(main_block, x_cpu, cpu_evals)
    with torch.no_grad():
      f.x0 = x
      got = f(0.0, x)
    from oracle import restate as R
    e_inf, e_2 = R.parity_error(got, ref)
    out['cpu_baseline'] = dict(host_info(), **{
      'value': round(1.0 / (4 * t_eval), 4), 'unit': 'steps/s', 'cores': torch.get_num_threads(),
      'threads_used': torch.get_num_threads(), 'kind': 'port',
      'kind_note': 'oracle/restate.py = the reference op sequence (index_select -> mul -> scatter_add, PyG softmax) in '
                   'torch CPU; the reference src/ itself needs /root/reference and third-party wheels that do not exist '
                   'on the GPU box',
      'sample': '%d full-size evaluations of f (= %.1f rk4 steps) of the same workload at the best of a few torch thread '
                'counts (`threads_used` of `host_cores`); steps/s = 1 / (4 t_eval)' % (cpu_evals, cpu_evals / 4.0),
      'ms_per_rhs_eval': round(t_eval * 1e3, 2)})
    out['parity_vs_oracle_one_eval'] = {'rel_max': e_inf, 'rel_l2': e_2}
    out['speedup_vs_cpu'] = round(steps_per_s / out['cpu_baseline']['value'], 1)
  print(json.dumps(out))


if __name__ == '__main__':
  main()
