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

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the CSR aggregation spmm_pair_kernel / spmm_wide_kernel
+ the hub-row fold): `achieved` = algorithmic bytes of DESIGN.md section 4 (SURVEY 8d's gather model) / its average launch
duration, measured here with HIP events around a captured graph of the four rk4-stage variants the solver runs; `peak` = HBM's
8 TB/s; `frac_algorithmic` = achieved / peak (exceeds 1 when the gathered table is resident in the Infinity Cache: the model
charges every cache-served row to HBM); `traffic` = the L2 -> fabric bytes of the same launches from rocprofv3 PMC passes that
THIS run makes over a child process of itself (`--pmc-child`; separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes; the stored record of profiles/hbm_traffic.json only when the passes cannot run, marked as such);
`frac_traffic` = traffic / duration / peak; `frac` = frac_traffic when the table is cache-resident (`bound: "mall"`) and
frac_algorithmic when it is not.  `stream_read_probe` = a coalesced L2-cold streaming read of the same table,
`hbm_bound_probe` = the same aggregation kernel on a device-generated 2-GiB table (d = 256, uniform degree 16), where every
gathered row comes from DRAM and algorithmic bytes / time / 8 TB/s is a true HBM fraction.  `roofline.secondary` times the
projection and the row attention.  `cpu_baseline` times the CPU oracle (the reference's op sequence) on this host's cores on a
bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

T_PROCESS_START = time.perf_counter()
DEFAULT_RUN_SECONDS = 290      # the default run (headline + every other configuration by child runs) is planned to end within this

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
  ap.add_argument('--graph', default='arxiv', choices=['arxiv', 'arxiv_flat', 'cora', 'rmat'],
                  help='arxiv: ogbn-arxiv shape with 40 planted communities (the headline); arxiv_flat: same degree skew, NO community '
                       'structure (what the node relabelling gains without planted locality)')
  ap.add_argument('--method', default='rk4', choices=['rk4', 'euler'], help='fixed-step method of the timed solve (BASELINE configs[0] is euler)')
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
  ap.add_argument('--no-live-pmc', action='store_true',
                  help='do not run the rocprofv3 --pmc passes over a child process (roofline.traffic then comes from the stored '
                       'record of profiles/hbm_traffic.json, marked as such)')
  ap.add_argument('--no-hbm-probe', action='store_true', help='skip roofline.hbm_bound_probe (a 2-GiB table, ~3 s)')
  ap.add_argument('--pmc-child', action='store_true',
                  help='internal: launch the kernels of one evaluation eagerly a few times and exit (the process the PMC passes profile)')
  ap.add_argument('--config', default=None, choices=['c4', 'cora-epoch', 'pubmed-adjoint', 'coauthor-adjoint', 'arxiv-adjoint'],
                  help='c4: BASELINE configs[3] -- ogbn-arxiv BLEND (beltrami split kernel, d = 64 + 98 = 162), block_transformer_rewiring in '
                       'evaluation mode, Laplacian function, dopri5 with tol_scale 11353, T = 3.676; prints its own JSON line (ms per forward).  '
                       'cora-epoch: the reference\'s flagship run -- best_params Cora (attention block, Laplacian function, dopri5, adjoint=False, '
                       '8 heads, A = 128, squareplus over columns) on a Cora-LCC-shaped graph: one epoch = run_GNN.py train() + test().  pubmed-adjoint: best_params '
                       'Pubmed\'s ODE block (attention block, Laplacian, dopri5, adjoint=True with adjoint_method adaptive_heun -- the reference\'s default '
                       'adjoint method) on a Pubmed-shaped graph: ms per training iteration of the block, native stages vs the flat host loop')
  ap.add_argument('--no-configs', action='store_true',
                  help='default line only: skip the `configs` block (the other BASELINE configurations, each measured by a child process of this script)')
  ap.add_argument('--configs-budget', type=float, default=None,
                  help='seconds the `configs` block may take in total (children that would not fit are skipped, and say so); default: what is '
                       'left of %d s since the start of this process, so that the default run ends within about five minutes' % DEFAULT_RUN_SECONDS)
  ap.add_argument('--parity-only', action='store_true', help='one evaluation of the CPU oracle for the parity figure, no timed cpu_baseline leg')
  ap.add_argument('--keep-pmc', default=None, help='directory that receives the raw counter_collection.csv files of the live PMC passes')
  ap.add_argument('--train', action='store_true',
                  help='training iteration instead of the inference solve: forward (tape-free native solver) + backward (native adjoint '
                       'solve, opt[adjoint] with adjoint_method rk4 / adjoint_step_size 1) of K steps each; prints its own JSON line')
  ap.add_argument('--no-adjoint', action='store_true',
                  help='with --train: opt[adjoint] off (run_GNN.py\'s default) -- recorded native solve + native reverse sweep, A/B against the host loop')
  ap.add_argument('--no-host-loop', action='store_true', help='with --train --no-adjoint: skip the host-loop A/B (the child the PMC passes profile)')
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
              tol_scale=1.0, data_norm='rw', method=args.method, step_size=1.0, max_iters=100, block='constant',
              function=args.function, time=float(args.steps))


class _Data(object):
  pass


GRAPH_NAMES = {'arxiv': 'ogbn-arxiv', 'arxiv_flat': 'ogbn-arxiv (no communities)', 'cora': 'Cora', 'rmat': 'RMAT-2M'}


def metric_name(graph, d, world=1, method='rk4'):
  """BASELINE.json's metric with the graph that was ACTUALLY run."""
  return 'ODE steps/sec (full-graph diffusion), %s d=%d %s' % (GRAPH_NAMES.get(graph, graph), d, method)


def workload_name(graph, function, K, method='rk4'):
  shape = {'arxiv': 'synthetic ogbn-arxiv-shaped graph (power-law degrees, 40 communities, shuffled ids)',
           'arxiv_flat': 'synthetic ogbn-arxiv-shaped graph WITHOUT community structure (power-law degrees, shuffled ids)',
           'cora': 'synthetic Cora-shaped graph (uniform random)',
           'rmat': 'synthetic R-MAT graph (Graph500 parameters, 2^21 nodes, 40 M generated edges, symmetrised)'}[graph]
  return '%s, GRAND-%s add_source, %s, step_size 1, T=%d, hipGraph-captured solver' % (
    shape, 'nl scaled_dot softmax attention' if function == 'transformer' else 'l',
    'rk4 3/8-rule' if method == 'rk4' else 'euler', K)


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
  if f.opt.get('method') == 'euler':      # the one stage the euler solver runs (ping-pong between two buffers)
    stages = [dict(stage=_lib.STAGE_EULER, y=y, out_y=ua, u=y), dict(stage=_lib.STAGE_EULER, y=ua, out_y=y, u=ua)] * 2
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
  # (the layout the solver uses for this shape: two tables q, k [n, A] when a key row is shorter than a cache line, else rows [n, 2A])
  split_ok = f.opt['attention_type'] == 'scaled_dot'
  if split_ok:
    q_, k_, ldqk, buf = ops.qk_tables(x, wqk, bqk, A)
    t_lin = timed_replay(lambda: ops.qk_tables(x, wqk, bqk, A, out=buf), 16)
  else:
    buf = ops.linear(x, wqk, bqk)
    q_, k_, ldqk = buf, buf[:, A:], 2 * A
    t_lin = timed_replay(lambda: ops.linear(x, wqk, bqk, out=buf), 16)
  st = ops.attention_struct(_lib.ATT_TYPES[f.opt['attention_type']], h, A, f.opt['attention_norm_idx'], f.opt['square_plus'],
                            q=q_, k=k_, ldqk=ldqk)
  t_att = timed_replay(lambda: ops.edge_attention(graph, st, True, False, False, like=x), 16)
  b_lin = n * (4 * d + 4 * 2 * A)
  b_att = E * (4 + 4 * A + 4) + n * (16 + 4 * A)
  out = [{'kernel': 'row attention: scores + softmax over the row + head mean (row_attention_sd_kernel, one launch per degree '
                    'class with the hub phases riding)', 'bytes': b_att, 'avg_us': round(t_att * 1e6, 2),
          'gbs': round(b_att / t_att / 1e9, 1)},
         {'kernel': 'q||k projection [n,d] x [d,2A] on the fp32 MFMA (linear_staged2_kernel / linear_lds_kernel)', 'bytes': b_lin,
          'avg_us': round(t_lin * 1e6, 2), 'gbs': round(b_lin / t_lin / 1e9, 1), 'key_table': bool(ldqk == A)}]
  return out


def source_sha16(rel):
  import hashlib
  try:
    return hashlib.sha256(open(os.path.join(ROOT, rel), 'rb').read()).hexdigest()[:16]
  except OSError:
    return None


KERNEL_SOURCES = ('graph-neural-pde_amd/csrc/spmm.hip', 'graph-neural-pde_amd/csrc/attention.hip', 'graph-neural-pde_amd/csrc/linear.hip',
                  'graph-neural-pde_amd/csrc/epilogue.h')


def kernel_sources_sha16():
  """One hash over every source file a kernel of the evaluation is compiled from: a stored PMC record is stale as soon as any
  of them changes (round 3 hashed spmm.hip only)."""
  import hashlib
  h = hashlib.sha256()
  for rel in KERNEL_SOURCES:
    try:
      h.update(open(os.path.join(ROOT, rel), 'rb').read())
    except OSError:
      return None
  return h.hexdigest()[:16]


def stream_read_probe(x):
  """Coalesced streaming read of the SAME table (gnpde_stream_read: 16-byte lanes, grid stride, eight loads in flight, nothing
  written but one float per workgroup), repeated back to back.  The table exceeds the L2s (32 MiB), so every pass finds its
  lines L2-cold; when it fits the 256-MiB Infinity Cache they come from there -- the rate the memory side can deliver this
  table's lines at, a hardware figure next to the aggregation's gather of them."""
  from gnpde_amd import _lib
  flat = x if x.is_contiguous() else x.contiguous()
  nfl = flat.numel() // 4 * 4
  sink = torch.empty(2048, dtype=torch.float32, device=x.device)
  L = _lib.lib()

  passes = max(1, min(64, int(2 ** 31 // max(nfl * 4, 1))))      # ~2 GiB read per launch

  def call():
    _lib.check(L.gnpde_stream_read(_lib.ptr(flat), nfl, passes, _lib.ptr(sink), sink.numel(), _lib.stream_of(flat)))
  t = timed_replay(call, 4)
  return {'gbs': round(passes * nfl * 4 / t / 1e9, 1), 'bytes_per_pass': nfl * 4, 'passes_per_launch': passes,
          'us_per_pass': round(t * 1e6 / passes, 2),
          'what': 'gnpde_stream_read: coalesced read of the same [n, d] table (16-byte lanes, 8 loads in flight per lane), %d whole-table '
                  'passes inside one launch: every pass is L2-cold (table > 32 MiB of L2), served by the Infinity Cache when the table '
                  'fits its 256 MiB, by DRAM otherwise' % passes}


def hbm_bound_probe(G, dev, log_n=21, d=256, k=16, reps=3):
  """The SAME aggregation entry (gnpde_spmm_rhs: what the solver launches, plain f = alpha (A u - u) + beta x0 epilogue) on a
  device-generated graph whose gathered table cannot sit in any cache: 2^log_n rows of d floats (2 GiB at the defaults, 8x the
  Infinity Cache), k uniformly random neighbours per row.  Every gathered row is DRAM traffic, so algorithmic bytes / time /
  8 TB/s is a fraction of the HBM roofline in the strict sense -- the driver-witnessed HBM-bound figure of the kernel whose
  headline shape (ogbn-arxiv, 83 MiB) is cache-resident."""
  from gnpde_amd import ops, graph as Gr
  n = 1 << log_n
  gen = torch.Generator(device=dev).manual_seed(4321)
  row = torch.arange(n, device=dev, dtype=torch.int64).repeat_interleave(k)
  col = torch.randint(0, n, (n * k,), device=dev, dtype=torch.int64, generator=gen)
  ei = torch.stack([row, col])
  del row, col
  t0 = time.perf_counter()
  g = Gr.CSRGraph(ei, n, device=dev)
  torch.cuda.synchronize()
  build_s = time.perf_counter() - t0
  E = g.e
  u = torch.empty(n, d, device=dev).normal_(generator=gen)
  x0 = torch.empty(n, d, device=dev).normal_(generator=gen)
  out = torch.empty_like(u)
  w = torch.empty(E, device=dev).uniform_(generator=gen) / k
  alpha = torch.zeros(1, device=dev)
  beta = torch.full((1,), 0.1, device=dev)
  t = timed_replay(lambda: ops.spmm_rhs(g, w, u, alpha, beta, x0, True, out=out), reps)
  nbytes = E * (8 + 4 * d) + n * (4 + 8 * d) + 4 * d * n       # B_l + source
  res = {'nodes': n, 'entries': E, 'd': d, 'neighbours_per_row': k, 'table_gib': round(n * d * 4 / 2 ** 30, 2),
         'graph_build_seconds': round(build_s, 2), 'avg_launch_us': round(t * 1e6, 1),
         'algorithmic_bytes_per_launch': nbytes, 'achieved': round(nbytes / t / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
         'frac': round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), 'frac_of_copy_rate': round(nbytes / t / 1e9 / HBM_COPY_GBS, 4),
         'what': 'gnpde_spmm_rhs (the aggregation + epilogue the solver launches) on a device-generated graph: %d rows of %d floats '
                 '(%.1f GiB table, %.0fx the Infinity Cache), %d uniformly random neighbours per row; algorithmic bytes '
                 'E (8 + 4 d) + N (4 + 8 d) + 4 d N / launch time / 8 TB/s' % (n, d, n * d * 4 / 2 ** 30, n * d * 4 / 2 ** 28, k)}
  del g, u, x0, out, w, ei
  torch.cuda.empty_cache()
  return res


def pmc_child(G, block, x, reps=3):
  """--pmc-child: the launches of the four evaluations of one rk4 step, eagerly, `reps` times -- per stage the projection, the row
  attention and that stage's variant of the aggregation, in the solver's order, on the solver's own graph -- and nothing else on the
  device afterwards (the parent folds the counters of the gnpde:: kernels by name)."""
  from gnpde_amd import ops, _lib
  f = block.odefunc
  graph, _ = solver_graph(f, x)
  dev = x.device
  bufs = [torch.randn_like(x) for _ in range(7)]
  y, k1, k2, k3, ua, ub, x0 = bufs
  alpha, beta = ops._scalar_dev(f.alpha_train, x), ops._scalar_dev(f.beta_train, x)
  stages = [dict(stage=_lib.STAGE_RK1C, out_y=ua, u=y), dict(stage=_lib.STAGE_RK2C, y=y, out_y=ub, u=ua),
            dict(stage=_lib.STAGE_RK3C, k1=ua, out_y=k1, u=ub), dict(stage=_lib.STAGE_RK4C, y=y, k1=ub, out_y=y, u=k1)]
  if f.opt.get('method') == 'euler':
    stages = [dict(stage=_lib.STAGE_EULER, y=y, out_y=ua, u=y), dict(stage=_lib.STAGE_EULER, y=ua, out_y=y, u=ua)] * 2
  w = torch.rand(max(graph.e, 1), device=dev) / 16
  att = None
  if hasattr(f, 'multihead_att_layer'):
    lay = f.multihead_att_layer
    wqk, bqk = lay.qk_weights()
    A, h = lay.attention_dim, lay.h
    qk = torch.empty(x.shape[0] * 2 * A, dtype=torch.float32, device=dev)     # (filled by the loop: exactly `reps` projection launches)
    split_ok = f.opt['attention_type'] == 'scaled_dot' and bool(_lib.lib().gnpde_linear_split_supported(
      _lib.ptr(x), x.shape[0], x.shape[1], x.stride(0), _lib.ptr(wqk), wqk.shape[0], wqk.stride(0), A))
    if split_ok:      # the solver's layout for this shape: two tables
      n_ = x.shape[0]
      q_, k_, ldqk = qk[:n_ * A].view(n_, A), qk[n_ * A:].view(n_, A), A
    else:
      q_ = qk.view(x.shape[0], 2 * A)
      k_, ldqk = q_[:, A:], 2 * A
    att = ops.attention_struct(_lib.ATT_TYPES[f.opt['attention_type']], h, A, f.opt['attention_norm_idx'], f.opt['square_plus'],
                               q=q_, k=k_, ldqk=ldqk)
  # as the solver issues them: EVERY aggregation is preceded by the projection and the row attention of its own stage input (they
  # sweep ~300 MB through the L2s in between: four aggregations back to back would find more of the state still cached than the
  # solve does -- 1.16 instead of 1.24 GB of L2 -> fabric traffic per launch, the difference the round-4 review found between the
  # live figure and the record of the solver's own launches)
  for _ in range(reps):
    for st in stages:
      kw = dict(st)
      u = kw.pop('u')
      if att is not None:
        if split_ok:
          ops.qk_tables(u, wqk, bqk, A, out=qk)
        else:
          ops.linear(u, wqk, bqk, out=qk.view(x.shape[0], 2 * A))
        ops.edge_attention(graph, att, True, False, False, like=x)
      ops.spmm_rhs(graph, w, u, alpha, beta, x0, True, dt=1.0, **kw)
  torch.cuda.synchronize()
  print(json.dumps({'pmc_child': 'done', 'aggregation_calls': len(stages) * reps, 'evaluations': len(stages) * reps}))


def pmc_passes(child, env, tag, keep_dir=None, timeout_s=150, marker='{"pmc_child"'):
  """L2 -> fabric bytes per kernel of the command `child`, measured IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE
  TCC_HIT_sum TCC_MISS_sum -- separate passes, with --kernel-trace only, as MI355X_MICROARCH.md section "rocprofv3 PMC slots"
  prescribes).  bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is in KiB and tallies the 128-B requests of wide coalesced
  reads at 64 B on gfx950 (same guide, section HBM).  Returns {kernel name: {...}} + '_calls' (the JSON line of the child that starts
  with `marker`) + '_seconds', or {'error': ...}.  keep_dir: the raw counter_collection.csv files are copied there."""
  import csv
  import glob
  import re
  import shutil
  import subprocess
  import tempfile
  exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
  if not os.path.exists(exe):
    return {'error': 'rocprofv3 not found'}
  base = tempfile.mkdtemp(prefix='gnpde_pmc_', dir='/tmp')
  acc, calls = {}, None
  t0 = time.perf_counter()
  try:
    for i, counters in enumerate((['FETCH_SIZE'], ['WRITE_SIZE', 'TCC_HIT_sum', 'TCC_MISS_sum'])):
      out_dir = os.path.join(base, 'pass%d' % i)
      cmd = [exe, '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', out_dir, '-o', 'p', '--'] + child
      res = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
      if res.returncode != 0:
        return {'error': 'rocprofv3 pass %d exited with %d: %s' % (i, res.returncode, (res.stderr or res.stdout)[-300:])}
      for line in res.stdout.splitlines():
        if line.startswith(marker):
          calls = json.loads(line)
      found = glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True)
      if not found:
        return {'error': 'rocprofv3 pass %d wrote no counter_collection.csv' % i}
      for path in found:
        if keep_dir:     # the raw counter records of this run, next to the line they produced
          try:
            os.makedirs(keep_dir, exist_ok=True)
            shutil.copy(path, os.path.join(keep_dir, 'live_pmc_%s_pass%d_%s.csv' % (tag, i, '_'.join(counters))))
          except OSError:
            pass
        for r in csv.DictReader(open(path)):
          name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
          name = re.sub(r'^void ', '', name).split('(')[0]
          if not name.startswith('gnpde::'):
            continue
          ent = acc.setdefault(name, {}).setdefault(r['Counter_Name'], [0, 0.0])
          ent[0] += 1
          ent[1] += float(r['Counter_Value'])
  except subprocess.TimeoutExpired:
    return {'error': 'rocprofv3 pass timed out after %d s' % timeout_s}
  except Exception as exc:   # noqa: BLE001 -- the line must not be lost to a profiler problem
    return {'error': repr(exc)[:300]}
  finally:
    shutil.rmtree(base, ignore_errors=True)
  out = {}
  for name, cs in acc.items():
    launches = max(v[0] for v in cs.values())
    tot = {c: v[1] for c, v in cs.items()}
    rec = {'launches': launches}
    if 'FETCH_SIZE' in tot and 'WRITE_SIZE' in tot:
      rec['fetch_bytes_total'] = 2.0 * tot['FETCH_SIZE'] * 1024
      rec['write_bytes_total'] = tot['WRITE_SIZE'] * 1024
      rec['bytes_per_launch'] = (rec['fetch_bytes_total'] + rec['write_bytes_total']) / launches
    if tot.get('TCC_HIT_sum', 0) + tot.get('TCC_MISS_sum', 0) > 0:
      rec['l2_hit_rate'] = round(tot['TCC_HIT_sum'] / (tot['TCC_HIT_sum'] + tot['TCC_MISS_sum']), 4)
    out[name] = rec
  out['_calls'] = calls
  out['_seconds'] = round(time.perf_counter() - t0, 1)
  return out


def live_pmc_traffic(args, reorder_mode, timeout_s=150):
  """The PMC passes over a child process of this script (--pmc-child) that launches the projection, the row attention and the four
  stage variants of the aggregation three times on the same graph, in the node order the parent's solver chose."""
  child = [sys.executable, os.path.abspath(__file__), '--pmc-child', '--graph', args.graph, '--scale', str(args.scale), '--seed', str(args.seed),
           '--function', args.function, '--norm-idx', str(args.norm_idx), '--steps', '1', '--warmup', '0', '--method', args.method]
  if args.att_dim:
    child += ['--att-dim', str(args.att_dim)]
  if args.heads:
    child += ['--heads', str(args.heads)]
  if args.square_plus:
    child += ['--square-plus']
  env = dict(os.environ, TMPDIR='/tmp', GNPDE_BENCH_REORDER=reorder_mode)
  return pmc_passes(child, env, '%s_%s' % (args.graph, args.function), getattr(args, 'keep_pmc', None), timeout_s)


def mode_pmc_traffic(args, mode_flags, tag, timeout_s=240):
  """The same two passes over a short run of this script in another mode (`--train`, `--config c4`): the child prints its own
  bench line, whose counts (evaluations, stages) turn the per-kernel totals into bytes per stage / per launch."""
  child = [sys.executable, os.path.abspath(__file__), '--scale', str(args.scale), '--seed', str(args.seed), '--no-live-pmc', '--no-cpu-baseline',
           '--no-configs', '--replays', '1'] + list(mode_flags)
  return pmc_passes(child, dict(os.environ, TMPDIR='/tmp'), tag, getattr(args, 'keep_pmc', None), timeout_s, marker='{"metric"')


def traffic_of(pmc, pattern, calls):
  """Bytes per call of the kernels whose name holds `pattern` (a call of the aggregation = its row kernel + the hub-row fold)."""
  recs = [(k, v) for k, v in pmc.items() if not k.startswith('_') and pattern in k and 'bytes_per_launch' in v]
  if not recs or not calls:
    return None
  total = sum(v['fetch_bytes_total'] + v['write_bytes_total'] for _, v in recs)
  big = max(recs, key=lambda kv: kv[1]['fetch_bytes_total'])
  return {'bytes_per_call': total / calls, 'fetch_bytes_per_call': sum(v['fetch_bytes_total'] for _, v in recs) / calls,
          'write_bytes_per_call': sum(v['write_bytes_total'] for _, v in recs) / calls,
          'kernels': {k: {'launches': v['launches'], 'bytes_per_launch': round(v['bytes_per_launch']), 'l2_hit_rate': v.get('l2_hit_rate')}
                      for k, v in recs},
          'l2_hit_rate': big[1].get('l2_hit_rate')}


def subset_parity(block, x, x_cpu, n_random=3000):
  """f is row-local once the neighbours are known: the rows of a SUBSET (the six largest hubs, 2-3-chunk rows, mid-degree rows,
  random rows) of one full-size evaluation against the oracle on the sub-graph those rows and all their neighbours induce
  (tests/test_rmat_gpu.py does the same) -- the parity check of a shape whose whole-graph oracle evaluation needs ~100 GB."""
  from oracle import restate as R
  f = block.odefunc
  lay = f.multihead_att_layer
  cpu = lambda t: t.detach().cpu()   # noqa: E731
  edge = cpu(f.edge_index)
  n = x.shape[0]
  deg = torch.bincount(edge[0], minlength=n)
  g = torch.Generator().manual_seed(9)
  hubs = torch.topk(deg, 6).indices
  long_small = torch.nonzero((deg > 512) & (deg <= 1100)).flatten()[:40]
  mid = torch.nonzero((deg > 16) & (deg <= 512)).flatten()
  mid = mid[torch.randperm(mid.numel(), generator=g)[:400]]
  rows = torch.unique(torch.cat([hubs, long_small, mid, torch.randperm(n, generator=g)[:n_random]]))
  pick = torch.zeros(n, dtype=torch.bool)
  pick[rows] = True
  keep = pick[edge[0]]
  r, c = edge[0][keep], edge[1][keep]
  nodes = torch.unique(torch.cat([rows, c]))
  sub_edge = torch.stack([torch.searchsorted(nodes, r), torch.searchsorted(nodes, c)])
  xs = x_cpu[nodes]
  with torch.no_grad():
    f.x0 = x
    got = f(0.0, x)
    ref = R.rhs_transformer(xs, sub_edge, cpu(lay.Q.weight), cpu(lay.Q.bias), cpu(lay.K.weight), cpu(lay.K.bias), lay.h,
                            cpu(f.alpha_train), cpu(f.beta_train), xs, False, True)[torch.searchsorted(nodes, rows)]
  e_inf, e_2 = R.parity_error(got[rows.to(got.device)], ref)
  return {'rel_max': e_inf, 'rel_l2': e_2, 'rows': int(rows.numel()), 'entries': int(keep.sum()), 'largest_row_entries': int(deg.max()),
          'what': 'one full-size evaluation, rows of a subset (6 largest hubs, 2-3-chunk rows, 400 mid-degree rows, %d random rows) vs the oracle on '
                  'the sub-graph they induce with all their neighbours' % n_random}


def cpu_baseline(block, x_cpu, evals):
  """Reference op sequence (oracle) on the host cores: a bounded number of full-size evaluations of f."""
  from oracle import restate as R
  f = block.odefunc
  cpu = lambda t: t.detach().cpu()
  edge = cpu(f.edge_index)
  if hasattr(f, 'multihead_att_layer'):
    lay = f.multihead_att_layer
    args = (cpu(lay.Q.weight), cpu(lay.Q.bias), cpu(lay.K.weight), cpu(lay.K.bias), lay.h)
    rhs = lambda y: R.rhs_transformer(y, edge, *args, cpu(f.alpha_train), cpu(f.beta_train), x_cpu, False, True,
                                      norm_idx=f.opt['attention_norm_idx'], square_plus=f.opt['square_plus'])
  else:
    w = cpu(f.edge_weight)
    rhs = lambda y: R.rhs_laplacian(y, edge, w, cpu(f.alpha_train), cpu(f.beta_train), x_cpu, False, True)
  # torch's CPU scatter/gather ops do not scale to every core of a large host: the thread count is chosen from the MEDIAN of
  # three evaluations per candidate (one sample each made the choice -- 8 or 32 threads -- a coin toss, +-15 % on the baseline)
  ncpu = os.cpu_count() or 1
  cands = sorted({c for c in (ncpu, 64, 32, 16, 8) if c <= ncpu}, reverse=True)
  if int(edge.shape[1]) * int(x_cpu.shape[1]) < 5_000_000:
    # a graph this small (Cora: 13 k entries x 80 columns) is not worth a sweep over thread counts -- re-sizing the OpenMP pool five times
    # cost the Cora children of the default run 9-19 s each, for a baseline that 8 threads give as well as 64
    cands = [min(ncpu, 8)]
  trials = {}
  with torch.no_grad():
    torch.set_num_threads(min(cands[0], 16) if evals == 0 else cands[0])
    out = rhs(x_cpu)  # warm-up, also the parity reference
    if evals == 0:       # --parity-only
      return None, out, {}
    for c in cands:
      torch.set_num_threads(c)
      ts = []
      for _ in range(3):
        t0 = time.perf_counter()
        rhs(x_cpu)
        ts.append(time.perf_counter() - t0)
      trials[c] = sorted(ts)[1]
    best_c = min(trials, key=lambda c: trials[c])
    torch.set_num_threads(best_c)
    t0 = time.perf_counter()
    for _ in range(evals):
      rhs(x_cpu)
    dt = (time.perf_counter() - t0) / evals
  return dt, out, {str(c): round(v * 1e3, 1) for c, v in trials.items()}


def c4_main(G, args, dev):
  """`--config c4`: BASELINE configs[3] as named -- ogbn-arxiv shape, BLEND (beltrami: 64 feature + 98 positional channels = d 162,
  the split exp kernel of reference src/function_transformer_attention.py:133-171), `block_transformer_rewiring` in evaluation mode
  (head-mean attention recomputed once per forward on the rw-normalised edge set, src/block_transformer_rewiring.py:185-241),
  Laplacian function, dopri5 with tol_scale 11353, T = 3.676 (best_params ogbn-arxiv).  One "step" of the metric is one FORWARD
  (the adaptive solver chooses its own steps): ms per forward, evaluations of f, and the aggregation's roofline at d = 162 (rows
  padded to 164 floats).  cpu_baseline: the restated torchdiffeq dopri5 (oracle/shims) over the oracle's right-hand side."""
  ei_cpu, n = G.synthetic.make_graph('arxiv', seed=args.seed, scale=args.scale)
  f0, p0 = 64, 98
  d = f0 + p0
  x_cpu = torch.randn(n, d, generator=torch.Generator().manual_seed(args.seed + 41)) * 0.5
  x, ei = x_cpu.to(dev), ei_cpu.to(dev)
  opt = dict(heads=2, attention_dim=32, attention_type='exp_kernel', attention_norm_idx=0, square_plus=False, reweight_attention=False,
             beltrami=True, feat_hidden_dim=f0, pos_enc_hidden_dim=p0, leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=10 ** 9,
             add_source=False, no_alpha_sigmoid=False, mix_features=False, hidden_dim=d, augment=False, adjoint=False,
             tol_scale=11353.558848254957, data_norm='rw', method='dopri5', step_size=1.0, max_iters=100, block='rewire_attention',
             function='laplacian', time=3.6760155951687636, att_samp_pct=0.81, use_flux=False, new_edges='k_hop_att', sparsify='S_hat',
             rw_addD=0.02, threshold_type='addD_rvR', rw_rmvR=0.02)
  data = _Data()
  data.x, data.edge_index, data.edge_attr, data.num_nodes = x, ei, None, n
  block = G.RewireAttODEblock(G.LaplacianODEFunc, [], opt, data, dev, t=torch.tensor([0, opt['time']])).to(dev)
  g = torch.Generator().manual_seed(args.seed + 5)
  with torch.no_grad():
    for name, p in block.named_parameters():
      if 'multihead_att_layer' in name and p.dim() >= 2:
        p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
      elif 'lengthscale' in name or 'output_var' in name:
        p.copy_((1.0 + 0.3 * torch.rand(p.shape, generator=g)).to(dev))
    block.odefunc.alpha_train.fill_(0.4)
  block.eval()
  block.set_x0(x)
  f = block.odefunc
  times = []
  with torch.no_grad():
    for _ in range(max(args.warmup, 1)):
      block(x)
    for _ in range(max(args.replays, 3)):
      f.nfe = 0
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      z = block(x)
      torch.cuda.synchronize()
      times.append(time.perf_counter() - t0)
  assert torch.isfinite(z).all()
  elapsed = sorted(times)[len(times) // 2]
  nfe = int(f.nfe)
  stats = dict(getattr(f, '_dopri5_stats', {}) or {})
  E = int(f.edge_index.shape[1])
  # the solve alone (the attention pass of the block outside): time of odeint on the function the block prepared
  with torch.no_grad():
    tt = torch.tensor([0.0, opt['time']], device=dev)
    kw = dict(method='dopri5', atol=block.atol, rtol=block.rtol)
    G.odeint(f, x, tt, **kw)
    solve = []
    for _ in range(3):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      G.odeint(f, x, tt, **kw)
      torch.cuda.synchronize()
      solve.append(time.perf_counter() - t0)
  t_solve = sorted(solve)[1]
  # roofline of the dominant kernel: the aggregation at d = 162 on padded rows, with the stage epilogue of the middle of a trial step
  # (LINCOMB with three earlier stage derivatives), HIP events around a captured graph of launches
  from gnpde_amd import ops, _lib
  view = f._locality_view(x) if hasattr(f, '_locality_view') else None
  bufs = [_lib.alloc_state(n, d, dev) for _ in range(3)]
  for b_ in bufs:
    b_.copy_(torch.randn(n, d, device=dev))
  u, y, out = bufs
  ld = u.stride(0)
  # through the descriptor the solver itself builds (padded rows -> 16-byte lanes), on the graph the solve runs on
  desc = f._descriptor(u, graph=None if view is None else view.graph)

  # the six launches of a trial step as the solver issues them: stage i streams y and the i earlier stage derivatives, writes k_i and
  # the next stage input; the seventh evaluation (first-same-as-last) writes k alone
  kbuf = [_lib.alloc_state(n, d, dev) for _ in range(6)]
  nxt = _lib.alloc_state(n, d, dev)

  def launch():
    for i in range(1, 6):
      ops.rhs_stage(desc, u, _lib.STAGE_LINCOMB, y=y, out_k=kbuf[i], out_y=nxt, prev=kbuf[:i], coef=[0.1] * (i + 1))
    ops.rhs_stage(desc, u, _lib.STAGE_LINCOMB, out_k=kbuf[0])
  try:
    t_agg = timed_replay(launch, 2) / 6
  except Exception:   # noqa: BLE001
    t_agg = None
  del kbuf, nxt
  # SURVEY 8d B_l (no source term in this configuration) + the stage algebra's streams, averaged over the six launches:
  # stage i reads y and i stage derivatives and writes the next stage input BEYOND B_l's own "u_i in, k_i out": i + 2 streams, none for the seventh
  streams = sum(i + 2 for i in range(1, 6)) / 6.0              # = 25 / 6 = 4.17 state-sized streams per launch
  bytes_agg = E * (8 + 4 * d) + n * (4 + 8 * d) + int(streams * 4 * d * n)
  traffic, traffic_src = None, None
  if not args.no_live_pmc and t_agg is not None:
    # counter traffic of the aggregation launches of a short run of this same configuration (two rocprofv3 --pmc passes over a child)
    pmc = mode_pmc_traffic(args, ['--config', 'c4', '--warmup', '1'], 'c4')
    if isinstance(pmc, dict) and 'error' not in pmc:
      wide = [(k, v) for k, v in pmc.items() if not k.startswith('_') and 'bytes_per_launch' in v and
              any(t in k for t in ('spmm_wide', 'spmm_rows', 'spmm_pair'))]
      if wide:
        launches = sum(v['launches'] for _, v in wide)
        tot = sum(v['fetch_bytes_total'] + v['write_bytes_total'] for k, v in pmc.items()
                  if not k.startswith('_') and 'bytes_per_launch' in v and 'spmm_' in k)
        traffic = tot / launches
        traffic_src = {'how': 'measured in this run: rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes, --kernel-trace only) '
                              'over a child process running this configuration (1 warm-up + 3 forwards); bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024, mean '
                              'over the %d aggregation launches of that run (row kernel + its share of the hub-row fold)' % launches,
                       'live': True, 'seconds': pmc.get('_seconds'),
                       'kernels': {k: {'launches': v['launches'], 'bytes_per_launch': round(v['bytes_per_launch']), 'l2_hit_rate': v.get('l2_hit_rate')}
                                   for k, v in pmc.items() if not k.startswith('_') and 'spmm_' in k and 'bytes_per_launch' in v}}
    elif isinstance(pmc, dict):
      traffic_src = {'live_pmc_error': pmc.get('error')}
  out_line = {
    'metric': 'forward passes/sec (adaptive solve), ogbn-arxiv BLEND d=162 dopri5',
    'value': round(1.0 / elapsed, 3), 'unit': 'forwards/s', 'n_gpus': 1, 'steps': 1, 'warmup': args.warmup,
    'ms_per_step': round(elapsed * 1e3, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
    'data': 'synthetic',
    'config': {'workload': 'BASELINE configs[3]: synthetic ogbn-arxiv-shaped graph, BLEND (beltrami split exp kernel, 64 feature + 98 positional '
                           'channels), block_transformer_rewiring (evaluation mode), Laplacian function, dopri5 tol_scale 11353, T = 3.676; '
                           'one step = one forward of the block (attention once + adaptive solve)',
               'nodes': n, 'edges_with_self_loops': E, 'd': d, 'row_stride': ld, 'attention_dim': 32, 'heads': 2, 'scale': args.scale},
    'ms_per_forward': round(elapsed * 1e3, 3), 'ms_solve_only': round(t_solve * 1e3, 3),
    'rhs_evals_per_forward': nfe, 'dopri5': stats,
    'ms_per_rhs_eval_incl_controller': round(t_solve * 1e3 / max(nfe, 1), 4),
    'aggregation_share_of_solve': None if t_agg is None else round(nfe * t_agg / t_solve, 4),
    'roofline': None if t_agg is None else {
      'kernel': 'CSR aggregation + explicit-RK stage epilogue at d = 162 (rows padded to 164 floats: spmm_wide_kernel, 41 of 64 16-byte lanes live), mean of the six launches of a dopri5 trial step, on the graph the solve runs on (relabelled: %s)' % (view is not None),
      'bound': 'mall', 'achieved': round(bytes_agg / t_agg / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
      'frac': None if traffic is None else round(traffic / t_agg / 1e9 / HBM_PEAK_GBS, 4),
      'frac_is': ('frac_traffic: L2 -> fabric counter bytes per aggregation launch / launch time / 8 TB/s (the 105-MiB state is Infinity-Cache resident)'
                  if traffic is not None else 'null: cache-resident table and no live counter traffic in this run; see frac_algorithmic'),
      'frac_algorithmic': round(bytes_agg / t_agg / 1e9 / HBM_PEAK_GBS, 4),
      'frac_algorithmic_is': 'gather-model bytes E (8 + 4 d) + N (4 + 8 d) + the stage streams (y, the earlier stage derivatives, k_i and the next stage input: 4.17 x 4 d N on average) / mean launch time / 8 TB/s; exceeds what is physical when rows are served from cache',
      'frac_traffic': None if traffic is None else round(traffic / t_agg / 1e9 / HBM_PEAK_GBS, 4),
      'algorithmic_bytes_per_launch': bytes_agg, 'avg_launch_us': round(t_agg * 1e6, 2),
      'traffic': None if traffic is None else round(traffic), 'traffic_source': traffic_src},
    'cpu_baseline': None,
  }
  if not args.no_cpu_baseline:
    # the restated torchdiffeq dopri5 over the oracle's right-hand side: ONE forward on the host (same accept / reject sequence)
    from oracle import restate as R
    from oracle.shims import install as SH
    cpu = lambda t: t.detach().cpu()   # noqa: E731
    w_edge = cpu(f.edge_weight)
    e_n = cpu(f.edge_index)
    calls = [0]

    def rhs(t, yv):
      calls[0] += 1
      return R.rhs_laplacian(yv, e_n, w_edge, cpu(f.alpha_train), cpu(f.beta_train), None, False, False)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    TD = SH          # oracle/shims/install.py: the restated torchdiffeq 0.2.1 (odeint with euler / rk4 / dopri5)
    t0 = time.perf_counter()
    with torch.no_grad():
      ref = TD.odeint(rhs, x_cpu, torch.tensor([0, opt['time']], dtype=torch.float32), method='dopri5', options={},
                      atol=opt['tol_scale'] * 1e-7, rtol=opt['tol_scale'] * 1e-9)[1]
    t_cpu = time.perf_counter() - t0
    from oracle.restate import parity_error
    e_inf, e_2 = parity_error(z, ref)
    out_line['cpu_baseline'] = dict(host_info(), value=round(1.0 / t_cpu, 5), unit='forwards/s', cores=torch.get_num_threads(), kind='port',
                                    sample='ONE forward solve: restated torchdiffeq 0.2.1 dopri5 (oracle/shims) over the oracle right-hand side '
                                           '(index_select -> mul -> scatter_add), %d evaluations, attention weights taken from the device' % calls[0],
                                    seconds=round(t_cpu, 2), rhs_evals=calls[0])
    out_line['parity_vs_restated_torchdiffeq'] = {'rel_max': e_inf, 'rel_l2': e_2, 'same_number_of_evaluations': bool(calls[0] == nfe)}
    out_line['speedup_vs_cpu'] = round(t_cpu / t_solve, 1)
  print(json.dumps(out_line))


CORA_BEST = dict(   # reference src/best_params.py 'Cora' (the entries the model reads), + run_GNN.py's defaults for the rest
  block='attention', function='laplacian', method='dopri5', adjoint=False, adjoint_method='adaptive_heun', adjoint_step_size=1,
  add_source=True, alpha_dim='sc', beta_dim='sc', attention_dim=128, attention_norm_idx=1, attention_type='scaled_dot', heads=8,
  hidden_dim=80, input_dropout=0.5, dropout=0.046878964627763316, square_plus=True, reweight_attention=False, mix_features=False,
  no_alpha_sigmoid=False, self_loop_weight=1, data_norm='rw', step_size=1, max_iters=100, max_nfe=2000, time=18.294754260552843,
  tol_scale=821.9773048827274, tol_scale_adjoint=1.0, beltrami=False, use_mlp=False, use_labels=False, fc_out=False, batch_norm=False,
  augment=False, leaky_relu_slope=0.2, optimizer='adamax', lr=0.022924849756740397, decay=0.00507685443154266, dataset='Cora',
  earlystopxT=3.0, max_test_steps=100, no_early=False, feat_hidden_dim=64, pos_enc_hidden_dim=16)


def cora_lcc_dataset(seed, dev):
  """A graph of the shape run_GNN.py trains Cora on (use_lcc: largest connected component, 2 485 nodes, 5 069 undirected edges =
  12 623 entries with self-loops), 1 433 bag-of-words features (row-normalised, ~18 words per paper), 7 classes, the planetoid-style
  split of the reference's set_train_val_test_split (20 per class train, rest of 1 500 development nodes validation, the others test)."""
  import numpy as np
  import gnpde_amd as G
  n, pairs, nfeat, ncls = 2485, 5069, 1433, 7
  rng = np.random.default_rng(seed)
  a, b = rng.integers(0, n, 2 * pairs), rng.integers(0, n, 2 * pairs)
  keep = a != b
  key = np.unique(np.minimum(a, b)[keep].astype(np.int64) * n + np.maximum(a, b)[keep])
  key = np.sort(rng.permutation(key)[:pairs])          # exactly `pairs` distinct non-loop pairs
  lo, hi = key // n, key % n
  row, col = np.concatenate([lo, hi]), np.concatenate([hi, lo])
  order = np.lexsort((col, row))
  ei = torch.from_numpy(np.stack([row[order], col[order]])).long()
  gen = torch.Generator().manual_seed(seed)
  x = (torch.rand(n, nfeat, generator=gen) < 18.0 / nfeat).float()
  x = x / x.sum(dim=1, keepdim=True).clamp_min(1.0)
  y = torch.randint(0, ncls, (n,), generator=gen)
  perm = torch.randperm(n, generator=gen)
  train = torch.zeros(n, dtype=torch.bool)
  for c in range(ncls):
    idx = perm[(y[perm] == c)][:20]
    train[idx] = True
  rest = perm[~train[perm]]
  val = torch.zeros(n, dtype=torch.bool)
  val[rest[:1500 - int(train.sum())]] = True
  test = ~(train | val)

  class _D(object):
    def __call__(self, *keys):
      for k in keys:
        yield k, getattr(self, k)
  data = _D()
  data.x, data.edge_index, data.edge_attr, data.y = x.to(dev), ei.to(dev), None, y.to(dev)
  data.train_mask, data.val_mask, data.test_mask = train.to(dev), val.to(dev), test.to(dev)
  data.num_nodes, data.num_features = n, nfeat
  return G.DummyDataset(data, ncls), x, ei


def cora_epoch_main(G, args, dev):
  """`--config cora-epoch`: the reference's flagship run, the one with a published number (notebooks/visualise_attention.ipynb:
  1.72-2.04 s per epoch, ~124 forward evaluations).  best_params Cora through gnpde_amd.GNN: encoder Linear 1433 -> 80, attention
  block (scaled-dot attention, 8 heads, A = 128, squareplus normalised over COLUMNS, computed once per forward with autograd history),
  Laplacian function, dopri5 (tol_scale 822, T = 18.29, adjoint = False), decoder; one EPOCH = run_GNN.py's train() (forward in train
  mode, cross-entropy on the train mask, backward, Adamax step; src/run_GNN.py:62-96) + test() (eval forward through the early-stopping
  test integrator to 3 T, three masked accuracies; :137-148, GNN_early.py:28-36).  Reports s/epoch, the NFE meters run_GNN.py prints,
  the split of an epoch into its phases (synchronised between phases in a separate pass), and which solve path ran."""
  import importlib
  O = importlib.import_module('gnpde_amd.odeint')      # (the package exports a FUNCTION of that name)
  opt = dict(CORA_BEST)
  dataset, x_cpu, ei_cpu = cora_lcc_dataset(args.seed, dev)
  data = dataset.data
  torch.manual_seed(args.seed)
  model = G.GNN(opt, dataset, dev).to(dev)
  # GNNEarly (reference src/GNN_early.py:28-36, 70-75): the early-stopping test integrator, handed the decoder before every forward
  model.odeblock.test_integrator = G.EarlyStopInt(model.T, opt, dev)
  model.odeblock.test_integrator.data = data
  params = [p for p in model.parameters() if p.requires_grad]
  optim = torch.optim.Adamax(params, lr=opt['lr'], weight_decay=opt['decay'])
  lf = torch.nn.CrossEntropyLoss()
  sync = torch.cuda.synchronize

  def train_step(timers=None):
    tick = (lambda k: None) if timers is None else (lambda k: (sync(), timers.__setitem__(k, time.perf_counter())))
    model.train()
    optim.zero_grad()
    tick('t0')
    out = model(data.x)
    loss = lf(out[data.train_mask], data.y.squeeze()[data.train_mask])
    tick('fwd')
    model.fm.update(model.getNFE())
    model.resetNFE()
    loss.backward()
    tick('bwd')
    optim.step()
    tick('opt')
    model.bm.update(model.getNFE())
    model.resetNFE()
    return loss

  @torch.no_grad()
  def test_step():
    model.eval()
    ti = model.odeblock.test_integrator
    ti.m2_weight = model.m2.weight.data.detach().clone()
    ti.m2_bias = model.m2.bias.data.detach().clone()
    logits, accs = model(data.x), []
    for _, mask in data('train_mask', 'val_mask', 'test_mask'):
      pred = logits[mask].max(1)[1]
      accs.append(pred.eq(data.y[mask]).sum().item() / mask.sum().item())
    nfe = model.getNFE()
    model.resetNFE()
    return accs, nfe

  W, K = max(args.warmup, 3), max(args.steps if args.steps != 100 else 20, 1)
  for _ in range(W):
    train_step()
    test_step()
  sync()
  model.fm.reset()
  model.bm.reset()
  ep, tr_t, te_t, eval_nfe = [], [], [], []
  for _ in range(K):
    sync()
    t0 = time.perf_counter()
    loss = train_step()
    lv = loss.item()                       # run_GNN.py: `return loss.item()`
    t1 = time.perf_counter()
    accs, nfe_eval = test_step()
    sync()
    t2 = time.perf_counter()
    stats_eval = dict(getattr(model.odeblock.odefunc, '_dopri5_stats', {}) or {})
    ep.append(t2 - t0)
    tr_t.append(t1 - t0)
    te_t.append(t2 - t1)
    eval_nfe.append(nfe_eval)
  assert lv == lv, 'loss is NaN'
  med = lambda v: sorted(v)[len(v) // 2]     # noqa: E731
  fwd_nfe = model.fm.sum / max(model.fm.cnt, 1)
  bwd_nfe = model.bm.sum / max(model.bm.cnt, 1)
  # phases, synchronised in between (a separate pass: the syncs themselves cost a little)
  ph = {'fwd': [], 'bwd': [], 'opt': []}
  for _ in range(5):
    tm = {}
    train_step(tm)
    ph['fwd'].append(tm['fwd'] - tm['t0'])
    ph['bwd'].append(tm['bwd'] - tm['fwd'])
    ph['opt'].append(tm['opt'] - tm['bwd'])
  # device-busy share: HIP events around the solve region only cannot see idle gaps, so measure the same training forward+backward once
  # with the kernel launches counted by the library (gnpde_launch_count, if exported) -- otherwise report wall times only
  f = model.odeblock.odefunc
  path = getattr(f, '_last_train_solve', None) or ('native recorded-tape dopri5' if f.__dict__.get('_tape_state') else
                                                   'host controller loop (odeint._solve_dopri5) over kernel-backed autograd Functions')
  E = int(f.edge_index.shape[1])
  out = {
    'metric': 'seconds per epoch (train + test), Cora best_params (attention block, Laplacian, dopri5, adjoint=False)',
    'value': round(med(ep), 6), 'unit': 's/epoch', 'n_gpus': 1, 'steps': K, 'warmup': W, 'ms_per_step': round(med(ep) * 1e3, 4),
    'higher_is_better': False, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
    'config': {'workload': 'reference best_params Cora on a Cora-LCC-shaped synthetic graph: GNN (encoder 1433 -> 80, attention block: scaled-dot, 8 heads, '
                           'A = 128, squareplus over columns; Laplacian function; dopri5 tol_scale 822, T = 18.29, adjoint=False; decoder 80 -> 7); one step = '
                           'one EPOCH = run_GNN.py train() + test() (early-stopping test integrator to 3 T)',
               'nodes': data.num_nodes, 'edges_with_self_loops': E, 'features': data.num_features, 'classes': dataset.num_classes,
               'hidden_dim': opt['hidden_dim'], 'heads': opt['heads'], 'attention_dim': opt['attention_dim'], 'optimizer': 'adamax'},
    'ms_train_step': round(med(tr_t) * 1e3, 3), 'ms_test_step': round(med(te_t) * 1e3, 3),
    'ms_train_phases_synchronised': {k: round(med(v) * 1e3, 3) for k, v in ph.items()},
    'nfe_forward_per_epoch': fwd_nfe, 'nfe_backward_per_epoch': bwd_nfe, 'nfe_test_per_epoch': med(eval_nfe),
    'train_solve_path': path, 'train_solve': dict(getattr(f, '_dopri5_stats', {}) or {}), 'test_solve': stats_eval,
    'final_loss': round(lv, 5), 'accuracies_last_epoch': [round(a, 4) for a in accs],
    'reference_published': {'s_per_epoch': [1.72, 2.04], 'nfe_forward': 124, 'where': 'notebooks/visualise_attention.ipynb:132-138 (real Cora, the authors\' GPU)',
                            'note': 'another graph (real Cora) on other hardware: context, not a baseline for vs_baseline'},
    'roofline': None, 'roofline_note': 'launch-latency bound: the 0.8-MB state lives in one L2; there is no bandwidth roofline to quote (DESIGN.md section 4, small graphs)',
    'cpu_baseline': None,
  }
  if not args.no_cpu_baseline:
    # the oracle's right-hand side under the restated torchdiffeq controller on the host cores: ONE training forward + backward of the
    # block in eval-mode arithmetic (no dropout), and the device block's forward against it (values + number of evaluations)
    from oracle import restate as R
    cpu = lambda t: t.detach().cpu()   # noqa: E731
    blk = model.odeblock
    lay = blk.multihead_att_layer
    model.eval()
    with torch.no_grad():
      h0 = model.encode(data.x)
      blk.set_x0(h0)
      f.nfe = 0
      # the block's forward with the plain integrator (AttODEblock.forward: attention once, then odeint) instead of the early-stopping one
      f.attention_weights = blk.get_attention_weights(h0)
      z_dev = G.odeint(f, h0, blk.t.type_as(h0), method='dopri5', options={'step_size': opt['step_size']}, atol=blk.atol, rtol=blk.rtol)[1]
      nfe_dev = int(f.nfe)
      f.nfe = 0
    hc = cpu(h0)
    e_n, _ = R.get_rw_adj(ei_cpu, None, 1, 1, data.num_nodes)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    ps = [cpu(p).clone().requires_grad_(True) for p in (lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias)]
    hx = hc.clone().requires_grad_(True)
    calls = [0]
    t0 = time.perf_counter()
    att, _ = R.transformer_attention(hx, e_n, ps[0], ps[1], ps[2], ps[3], opt['heads'], norm_idx=1, square_plus=True)

    def rhs(t, yv):
      calls[0] += 1
      return R.rhs_laplacian(yv, e_n, att, cpu(f.alpha_train), cpu(f.beta_train), hc, False, True)
    zr = O._solve_dopri5(rhs, hx, torch.tensor([0, opt['time']]), opt['tol_scale'] * 1e-9, opt['tol_scale'] * 1e-7)[1]
    t_f = time.perf_counter() - t0
    (zr ** 2).sum().backward()
    t_cpu = time.perf_counter() - t0
    e_inf, e_2 = R.parity_error(z_dev, zr.detach())
    out['cpu_baseline'] = dict(host_info(), value=round(t_cpu, 4), unit='s per block forward+backward', cores=torch.get_num_threads(), kind='port',
                               sample='ONE forward + backward of the ODE block alone (attention once + dopri5 solve, %d evaluations; no encoder / decoder / '
                                      'optimiser / test pass): the oracle right-hand side under the restated torchdiffeq 0.2.1 controller, torch CPU autograd' % calls[0],
                               forward_seconds=round(t_f, 4), rhs_evals=calls[0])
    out['parity_vs_restated_torchdiffeq'] = {'rel_max': e_inf, 'rel_l2': e_2, 'rhs_evals_device': nfe_dev, 'rhs_evals_oracle': calls[0],
                                             'same_number_of_evaluations': bool(nfe_dev == calls[0]),
                                             'what': 'eval-mode forward of the block (plain dopri5 to T) on the device vs the CPU oracle'}
  print(json.dumps(out))


def pubmed_adjoint_main(G, args, dev):
  """`--config pubmed-adjoint`: the ODE block of best_params Pubmed -- attention block (cosine_sim, 1 head, A = 16, squareplus), Laplacian
  function, d = 128, dopri5 `tol_scale` 1991, T = 12.94, `adjoint = True` with `adjoint_method = adaptive_heun` (run_GNN.py's default
  adjoint method) and `tol_scale_adjoint` 16 324 -- on a Pubmed-shaped graph (19 717 nodes, 44 324 undirected edges): one training
  iteration of the block (forward: the device-controlled dopri5; backward: torchdiffeq's augmented system integrated component-wise with
  native stages, odeint._adjoint_adaptive_native) next to the same iteration through torchdiffeq's flat-vector loop
  (opt['gnpde_host_adjoint'])."""
  import numpy as np
  preset = getattr(args, 'config', 'pubmed-adjoint')
  if preset == 'arxiv-adjoint':
    # best_params ogbn-arxiv, the reference's largest flagship run, in TRAINING: hard-attention block (the strongest 81 % of the edges by
    # head-mean attention carry the diffusion, reference src/block_transformer_hard_attention.py:48-66), Laplacian function, d = 162,
    # dopri5 forward (tol_scale 11 353, T = 3.676), adjoint_method rk4 with adjoint_step_size 1
    ei_cpu, n = G.synthetic.make_graph('arxiv', seed=args.seed, scale=args.scale)
    d = 162
    ei = ei_cpu.to(dev)
  else:
    n, pairs, d = (19717, 44324, 128) if preset == 'pubmed-adjoint' else (18333, 81894, 16)
    rng = np.random.default_rng(args.seed + 11)
    a, b = rng.integers(0, n, 2 * pairs), rng.integers(0, n, 2 * pairs)
    keep = a != b
    key = np.unique(np.minimum(a, b)[keep].astype(np.int64) * n + np.maximum(a, b)[keep])
    key = np.sort(rng.permutation(key)[:pairs])
    lo, hi = key // n, key % n
    row, col = np.concatenate([lo, hi]), np.concatenate([hi, lo])
    order = np.lexsort((col, row))
    ei = torch.from_numpy(np.stack([row[order], col[order]])).long().to(dev)
  x = (torch.randn(n, d, generator=torch.Generator().manual_seed(args.seed + 12)) * 0.5).to(dev)
  base = dict(heads=1, attention_dim=16, attention_type='cosine_sim', attention_norm_idx=0, square_plus=True, reweight_attention=False, beltrami=False,
              leaky_relu_slope=0.2, self_loop_weight=1, max_nfe=5000, add_source=True, no_alpha_sigmoid=False, mix_features=False, hidden_dim=d,
              augment=False, adjoint=True, adjoint_method='adaptive_heun', adjoint_step_size=1, tol_scale=1991.0688305523001,
              tol_scale_adjoint=16324.368093998313, data_norm='rw', method='dopri5', step_size=1, max_iters=100, block='attention',
              function='laplacian', time=12.942327880200853)
  label = 'Pubmed'
  if preset == 'coauthor-adjoint':
    # best_params CoauthorCS: hidden_dim 16, 4 heads, A = 8, scaled_dot normalised over columns with squareplus, no source term, no self-loops,
    # dopri5 tol_scale 9349, T = 3.126, adjoint_method dopri5 with tol_scale_adjoint 6599
    label = 'CoauthorCS'
    base.update(heads=4, attention_dim=8, attention_type='scaled_dot', attention_norm_idx=1, square_plus=True, add_source=False, self_loop_weight=0,
                adjoint_method='dopri5', tol_scale=9348.983916372074, tol_scale_adjoint=6599.1250595331385, time=3.126400580172773, max_nfe=3000)
  Block = G.AttODEblock
  if preset == 'arxiv-adjoint':
    label = 'ogbn-arxiv'
    Block = G.HardAttODEblock
    base.update(heads=2, attention_dim=32, attention_type='scaled_dot', attention_norm_idx=0, square_plus=False, add_source=False, self_loop_weight=1,
                adjoint_method='rk4', adjoint_step_size=1, tol_scale=11353.558848254957, tol_scale_adjoint=1.0, time=3.6760155951687636, max_nfe=500,
                block='hard_attention', att_samp_pct=0.8105268910037231, use_flux=False)
  data = _Data()
  data.x, data.edge_index, data.edge_attr, data.num_nodes = x, ei, None, n
  res = {}
  for host in (False, True):
    opt = dict(base, gnpde_host_adjoint=host)
    block = Block(G.LaplacianODEFunc, [], opt, data, dev, t=torch.tensor([0, opt['time']])).to(dev)
    g = torch.Generator().manual_seed(args.seed + 13)
    with torch.no_grad():         # EVERY parameter from the seeded generator (nn.Linear's own bias init draws from the global RNG: two blocks would differ)
      for name, p in block.named_parameters():
        if p.dim() >= 2:
          p.copy_((torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5).to(dev))
        elif name.endswith('.bias'):
          p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
    block.train()
    runs = []
    for it in range(2 + (max(args.replays, 3) if not host else 2)):
      xin = x.clone().requires_grad_(True)
      block.set_x0(xin)
      block.odefunc.nfe = 0
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      z = block(xin)
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      nf = block.odefunc.nfe
      z.sum().backward()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      runs.append((t1 - t0, t2 - t1, nf, block.odefunc.nfe - nf))
    runs = runs[2:]
    fw = sorted(r[0] for r in runs)[len(runs) // 2]
    bw = sorted(r[1] for r in runs)[len(runs) // 2]
    res[host] = dict(forward_ms=round(fw * 1e3, 3), backward_ms=round(bw * 1e3, 3), evals_forward=runs[-1][2], augmented_evals_backward=runs[-1][3],
                     grad_x=xin.grad.detach().clone(), z=z.detach())
  from oracle import restate as R
  e_inf, e_2 = R.parity_error(res[False]['grad_x'], res[True]['grad_x'])
  nat, hst = res[False], res[True]
  out = {
    'metric': 'ms per training iteration of the ODE block (forward + %s adjoint backward), %s shape d=%d' % (base['adjoint_method'], label, d),
    'value': round(nat['forward_ms'] + nat['backward_ms'], 3), 'unit': 'ms', 'n_gpus': 1, 'steps': 1, 'warmup': 2,
    'ms_per_step': round(nat['forward_ms'] + nat['backward_ms'], 3), 'higher_is_better': False, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
    'data': 'synthetic',
    'config': {'workload': ('ODE block of best_params %s (%s block: %s, %d head(s), A = %d%s; Laplacian function; dopri5 tol_scale %.0f, '
                            'T = %.2f; adjoint=True, adjoint_method %s, tol_scale_adjoint %.0f) on a %s-shaped synthetic graph; one step = one training '
                            'iteration of the block (loss = sum of the output)') % (label, base['block'].replace('_', '-'), base['attention_type'], base['heads'], base['attention_dim'],
                                                                                     ', squareplus' if base['square_plus'] else '',
                                                                                     base['tol_scale'], base['time'], base['adjoint_method'],
                                                                                     base['tol_scale_adjoint'], label),
               'nodes': n, 'edges_with_self_loops': int(ei.shape[1]) + (n if base['self_loop_weight'] else 0), 'd': d},
    'forward_ms': nat['forward_ms'], 'backward_ms': nat['backward_ms'], 'evals_forward': nat['evals_forward'],
    'augmented_evals_backward': nat['augmented_evals_backward'],
    'flat_host_loop': {k: hst[k] for k in ('forward_ms', 'backward_ms', 'evals_forward', 'augmented_evals_backward')},
    'backward_speedup_vs_flat_host_loop': round(hst['backward_ms'] / nat['backward_ms'], 2),
    'parity_vs_flat_host_loop': {'grad_x_rel_max': e_inf, 'grad_x_rel_l2': e_2, 'z_bitwise_equal': bool(torch.equal(nat['z'], hst['z'])),
                                 'what': 'dL/dx of the same iteration (identical parameters) through torchdiffeq\'s flat-vector loop; the two take the same '
                                         'steps when `augmented_evals_backward` agree'},
    'roofline': None, 'roofline_note': ('the aggregation kernels of the headline / C4 lines at this width (see c4_arxiv_blend_dopri5)' if preset == 'arxiv-adjoint' else
                                        'launch-bound (10-MB state, ~250 us of host work per augmented evaluation): no bandwidth roofline to quote'),
    'cpu_baseline': None,
  }
  print(json.dumps(out))


def train_main(G, args, opt, cfg, ei, n, x, dev):
  """`--train`: one training iteration of the ODE block at the benchmark shape -- forward = the tape-free native solver (K rk4
  steps, as inference), backward = the native adjoint solve (csrc/adjoint.hip: K rk4 steps of the augmented system, 4 K stages of
  f + VJP + parameter gradients, one hipGraph), as the reference trains ogbn-arxiv (best_params adjoint=True; src/base_classes.py:44-47,
  run_GNN.py:62-96).  The loss is sum(z * c) with a fixed random c; K steps forward AND backward are inside the timed region."""
  d, K, W = cfg['d'], args.steps, args.warmup
  A, h = opt['attention_dim'], opt['heads']
  topt = dict(opt, adjoint=True, adjoint_method='rk4', adjoint_step_size=1.0, tol_scale_adjoint=1.0, time=float(K))
  block = make_block(G, topt, ei, n, x, dev, float(K), args.seed)
  block.train()
  c = torch.randn(n, d, generator=torch.Generator().manual_seed(args.seed + 3)).to(dev)

  def iteration():
    for p in block.parameters():
      p.grad = None
    xin = x.clone().requires_grad_(True)
    block.set_x0(xin)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z = block(xin)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    (z * c).sum().backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, xin.grad

  for _ in range(max(W, 2)):          # (graph capture, relabelling probe and allocator warm-up happen here)
    iteration()
  runs = [iteration() for _ in range(max(args.replays, 1))]
  fw = sorted(r[0] for r in runs)[len(runs) // 2]
  bw = sorted(r[1] for r in runs)[len(runs) // 2]
  gx = runs[-1][2]
  assert torch.isfinite(gx).all()
  f = block.odefunc
  E = int(f.edge_index.shape[1])
  native = bool(f.__dict__.get('_adjoint_state'))
  # algorithmic bytes of ONE adjoint stage (f + VJP + parameter gradients), gather model as SURVEY 8d:
  agg = E * (8 + 4 * d) + n * (4 + 8 * d) + 4 * d * n                       # an aggregation with a source term (B_l + source)
  b_rows = agg + 4 * E + 4 * d * n                                           # F + r_e written + g_i read
  b_vt = agg                                                                 # V on the transposed CSR (P as the source term)
  b_proj = n * (4 * d + 8 * A)                                               # q||k projection
  b_att = E * (4 + 4 * A + 4) + n * (16 + 4 * A)                             # row attention -> w
  b_attb = E * (4 + 4 + 4 * A + 4 * h) + n * (16 + 4 * A)                    # normaliser backward: colidx, r, k rows in; ds out
  b_dqk = 2 * (E * (4 + 4 * h + 4 * A) + n * (16 + 4 * A)) + E * 4           # d q and d k (+ the position map of the transposed graph)
  b_pg = n * (8 * A + 4 * d)                                                 # P = [dq dk] [Wq;Wk]
  b_perm = E * 12                                                            # weights into the transposed order
  b_gram = n * (8 * A + 4 * d)                                               # [dq dk]^T u_y
  stage_bytes = b_rows + b_vt + b_proj + b_att + b_attb + b_dqk + b_pg + b_perm + b_gram
  t_stage = bw / (4 * K)
  traffic, traffic_src = None, None
  if not args.no_live_pmc and native:
    # counter traffic of ONE adjoint stage: the per-kernel totals of a short run of this mode (two rocprofv3 --pmc passes over a child).
    # Kernels that also run in the forward solve (projection, row attention, aggregation, folds) are charged to the backward by
    # their share of the launches: S stages against F forward evaluations, one launch of each per evaluation / stage.
    kc, wc = 2, 2
    flags = ['--train', '--steps', str(kc), '--warmup', str(wc), '--graph', args.graph]
    pmc = mode_pmc_traffic(args, flags, 'train')
    if isinstance(pmc, dict) and 'error' not in pmc:
      iters = max(wc, 2) + 1
      F = S = float(iters * 4 * kc)
      shared = ('linear_staged', 'linear_persistent', 'row_attention', 'spmm_pair', 'spmm_wide', 'spmm_rows', 'spmm_long_reduce',
                'hub_rowpart_fold', 'normalise_heads')
      tot, by = 0.0, {}
      for k, v in pmc.items():
        if k.startswith('_') or 'bytes_per_launch' not in v:
          continue
        b = v['fetch_bytes_total'] + v['write_bytes_total']
        share = S / (S + F) if any(t in k for t in shared) else 1.0
        by[k] = {'launches': v['launches'], 'bytes_per_launch': round(v['bytes_per_launch']), 'l2_hit_rate': v.get('l2_hit_rate'),
                 'charged_to_backward': round(share, 3)}
        tot += b * share
      traffic = tot / S
      traffic_src = {'how': 'measured in this run: rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes, --kernel-trace only) over '
                            'a child process running `bench.py --train --steps %d` (%d iterations = %d forward evaluations + %d adjoint stages); bytes = '
                            '(2 FETCH_SIZE + WRITE_SIZE) * 1024 summed over every gnpde kernel, kernels shared with the forward solve charged S / (S + F), '
                            'per stage' % (kc, iters, int(F), int(S)),
                     'live': True, 'seconds': pmc.get('_seconds'), 'kernels': by}
    elif isinstance(pmc, dict):
      traffic_src = {'live_pmc_error': pmc.get('error')}
  resident = n * d * 4 < 2 ** 28
  vjp_parity = None
  if not args.no_cpu_baseline:
    # parity of the training arithmetic at FULL size: f and its vector-Jacobian product (d/dx, d/dWq, d/dWk, d/dalpha) of one evaluation
    # through the native backward kernels against CPU autograd through the oracle's right-hand side (float32, 32 threads)
    from oracle import restate as R
    lay = f.multihead_att_layer
    cpu = lambda t: t.detach().cpu()   # noqa: E731
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    xg = x.clone().requires_grad_(True)
    f.x0 = x
    for p_ in block.parameters():
      p_.grad = None
    fv = f(0.0, xg)
    fv.backward(c)
    torch.cuda.synchronize()
    xc = cpu(x).requires_grad_(True)
    ps = [cpu(t).clone().requires_grad_(True) for t in (lay.Q.weight, lay.Q.bias, lay.K.weight, lay.K.bias, f.alpha_train, f.beta_train)]
    t0 = time.perf_counter()
    fr = R.rhs_transformer(xc, cpu(f.edge_index), ps[0], ps[1], ps[2], ps[3], lay.h, ps[4], ps[5], cpu(x), False, True)
    fr.backward(cpu(c))
    t_cpu = time.perf_counter() - t0
    pe = R.parity_error
    vjp_parity = {'f': pe(fv, fr.detach())[0], 'dx': pe(xg.grad, xc.grad)[0], 'dWq': pe(lay.Q.weight.grad, ps[0].grad)[0],
                  'dWk': pe(lay.K.weight.grad, ps[2].grad)[0], 'dalpha': pe(f.alpha_train.grad.reshape(1), ps[4].grad.reshape(1))[0],
                  'what': 'rel max error of one full-size evaluation and its VJP (native kernels) vs CPU autograd through the oracle; '
                          'gradient bar 2e-4 (tests/test_autograd_gpu.py)', 'cpu_seconds_f_plus_vjp': round(t_cpu, 2)}
  out = {
    'metric': 'training ODE steps/sec (forward solve + adjoint backward solve), %s d=%d rk4' % (GRAPH_NAMES.get(args.graph, args.graph), d),
    'value': round(K / (fw + bw), 3), 'unit': 'steps/s', 'n_gpus': 1, 'steps': K, 'warmup': W,
    'ms_per_step': round(1e3 * (fw + bw) / K, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
    'dtype': 'f32', 'data': 'synthetic',
    'config': {'workload': workload_name(args.graph, args.function, K) + '; TRAINING iteration: forward + adjoint (rk4, step 1) backward',
               'graph': args.graph, 'nodes': n, 'edges_with_self_loops': E, 'd': d, 'attention_dim': A, 'heads': h,
               'native_adjoint_solver': native, 'rhs_evals_forward': 4 * K, 'f_plus_vjp_stages_backward': 4 * K},
    'forward_ms': round(fw * 1e3, 3), 'backward_ms': round(bw * 1e3, 3),
    'f_plus_vjp_ms': round(t_stage * 1e3, 4),
    'roofline': {'kernel': 'one stage of the adjoint solve = f + VJP + parameter gradients (projection, row attention, adjoint_rows_kernel '
                           '[aggregation + SDDMM + dots], normaliser backward, d q / d k row sums, P GEMM, permute, aggregation on the '
                           'transposed CSR, Gram + folds), timed as backward wall time / stages',
                 'bound': 'mall' if resident else 'hbm', 'achieved': round(stage_bytes / t_stage / 1e9, 1), 'peak': HBM_PEAK_GBS,
                 'unit': 'GB/s',
                 'frac': (round(traffic / t_stage / 1e9 / HBM_PEAK_GBS, 4) if traffic is not None else
                          (None if resident else round(stage_bytes / t_stage / 1e9 / HBM_PEAK_GBS, 4))),
                 'frac_is': ('frac_traffic: L2 -> fabric counter bytes of one stage / its time / 8 TB/s' if traffic is not None else
                             'null: cache-resident tables and no live counter traffic in this run; see frac_algorithmic' if resident else
                             'frac_algorithmic'),
                 'frac_algorithmic': round(stage_bytes / t_stage / 1e9 / HBM_PEAK_GBS, 4),
                 'frac_traffic': None if traffic is None else round(traffic / t_stage / 1e9 / HBM_PEAK_GBS, 4),
                 'algorithmic_bytes_per_stage': stage_bytes,
                 'bytes_by_kernel': {'adjoint_rows': b_rows, 'aggregation_transposed': b_vt, 'projection': b_proj, 'row_attention': b_att,
                                     'normaliser_backward': b_attb, 'dq_dk_row_sums': b_dqk, 'p_gemm': b_pg, 'permute': b_perm, 'gram': b_gram},
                 'traffic': None if traffic is None else round(traffic), 'traffic_source': traffic_src},
    'cpu_baseline': None if vjp_parity is None else dict(host_info(), value=round(1.0 / (4 * vjp_parity['cpu_seconds_f_plus_vjp']), 4), unit='steps/s',
                                                         cores=torch.get_num_threads(), kind='port',
                                                         sample='ONE full-size evaluation of f + its VJP by torch CPU autograd through the oracle right-hand '
                                                                'side; steps/s = 1 / (4 x that), the backward half of a training step only'),
    'parity_vjp_one_eval_vs_oracle': vjp_parity,
    'note': 'rocprofv3 --kernel-trace --stats of this command: profiles/r05_train_kernel_stats.csv',
  }
  print(json.dumps(out))


def train_no_adjoint_main(G, args, opt, cfg, ei, n, x, dev):
  """`--train --no-adjoint`: one training iteration of the ODE block at the benchmark shape the way `run_GNN.py --function transformer
  --block constant --method rk4` runs it by default (opt['adjoint'] off: reference run_GNN.py:336, src/base_classes.py:44-47 ->
  torchdiffeq.odeint, loss.backward() through the solver loop).  Here: forward = the recorded native solve (the inference hipGraph with
  every stage input written to a tape slot), backward = the native reverse sweep over the record (4 K VJP stages, one hipGraph);
  A/B against this package's differentiable host loop over the kernel-backed autograd Functions (GNPDE_HOST_FIXED_TRAINING=1: what
  ran before round 6).  K steps forward AND backward inside the timed region; loss = sum(z * c)."""
  d, K, W = cfg['d'], args.steps, args.warmup
  A, h = opt['attention_dim'], opt['heads']
  topt = dict(opt, adjoint=False, time=float(K))
  c = torch.randn(n, d, generator=torch.Generator().manual_seed(args.seed + 3)).to(dev)

  def measure(host_loop, replays):
    block = make_block(G, dict(topt, gnpde_host_fixed_training=host_loop), ei, n, x, dev, float(K), args.seed)
    block.train()

    def iteration():
      for p in block.parameters():
        p.grad = None
      xin = x.clone().requires_grad_(True)
      block.set_x0(xin)
      block.odefunc.nfe = 0
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      z = block(xin)
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      (z * c).sum().backward()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      return t1 - t0, t2 - t1, xin.grad, z.detach()

    for _ in range(max(W, 2) if not host_loop else 1):
      iteration()
    runs = [iteration() for _ in range(max(replays, 1))]
    fw = sorted(r[0] for r in runs)[len(runs) // 2]
    bw = sorted(r[1] for r in runs)[len(runs) // 2]
    f = block.odefunc
    grads = {k: p.grad.detach().clone() for k, p in block.named_parameters() if p.grad is not None}
    res = dict(fw=fw, bw=bw, gx=runs[-1][2].clone(), z=runs[-1][3].clone(), grads=grads, nfe=f.nfe, path=getattr(f, '_last_train_solve', None),
               E=int(f.edge_index.shape[1]))
    del block, runs
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res

  rec = measure(False, args.replays)
  assert torch.isfinite(rec['gx']).all()
  if args.no_host_loop:
    host = {'error': 'skipped (--no-host-loop)'}
  else:
    try:
      host = measure(True, 1)
    except Exception as exc:   # noqa: BLE001
      host = {'error': repr(exc)[:300]}
  # counter traffic of ONE VJP stage: two rocprofv3 --pmc passes over a short child run of this mode (no host-loop A/B in it); the kernels
  # of the reverse sweep do not run in the forward solve, so their totals / the number of recorded evaluations is the stage's traffic
  traffic, traffic_src = None, None
  if not args.no_live_pmc and not args.no_host_loop:
    kc, wc = 2, 2
    pmc = mode_pmc_traffic(args, ['--train', '--no-adjoint', '--no-host-loop', '--steps', str(kc), '--warmup', str(wc), '--graph', args.graph,
                                  '--function', args.function], 'train_no_adjoint')
    if isinstance(pmc, dict) and 'error' not in pmc:
      iters = max(wc, 2) + 1
      S = float(iters * 4 * kc)
      sweep = ('adjoint_rows', 'adjoint_long_reduce', 'permute_f32', 'attention_rows_bwd', 'attention_hub', 'head_rowsum', 'linear_lds', 'stage_combine',
               'adjoint_gram', 'adjoint_dots_fold', 'adjoint_param_fold', 'normalise_heads_bwd', 'att_bwd', 'seg_dot', 'gat_', 'exp_node', 'head_spmm')
      tot, by = 0.0, {}
      for k, v in pmc.items():
        if k.startswith('_') or 'bytes_per_launch' not in v or not any(t in k for t in sweep):
          continue
        by[k] = {'launches': v['launches'], 'bytes_per_launch': round(v['bytes_per_launch']), 'l2_hit_rate': v.get('l2_hit_rate')}
        tot += v['fetch_bytes_total'] + v['write_bytes_total']
      traffic = tot / S
      traffic_src = {'how': 'measured in this run: rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes, --kernel-trace only) '
                            'over a child `bench.py --train --no-adjoint --no-host-loop --steps %d` (%d iterations = %d recorded evaluations swept); bytes = '
                            '(2 FETCH_SIZE + WRITE_SIZE) * 1024 over the kernels of the reverse sweep, per VJP stage' % (kc, iters, int(S)),
                     'live': True, 'seconds': pmc.get('_seconds'), 'kernels': by}
    elif isinstance(pmc, dict):
      traffic_src = {'live_pmc_error': pmc.get('error')}
  E = rec['E']
  agg = E * (8 + 4 * d) + n * (4 + 8 * d) + 4 * d * n
  # algorithmic bytes of ONE VJP stage of the cotangent-side sweep (gather model as SURVEY 8d): the row kernel on the
  # transposed graph (an aggregation + the own recorded row + the edge products written), two 4-byte permutations (weights in, products
  # back), the normaliser backward, d q / d k, P, the combine pass (S, P, u_a, x0 and on average 1.5 stage operands in, the next cotangent
  # out), the Gram pass; neither the projection nor the attention forward (recorded) nor a second aggregation
  b_rows = agg + 4 * E + 4 * d * n
  b_perm = 2 * (E * 12)
  b_attb = E * (4 + 4 + 4 * A + 4 * h) + n * (16 + 4 * A)
  b_dqk = 2 * (E * (4 + 4 * h + 4 * A) + n * (16 + 4 * A)) + E * 4
  b_pg = n * (8 * A + 4 * d)
  b_comb = int(n * 4 * d * 6.5)
  b_gram = n * (8 * A + 4 * d)
  stage_bytes = b_rows + b_perm + b_attb + b_dqk + b_pg + b_comb + b_gram
  t_stage = rec['bw'] / (4 * K)
  parity = None
  if 'error' not in host:
    from oracle import restate as R
    gi, g2 = R.parity_error(rec['gx'], host['gx'])
    zi, z2 = R.parity_error(rec['z'], host['z'])
    # (errors against the largest gradient of the same module: a gradient that is zero in exact arithmetic -- K.bias under a softmax over
    #  rows -- is rounding noise on both sides, as in tests/test_tape_gpu.py)
    worst = 0.0
    scale = {}
    for k, v in host['grads'].items():
      mod = k.rsplit('.', 2)[0] if 'multihead' in k else k
      scale[mod] = max(scale.get(mod, 0.0), float(v.abs().max()))
    for k, v in host['grads'].items():
      mod = k.rsplit('.', 2)[0] if 'multihead' in k else k
      if k in rec['grads'] and scale[mod] > 0:
        worst = max(worst, float((rec['grads'][k] - v).abs().max()) / scale[mod])
    parity = {'grad_x_rel_max': gi, 'grad_x_rel_l2': g2, 'z_rel_max': zi, 'parameter_gradients_rel_max_module_scaled': worst, 'nfe_equal': rec['nfe'] == host['nfe']}
  out = {
    'metric': 'training steps/sec WITHOUT the adjoint method (forward + loss.backward() through the solver), %s d=%d rk4' % (GRAPH_NAMES.get(args.graph, args.graph), d),
    'value': round(K / (rec['fw'] + rec['bw']), 3), 'unit': 'steps/s', 'n_gpus': 1, 'steps': K, 'warmup': W,
    'ms_per_step': round(1e3 * (rec['fw'] + rec['bw']) / K, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
    'dtype': 'f32', 'data': 'synthetic',
    'config': {'workload': workload_name(args.graph, args.function, K, args.method) + '; TRAINING with opt[adjoint] off: recorded solve + native reverse sweep',
               'graph': args.graph, 'nodes': n, 'edges_with_self_loops': E, 'd': d, 'attention_dim': A, 'heads': h, 'adjoint': False,
               'tape_bytes': (4 * K + 1) * n * d * 4 + 4 * K * (n * 2 * A * 4 + E * 4)},
    'train_solve_path': rec['path'],
    'forward_ms': round(1e3 * rec['fw'], 3), 'backward_ms': round(1e3 * rec['bw'], 3), 'vjp_stage_ms': round(1e3 * t_stage, 4),
    'rhs_evals_forward': rec['nfe'],
    'host_loop': host if 'error' in host else {
      'forward_ms': round(1e3 * host['fw'], 3), 'backward_ms': round(1e3 * host['bw'], 3), 'path': host['path'] or 'differentiable host loop (odeint._solve_fixed_host)',
      'steps_per_s': round(K / (host['fw'] + host['bw']), 3)},
    'speedup_vs_host_loop': None if 'error' in host else round((host['fw'] + host['bw']) / (rec['fw'] + rec['bw']), 2),
    'parity_vs_host_loop': parity,
    'roofline': {'kernel': 'one VJP stage of the cotangent-side reverse sweep (permutation, row kernel on the transposed graph, permutation, normaliser backward, d q / d k, P, combine pass, Gram)',
                 'bound': 'hbm', 'achieved': round(stage_bytes / t_stage / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                 'frac': round((traffic if traffic else stage_bytes) / t_stage / 1e9 / HBM_PEAK_GBS, 4),
                 'frac_algorithmic': round(stage_bytes / t_stage / 1e9 / HBM_PEAK_GBS, 4),
                 'frac_traffic': None if traffic is None else round(traffic / t_stage / 1e9 / HBM_PEAK_GBS, 4),
                 'traffic': None if traffic is None else round(traffic), 'traffic_source': traffic_src, 'algorithmic_bytes_per_stage': stage_bytes},
    'cpu_baseline': None,
  }
  print(json.dumps(out))


CONFIG_CHILDREN = (
  # (key, BASELINE.json reference, flags, timeout s, seconds it took in the last evidence run).  Every child is this script in another mode
  # and prints its own full JSON line; the parent keeps a summary.  Ordered by what the line must not lose: the BASELINE configurations
  # first, the variants DESIGN.md quotes last -- a child that would not fit what is left of the budget is skipped and says so.
  ('c1_cora_grand_l_euler_T4', 'configs[0] as named: Cora GRAND-l, euler, step_size 1, T = 4',
   ['--graph', 'cora', '--function', 'laplacian', '--method', 'euler', '--steps', '4', '--warmup', '4', '--no-live-pmc', '--no-hbm-probe', '--replays', '21'], 120, 4),
  ('c1_cora_grand_l_rk4', 'configs[0] shape with rk4 (per-step time over 100 steps)',
   ['--graph', 'cora', '--function', 'laplacian', '--steps', '100', '--warmup', '10', '--no-live-pmc', '--no-hbm-probe'], 120, 4),
  ('c2_cora_grand_nl_rk4_row_softmax', 'configs[1]: Cora GRAND-nl scaled_dot, rk4 (A = 128, 8 heads), softmax over rows',
   ['--graph', 'cora', '--steps', '100', '--warmup', '10', '--no-live-pmc', '--no-hbm-probe'], 120, 5),
  ('c2_cora_grand_nl_rk4_as_run_GNN_runs_it', 'configs[1] with best_params Cora normaliser: squareplus over columns',
   ['--graph', 'cora', '--steps', '100', '--warmup', '10', '--square-plus', '--norm-idx', '1', '--no-live-pmc', '--no-hbm-probe'], 120, 5),
  ('cora_best_params_epoch', 'the reference\'s flagship run (best_params Cora: attention block, Laplacian, dopri5, adjoint=False): s per epoch',
   ['--config', 'cora-epoch', '--steps', '20', '--warmup', '3'], 240, 4),
  ('pubmed_block_adaptive_heun_adjoint', 'best_params Pubmed\'s ODE block in training: dopri5 forward, adjoint_method adaptive_heun (the reference\'s default)',
   ['--config', 'pubmed-adjoint'], 200, 5),
  ('coauthorcs_block_dopri5_adjoint', 'best_params CoauthorCS\'s ODE block in training: dopri5 forward, adjoint_method dopri5',
   ['--config', 'coauthor-adjoint'], 200, 3),
  ('arxiv_block_rk4_adjoint', 'best_params ogbn-arxiv\'s ODE block in training (the reference\'s largest flagship run): hard-attention block, d = 162, dopri5 forward, adjoint_method rk4',
   ['--config', 'arxiv-adjoint'], 300, 6),
  ('c3_training_iteration', 'configs[2] shape, TRAINING: forward + native adjoint backward (rk4 both ways)',
   ['--train', '--steps', '10', '--warmup', '2'], 400, 15),
  ('c3_training_iteration_adjoint_off', 'configs[2] shape, TRAINING as run_GNN.py runs rk4 by default (adjoint off): recorded solve + native reverse sweep, host-loop A/B',
   ['--train', '--no-adjoint', '--steps', '10', '--warmup', '2', '--no-live-pmc'], 300, 5),     # (its counter passes: profiles/r06_train_no_adjoint.json)
  ('c4_arxiv_blend_dopri5', 'configs[3]: ogbn-arxiv BLEND, rewiring block, dopri5',
   ['--config', 'c4', '--warmup', '2'], 400, 34),
  ('c5_rmat_one_gpu', 'configs[4] shape on ONE GPU: R-MAT 2^21 nodes, d = 256 (the 8-GPU run is the driver\'s)',
   ['--graph', 'rmat', '--steps', '4', '--warmup', '1', '--no-hbm-probe'], 900, 72),
  ('c3_normalised_over_columns_squareplus', 'configs[2] shape with attention_norm_idx 1 + squareplus (the reference\'s Cora / Citeseer normaliser at scale)',
   ['--steps', '20', '--warmup', '5', '--norm-idx', '1', '--square-plus', '--no-live-pmc', '--no-hbm-probe', '--parity-only'], 300, 6),
  ('c3_arxiv_relabelling_off', 'configs[2] with the node relabelling switched off (GNPDE_REORDER=0)',
   ['--steps', '20', '--warmup', '5', '--no-live-pmc', '--no-hbm-probe', '--no-cpu-baseline'], 200, 3),
  ('c3_arxiv_flat_no_planted_communities', 'configs[2] shape on a graph WITHOUT community structure (what relabelling gains without planted locality)',
   ['--graph', 'arxiv_flat', '--steps', '20', '--warmup', '5', '--no-live-pmc', '--no-hbm-probe', '--parity-only'], 300, 6),
)


HEADLINE_MAX_BYTES = 4096     # the driver keeps an 8-KB tail of stdout: the LAST line must be the headline object and must fit with room


def _strip_prose(obj, depth=0):
  """Drop the explanatory strings (`*_is`, `what`, `note`, `how`, `sample`) of a bench object: they belong to the detail line."""
  if isinstance(obj, dict):
    return {k: _strip_prose(v, depth + 1) for k, v in obj.items()
            if not (k.endswith('_is') or k.endswith('_note') or k in ('what', 'note', 'how', 'sample', 'command'))}
  if isinstance(obj, list):
    return [_strip_prose(v, depth + 1) for v in obj]
  return obj


def configs_summary(configs):
  """{key: [value, unit, frac | null]} of the `configs` block, small enough for the headline line."""
  if not isinstance(configs, dict):
    return None
  out = {}
  for key, rec in configs.items():
    if not isinstance(rec, dict):
      continue
    if 'error' in rec or 'skipped' in rec:
      out[key] = ['error' if 'error' in rec else 'skipped', None, None]
      continue
    r = rec.get('roofline') or {}
    frac = r.get('frac')
    if frac is None:
      frac = r.get('frac_traffic')
    out[key] = [rec.get('value'), rec.get('unit'), frac]
  return out


def headline(out, configs=None):
  """The compact object of the LAST stdout line: BASELINE.json's metric, `roofline` and `cpu_baseline` by the keys the task's contract
  names, and a summary of the other configurations.  Everything else this run measured is on the earlier `bench_detail` /
  `bench_configs` lines."""
  keep = {k: out.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                  'vs_baseline', 'dtype', 'data') if k in out}
  cfg = out.get('config') or {}
  keep['config'] = {k: cfg.get(k) for k in ('workload', 'graph', 'nodes', 'edges_with_self_loops', 'd', 'attention_dim', 'heads',
                                            'rhs_evals_per_step', 'hipgraph', 'parallelism', 'transport', 'ranks_seen', 'driver', 'edge_cut',
                                            'ranks_share_one_device', 'max_bytes_on_one_link_per_evaluation', 'finite',
                                            'sharded_vs_unpartitioned_timed_solve_rel_max') if k in cfg}
  r = out.get('roofline')
  if isinstance(r, dict):
    rk = {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_algorithmic', 'frac_traffic', 'traffic', 'avg_launch_us',
                                'algorithmic_bytes_per_launch', 'algorithmic_bytes_per_stage') if k in r}
    rk['kernel'] = str(r.get('kernel', ''))[:80]
    src = r.get('traffic_source')
    if isinstance(src, dict):
      rk['traffic_live'] = bool(src.get('live')) and not src.get('stale', False)
      if src.get('l2_hit_rate') is not None:
        rk['l2_hit_rate'] = src.get('l2_hit_rate')
    probe = r.get('hbm_bound_probe')
    if isinstance(probe, dict) and probe.get('frac') is not None:
      rk['hbm_bound_probe_frac'] = probe.get('frac')
    sec = []
    for ent in r.get('secondary') or []:
      if isinstance(ent, dict) and 'avg_us' in ent:
        sec.append({'kernel': str(ent.get('kernel', ''))[:40], 'avg_us': ent.get('avg_us'),
                    'frac_algorithmic': None if ent.get('gbs') is None else round(ent['gbs'] / HBM_PEAK_GBS, 4),
                    'frac_traffic': ent.get('frac_traffic')})
    if sec:
      rk['secondary'] = sec
    keep['roofline'] = rk
  else:
    keep['roofline'] = None
  cb = out.get('cpu_baseline')
  keep['cpu_baseline'] = None if not isinstance(cb, dict) else {k: cb.get(k) for k in ('value', 'unit', 'cores', 'kind', 'ms_per_rhs_eval') if k in cb}
  if isinstance(cb, dict) and cb.get('sample'):
    keep['cpu_baseline']['sample'] = str(cb['sample'])[:160]
  for k in ('parity_vs_oracle_one_eval', 'parity_vs_oracle_row_subset', 'speedup_vs_cpu', 'timing'):
    if k in out:
      keep[k] = _strip_prose(out[k])
  summ = configs_summary(configs)
  if summ is not None:
    keep['configs_summary'] = summ
    keep['configs_seconds'] = configs.get('_seconds') if isinstance(configs, dict) else None
  keep['detail'] = 'earlier stdout lines {"bench_detail": ...} and {"bench_configs": ...} of this run'
  line = json.dumps(keep)
  # never let the line outgrow the driver's tail: shed the optional parts, largest first
  for victim in ('configs_summary', 'timing', 'parity_vs_oracle_row_subset'):
    if len(line) < HEADLINE_MAX_BYTES:
      break
    if victim == 'configs_summary' and isinstance(keep.get(victim), dict):
      keep[victim] = {k: v[0] for k, v in keep[victim].items()}
    else:
      keep.pop(victim, None)
    line = json.dumps(keep)
  if len(line) >= HEADLINE_MAX_BYTES and isinstance(keep.get('roofline'), dict):
    keep['roofline'].pop('secondary', None)
    keep.pop('configs_summary', None)
    line = json.dumps(keep)
  return keep


def emit(out, configs=None):
  """Print what the run measured: the full object and the `configs` block each on a line of its own, then -- LAST, so that it is
  what the driver's tail of stdout ends with -- the compact headline object."""
  print(json.dumps({'bench_detail': out}))
  if configs is not None:
    print(json.dumps({'bench_configs': configs}))
  sys.stdout.flush()
  print(json.dumps(headline(out, configs)))
  sys.stdout.flush()


def summarise_child(line):
  """What the parent keeps of a child's bench line."""
  r = line.get('roofline') or {}
  keep = {k: line.get(k) for k in ('metric', 'value', 'unit', 'steps', 'ms_per_step', 'higher_is_better') if k in line}
  cfg = line.get('config') or {}
  keep['workload'] = cfg.get('workload')
  for k in ('nodes', 'edges_with_self_loops', 'd', 'attention_dim', 'heads', 'attention_norm_idx', 'square_plus', 'native_adjoint_solver'):
    if k in cfg:
      keep[k] = cfg[k]
  rel = cfg.get('node_relabelling')
  if isinstance(rel, dict):
    keep['node_relabelling'] = {k: rel.get(k) for k in ('order', 'n_parts', 'entries_inside_a_part', 'aggregation_speedup_measured', 'solve_equal_to_unrelabelled_bitwise') if k in rel}
  elif 'node_relabelling' in cfg:
    keep['node_relabelling'] = None
  for k in ('parity_vs_oracle_one_eval', 'parity_vs_restated_torchdiffeq', 'parity_vs_oracle_row_subset', 'parity_vjp_one_eval_vs_oracle', 'speedup_vs_cpu',
            'ms_per_forward', 'ms_solve_only', 'rhs_evals_per_forward', 'dopri5', 'forward_ms', 'backward_ms', 'f_plus_vjp_ms', 'flat_host_loop',
            'backward_speedup_vs_flat_host_loop', 'parity_vs_flat_host_loop', 'evals_forward', 'augmented_evals_backward', 'ms_train_step', 'ms_test_step',
            'ms_train_phases_synchronised', 'nfe_forward_per_epoch', 'nfe_backward_per_epoch', 'nfe_test_per_epoch', 'train_solve_path', 'timing',
            'host_loop', 'speedup_vs_host_loop', 'vjp_stage_ms'):
    if k in line:
      v = line[k]
      if isinstance(v, dict) and 'what' in v:
        v = {a: b for a, b in v.items() if a != 'what'}
      keep[k] = v
  if r:
    keep['roofline'] = {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_algorithmic', 'frac_traffic', 'traffic', 'avg_launch_us',
                                              'algorithmic_bytes_per_launch', 'algorithmic_bytes_per_stage') if k in r}
    keep['roofline']['kernel'] = str(r.get('kernel', ''))[:90]
  else:
    keep['roofline'] = None
  cb = line.get('cpu_baseline')
  keep['cpu_baseline'] = None if not cb else {k: cb.get(k) for k in ('value', 'unit', 'cores', 'kind') if k in cb}
  return keep


def run_configs(args, budget_s):
  """The other BASELINE configurations (and the variants DESIGN.md quotes), each measured NOW by a child process of this script
  on the same GPU, so that the one line the driver records witnesses them all."""
  import subprocess
  t_start = time.perf_counter()
  out = {}
  for key, what, flags, limit, expect in CONFIG_CHILDREN:
    left = budget_s - (time.perf_counter() - t_start)
    if left < 1.15 * expect + 5:      # (a child cut off by the budget would have spent its time for nothing)
      out[key] = {'skipped': 'the configs block had %.0f s of its %.0f s budget left, this child takes about %d s: run `python bench.py %s`'
                             % (max(left, 0.0), budget_s, expect, ' '.join(flags)), 'what': what}
      continue
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--no-configs', '--seed', str(args.seed)] + flags
    env = dict(os.environ)
    if key == 'c3_arxiv_relabelling_off':
      env['GNPDE_REORDER'] = '0'
    t0 = time.perf_counter()
    try:
      res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=min(limit, left))
      line = None
      for ln in reversed(res.stdout.splitlines()):   # the child's full object: its `bench_detail` line, or the one line of the other modes
        if ln.startswith('{"bench_detail"'):
          line = json.loads(ln)['bench_detail']
          break
        if ln.startswith('{"metric"') and line is None:
          line = json.loads(ln)
      if line is None:
        out[key] = {'error': 'rc %d, no bench line; stderr tail: %s' % (res.returncode, (res.stderr or '')[-300:]), 'what': what}
      else:
        out[key] = dict(summarise_child(line), what=what, command='python bench.py ' + ' '.join(flags), seconds=round(time.perf_counter() - t0, 1))
    except subprocess.TimeoutExpired:
      out[key] = {'error': 'timed out after %.0f s' % min(limit, left), 'what': what}
    except Exception as exc:   # noqa: BLE001
      out[key] = {'error': repr(exc)[:300], 'what': what}
  out['_seconds'] = round(time.perf_counter() - t_start, 1)
  return out


def launch_ranks(n_ranks):
  """`python bench.py --gpus N` without a launcher: run this script as N ranks under torch.distributed.run (one process per GPU, RCCL) on
  a free local port and pass the ranks' stdout through, so that rank 0's headline stays the last line.  With fewer than N devices the
  ranks can only share one (GNPDE_RANKS_SHARE_DEVICE=1: a functional run, the line says so) -- never silently."""
  import socket
  import subprocess
  have = torch.cuda.device_count()
  if have < n_ranks and os.environ.get('GNPDE_RANKS_SHARE_DEVICE', '0') != '1':
    raise SystemExit('--gpus %d but this host shows %d HIP device(s); set GNPDE_RANKS_SHARE_DEVICE=1 for a functional run of the %d-rank '
                     'path on one device (not a scaling measurement)' % (n_ranks, have, n_ranks))
  with socket.socket() as sock:
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_ranks), '--master-addr', '127.0.0.1',
         '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  res = subprocess.run(cmd, env=env)
  if res.returncode != 0:
    raise SystemExit(res.returncode)


def main():
  args = parse()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus > 1 and 'RANK' not in os.environ:
    return launch_ranks(args.gpus)        # `python bench.py --gpus N` as the driver runs `--gpus 1`: spawn the N ranks ourselves
  if args.gpus != world:
    raise SystemExit('--gpus %d but WORLD_SIZE is %d' % (args.gpus, world))
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
    return D.bench_main(args, rank, world, dev, emit=emit)

  if args.config == 'c4':
    return c4_main(G, args, dev)
  if args.config == 'cora-epoch':
    return cora_epoch_main(G, args, dev)
  if args.config in ('pubmed-adjoint', 'coauthor-adjoint', 'arxiv-adjoint'):
    return pubmed_adjoint_main(G, args, dev)
  cfg = G.synthetic.CONFIGS[args.graph]
  ei_cpu, n = G.synthetic.make_graph(args.graph, seed=args.seed, scale=args.scale)
  opt = build_opt(cfg, args)
  d = cfg['d']
  x_cpu = torch.randn(n, d, generator=torch.Generator().manual_seed(args.seed))
  x = x_cpu.to(dev)
  ei = ei_cpu.to(dev)
  K, W = args.steps, args.warmup
  use_graph = not args.no_graph

  if os.environ.get('GNPDE_BENCH_REORDER'):      # the PMC child runs on the node order its parent's solver chose
    opt['gnpde_reorder'] = os.environ['GNPDE_BENCH_REORDER']
  main_block = make_block(G, opt, ei, n, x, dev, float(K), args.seed)
  main_block.set_x0(x)
  if args.pmc_child:
    with torch.no_grad():
      pmc_child(G, main_block, x)
    return
  if args.train:
    del main_block
    if args.no_adjoint:
      return train_no_adjoint_main(G, args, opt, cfg, ei, n, x, dev)
    return train_main(G, args, opt, cfg, ei, n, x, dev)
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
  evals_per_step = 4 if args.method == 'rk4' else 1
  f = main_block.odefunc
  E = int(f.edge_index.shape[1])
  A, h = opt['attention_dim'], opt['heads']

  # roofline of the dominant kernel (B_spmm = E (4 + 4 + 4d) + N (4 + 8d) + 4dN with add_source)
  if args.no_roofline_probe:
    print(json.dumps({'metric': metric_name(args.graph, d, method=args.method), 'value': round(steps_per_s, 3), 'unit': 'steps/s', 'n_gpus': 1,
                      'steps': K, 'warmup': W, 'ms_per_step': round(1e3 * elapsed / K, 4), 'higher_is_better': True,
                      'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                      'config': {'workload': workload_name(args.graph, args.function, K, args.method), 'graph': args.graph, 'nodes': n,
                                 'edges_with_self_loops': E, 'd': d, 'hipgraph': use_graph},
                      'roofline': None, 'cpu_baseline': None, 'note': 'profiling run: --no-roofline-probe'}))
    return
  t_spmm, graph, kname, fused = dominant_kernel_time(G, main_block, x)
  bytes_l = E * (8 + 4 * d) + n * (4 + 8 * d) + 4 * d * n                      # SURVEY.md 8d, B_l + source
  bytes_nl = E * (4 + 4 * A + 4 * d) + n * (4 + 12 * A + 12 * d) + 4 * d * n   # SURVEY.md 8d, B_nl + source
  bytes_spmm = bytes_nl if fused else bytes_l
  achieved = bytes_spmm / t_spmm / 1e9
  ceiling = gather_ceiling(G, x, n, E, graph)          # informational only (a kernel of this repository, not a roofline)
  try:
    stream = stream_read_probe(x)
  except Exception as exc:   # noqa: BLE001
    stream = {'error': repr(exc)[:200]}
  try:
    secondary = [] if fused else secondary_kernels(G, main_block, x, E, n, ceiling)
  except Exception as exc:   # (a probe that cannot run must not cost the line)
    secondary = [{'error': repr(exc)[:200]}]
  bytes_eval = bytes_nl if args.function == 'transformer' else bytes_l
  # compulsory DRAM bytes of one aggregation launch with perfect reuse of gathered rows (SURVEY 8d B_min without the
  # projection): colidx + w + one read of u + the per-row streams (x0 in, out)
  dram_floor = E * 8 + n * (4 + 12 * d)
  state_mb = n * d * 4 / 2 ** 20
  resident = bool(state_mb < 256)
  _, view = solver_graph(f, x)
  reorder_mode = '0' if view is None else ('degree' if 'descending' in str(view.stats.get('order', '')) else 'parts')

  # ---- counter traffic: live PMC passes over a child of this script; the stored record only as a marked fallback ----
  now_hash = kernel_sources_sha16()
  traffic, traffic_src = None, None
  pmc = None if (args.no_live_pmc or fused) else live_pmc_traffic(args, reorder_mode, timeout_s=150 if args.graph != 'rmat' else 420)
  if isinstance(pmc, dict) and 'error' not in pmc and pmc.get('_calls'):
    agg = traffic_of(pmc, 'spmm_', pmc['_calls']['aggregation_calls'])
    if agg is not None:
      traffic = agg['bytes_per_call']
      traffic_src = {'how': 'measured in this run: rocprofv3 --pmc (two passes: FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, --kernel-trace '
                            'only) over a child process of bench.py that launches the kernels of %d evaluations in the solver\'s order (projection, row attention, aggregation per stage) on the same graph; '
                            'bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024 per call of the aggregation (row kernel + hub-row fold)'
                            % pmc['_calls']['evaluations'],
                     'live': True, 'stale': False, 'seconds': pmc.get('_seconds'), 'fetch_bytes': round(agg['fetch_bytes_per_call']),
                     'write_bytes': round(agg['write_bytes_per_call']), 'l2_hit_rate': agg['l2_hit_rate'], 'kernels': agg['kernels'],
                     'kernel_sources_sha16': now_hash}
      for ent, pat, ncalls in zip(secondary, ('row_attention', 'linear_'), (pmc['_calls']['evaluations'],) * 2):
        rec = traffic_of(pmc, pat, ncalls)
        if rec is not None and 'avg_us' in ent:
          ent['traffic'] = round(rec['bytes_per_call'])
          ent['traffic_gbs'] = round(rec['bytes_per_call'] / (ent['avg_us'] * 1e-6) / 1e9, 1)
          ent['frac_traffic'] = round(ent['traffic_gbs'] / HBM_PEAK_GBS, 4)
          ent['l2_hit_rate'] = [v.get('l2_hit_rate') for v in rec['kernels'].values()]
          ent['traffic_is'] = 'live PMC passes of this run'
  pmc_error = pmc.get('error') if isinstance(pmc, dict) and 'error' in pmc else None
  if traffic is None:
    tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    try:
      stored = json.load(open(tpath)).get('%s_d%d_%s' % (args.graph, d, 'fused' if fused else 'spmm'))
    except Exception:   # noqa: BLE001
      stored = None
    # (the record belongs to ONE node order: a run on another order -- relabelling off, or decided differently -- has no stored traffic)
    if isinstance(stored, dict) and stored.get('bytes_per_launch') and stored.get('node_order', 'parts') == reorder_mode:
      traffic = stored['bytes_per_launch']
      traffic_src = {k: stored.get(k) for k in ('kernel', 'commit', 'fetch_bytes', 'write_bytes', 'l2_hit_rate', 'method')}
      rec_hash = stored.get('kernel_sources_sha16')
      traffic_src.update(how='STORED record of profiles/hbm_traffic.json (an earlier run of tools/pmc_traffic.py), not measured in this run',
                         live=False, kernel_sources_sha16=now_hash, record_kernel_sources_sha16=rec_hash,
                         # stale unless the record carries the hash of ALL kernel sources as they are now (spmm / attention / linear / epilogue)
                         stale=bool(rec_hash is None or now_hash is None or rec_hash != now_hash))
      if pmc_error:
        traffic_src['live_pmc_error'] = pmc_error
  elif pmc_error:
    traffic_src['live_pmc_error'] = pmc_error

  # What `achieved`, `peak` and `frac` are (the task's contract):
  #   achieved         = algorithmic (gather-model) bytes of one aggregation call / its measured duration
  #   peak             = HBM3E's 8 TB/s
  #   frac_algorithmic = achieved / peak.  Exceeds 1 when the gathered table sits in the 256-MiB Infinity Cache (ogbn-arxiv: 83 MiB):
  #                      the model charges every cache-served row to HBM.
  #   frac_traffic     = counter bytes that crossed the L2 -> fabric boundary (Infinity-Cache hits included) / duration / peak
  #   frac             = frac_traffic for a cache-resident table (bound "mall"), frac_algorithmic for a DRAM-resident one (bound "hbm");
  #                      without any counter record, frac_algorithmic in both cases -- never a ceiling of our own making.
  frac_alg = achieved / HBM_PEAK_GBS
  frac_traffic = None if traffic is None else traffic / t_spmm / 1e9 / HBM_PEAK_GBS
  # (a live counter record is the physical figure for a DRAM-resident table as well: there the gather model credits no reuse of the hub
  #  columns, which are L2 hits, and exceeds what crossed the fabric)
  use_traffic = frac_traffic is not None and not traffic_src.get('stale', True) and (resident or bool(traffic_src.get('live')))
  try:
    hbm_probe = None if args.no_hbm_probe else hbm_bound_probe(G, dev)
  except Exception as exc:   # noqa: BLE001
    hbm_probe = {'error': repr(exc)[:200]}
  out = {
    'metric': metric_name(args.graph, d, method=args.method),
    'value': round(steps_per_s, 3), 'unit': 'steps/s', 'n_gpus': 1, 'steps': K, 'warmup': W,
    'ms_per_step': round(1e3 * elapsed / K, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
    'dtype': 'f32', 'data': 'synthetic',
    'timing': {'replays': len(times), 'statistic': 'median', 'ms_per_step_min': round(1e3 * min(times) / K, 4),
               'ms_per_step_max': round(1e3 * max(times) / K, 4)},
    'config': {'workload': workload_name(args.graph, args.function, K, args.method),
               'graph': args.graph, 'nodes': n, 'edges_with_self_loops': E, 'd': d, 'attention_dim': A, 'heads': h,
               'rhs_evals_per_step': evals_per_step, 'hipgraph': use_graph, 'scale': args.scale,
               'attention_norm_idx': args.norm_idx, 'square_plus': args.square_plus,
               'early_stop_evaluator': bool(args.early_stop),
               'long_rows': graph.n_long_rows,
               # how the aggregation launches deal the rows to the 8 XCDs (gnpde_graph_t.xcd_deal, chosen per graph from the
               # measured imbalance of contiguous eighths; GNPDE_TUNE=10=1 / 10=2 force one or the other for A/B runs)
               'xcd_row_deal': {0: 'hashed_blocks' if graph.struct.xcd_deal == 1 else 'contiguous_eighths',
                                1: 'contiguous_eighths (forced)', 2: 'hashed_blocks (forced)'}.get(xcd_knob, '?'),
               'xcd_contiguous_imbalance': round(graph.xcd_imbalance_contiguous, 4),
               'algorithmic_bytes_per_rhs_eval': bytes_eval,
               'eval_gbs_vs_gather_model': round(bytes_eval * evals_per_step * steps_per_s / 1e9, 1)},
    'roofline': {'kernel': kname, 'bound': 'mall' if resident else 'hbm',
                 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                 # cache-resident table: `frac` is the counter figure or nothing -- the gather model exceeds the peak there and must not
                 # pass for a roofline fraction (`frac_algorithmic` stays next to it); DRAM-resident table: the algorithmic fraction
                 'frac': round(frac_traffic, 4) if use_traffic else (None if (resident or frac_alg > 1.0) else round(frac_alg, 4)),
                 'frac_is': ('frac_traffic: L2 -> fabric bytes of one aggregation call (counter record, `traffic`; includes Infinity-Cache hits) / '
                             'its duration / the 8 TB/s HBM peak -- the gathered table (%.0f MiB) is resident in the 256-MiB Infinity Cache, so '
                             'the gather model (`achieved`, `frac_algorithmic`) charges cache-served rows to HBM and exceeds the peak' % state_mb)
                 if use_traffic and resident else ('frac_traffic: L2 -> fabric bytes of one aggregation call (live counter passes of this run) / its duration / '
                                                   'the 8 TB/s HBM peak; the %.0f-MiB table does not fit the Infinity Cache: DRAM-bound' % state_mb)
                 if use_traffic else ('null: the gathered table is cache-resident and this run has no live counter traffic (see frac_algorithmic, '
                                      'which exceeds 1 by construction there)' if resident else
                                      'null: the gather model (frac_algorithmic) exceeds the peak -- it credits no reuse, and the hub columns of this '
                                      'graph are L2 hits; no live counter traffic in this run' if frac_alg > 1.0 else
                                      'frac_algorithmic: algorithmic bytes / duration / the 8 TB/s HBM peak (SURVEY 8d)'),
                 'frac_algorithmic': round(frac_alg, 4),
                 'frac_algorithmic_is': 'gather model (every non-zero fetches its neighbour row, no cache reuse credited) / duration / 8 TB/s'
                                        + ('; table cache-resident: not a fraction of anything physical' if resident else ''),
                 'frac_traffic': None if frac_traffic is None else round(frac_traffic, 4),
                 'traffic': None if traffic is None else round(traffic),
                 'traffic_gbs': None if traffic is None else round(traffic / t_spmm / 1e9, 1),
                 'traffic_is': 'L2 -> fabric bytes per aggregation call (includes Infinity-Cache hits: MI355X_MICROARCH.md section HBM)',
                 'traffic_source': traffic_src,
                 'achieved_is': 'algorithmic (gather-model) bytes of one aggregation call / its average duration over the four rk4 stage '
                                'variants, timed in this run (HIP events around a captured graph of the launches on the launch stream)',
                 'algorithmic_bytes_per_launch': bytes_spmm, 'avg_launch_us': round(t_spmm * 1e6, 2),
                 'hbm_copy_rate': HBM_COPY_GBS, 'frac_algorithmic_of_copy_rate': round(achieved / HBM_COPY_GBS, 4),
                 'compulsory_gather_bytes_per_launch': E * (4 + 4 * d) + n * (4 + 12 * d),
                 'dram_floor_bytes': dram_floor, 'frac_dram_floor': round(dram_floor / t_spmm / 1e9 / HBM_PEAK_GBS, 4),
                 'gathered_table_mib': round(state_mb, 1), 'table_fits_infinity_cache': resident,
                 'stream_read_probe': stream,
                 'hbm_bound_probe': hbm_probe,
                 'row_gather_gbs': round(E * 4 * d / t_spmm / 1e9, 1),
                 'own_gather_kernel': None if ceiling is None else dict(
                   ceiling, note='informational: a kernel of this repository (balanced gather of the same column ids), NOT the roofline'),
                 'epilogue_stream_bytes_per_launch_avg': 4 * d * n * 3,   # x0 + out every stage, y and k1 in two of four: 12 per step
                 'secondary': secondary},
  }
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
    try:
      out['parity_vs_oracle_row_subset'] = subset_parity(main_block, x, x_cpu)
    except Exception as exc:   # noqa: BLE001
      out['parity_vs_oracle_row_subset'] = {'error': repr(exc)[:200]}
  elif args.parity_only:
    # (children of the default run: the timed baseline is the headline's; here only the parity figure of this configuration)
    ref = cpu_baseline(main_block, x_cpu, 0)[1]
    with torch.no_grad():
      f.x0 = x
      got = f(0.0, x)
    from oracle import restate as R
    e_inf, e_2 = R.parity_error(got, ref)
    out['parity_vs_oracle_one_eval'] = {'rel_max': e_inf, 'rel_l2': e_2}
  elif not args.no_cpu_baseline:
    t_eval, ref, thread_trials = cpu_baseline(main_block, x_cpu, cpu_evals)
    with torch.no_grad():
      f.x0 = x
      got = f(0.0, x)
    from oracle import restate as R
    e_inf, e_2 = R.parity_error(got, ref)
    out['cpu_baseline'] = dict(host_info(), **{
      'value': round(1.0 / (evals_per_step * t_eval), 4), 'unit': 'steps/s', 'cores': torch.get_num_threads(),
      'threads_used': torch.get_num_threads(), 'kind': 'port',
      'kind_note': 'oracle/restate.py = the reference op sequence (index_select -> mul -> scatter_add, PyG softmax) in '
                   'torch CPU; the reference src/ itself needs /root/reference and third-party wheels that do not exist '
                   'on the GPU box',
      'sample': '%d full-size evaluations of f (= %.1f solver steps) of the same workload at the torch thread count with the best median '
                'of three evaluations (`thread_trials_ms`: median ms per evaluation by thread count); steps/s = 1 / (%d t_eval)' % (cpu_evals, cpu_evals / float(evals_per_step), evals_per_step),
      'thread_trials_ms': thread_trials,
      'ms_per_rhs_eval': round(t_eval * 1e3, 2)})
    out['parity_vs_oracle_one_eval'] = {'rel_max': e_inf, 'rel_l2': e_2}
    out['speedup_vs_cpu'] = round(steps_per_s / out['cpu_baseline']['value'], 1)
  default_run = (args.graph == 'arxiv' and args.function == 'transformer' and args.method == 'rk4' and args.norm_idx == 0 and not args.square_plus
                 and not args.early_stop and args.scale == 1.0 and use_graph)
  if default_run and not args.no_configs:
    # free this process's device memory first: the children run on the same GPU, one at a time
    del main_block, z, x, ei
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    budget = args.configs_budget
    if budget is None:
      budget = max(90.0, DEFAULT_RUN_SECONDS - (time.perf_counter() - T_PROCESS_START))
    configs = run_configs(args, budget)
  else:
    configs = None
  emit(out, configs)


if __name__ == '__main__':
  main()
