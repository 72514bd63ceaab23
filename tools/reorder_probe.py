"""Does relabelling the nodes by community pay?  (GPU box only.)  The arxiv-shaped stand-in (40 communities of Zipf sizes, 65 % of
the edges inside a community, ids shuffled): attention + aggregation on the graph as given, relabelled by the generator's own
communities (the best any clustering could do), and relabelled by the native label-propagation partitioner with P parts.
Entry order inside a row is kept, so the sums are bit-identical up to the row permutation."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib
from gnpde_amd.graph import partition_rows, LocalityView
import bench

dev = torch.device('cuda:0')
cfg = G.synthetic.CONFIGS['arxiv']
n = cfg['n']
ei_np, relabel, comm = G.synthetic.community_powerlaw_graph(n, cfg['pairs'], 0)
ei = torch.as_tensor(ei_np) if not torch.is_tensor(ei_np) else ei_np
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
d, A, h = cfg['d'], 16, 4
gen = torch.Generator().manual_seed(3)
x = torch.randn(n, d, generator=gen)
wqk = (torch.randn(2 * A, d, generator=gen) / d ** 0.5).to(dev)
alpha, beta = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
comm_of = np.empty(n, dtype=np.int64)
comm_of[np.asarray(relabel)] = np.asarray(comm)
base = G.CSRGraph(ei2.to(dev), n)
rowptr, colidx = base.t['rowptr'].cpu(), base.t['colidx'].cpu()


def run(label, order, extra=None, knob=0):
  """order: new position -> old node id (None: as given)."""
  if order is None:
    e, xs = ei2, x
    inv = None
  else:
    order = torch.as_tensor(order, dtype=torch.int64)
    inv = torch.empty(n, dtype=torch.int64)
    inv[order] = torch.arange(n)
    e, xs = inv[ei2], x[order]
  if order is None:
    g = G.CSRGraph(e.to(dev), n)
  else:
    view = LocalityView(base, order, {})       # relabelled CSR in the caller's edge order
    g = view.graph
  ops.tune(_lib.TUNE_XCD_ROWS, knob)
  extra = dict(extra or {}, xcd_deal={0: ['contiguous', 'hashed'][g.struct.xcd_deal], 1: 'contiguous (forced)', 2: 'hashed (forced)'}[knob],
               xcd_imbalance_contiguous=round(g.xcd_imbalance_contiguous, 3))
  xd = xs.to(dev)
  x0 = xd.clone()
  out = torch.empty_like(xd)
  qk = ops.linear(xd, wqk, torch.zeros(2 * A, device=dev))
  st = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False, q=qk, k=qk[:, A:], ldqk=2 * A)
  w = ops.edge_attention(g, st, True, False, False, like=xd)[0]
  t_att = bench.timed_replay(lambda: ops.edge_attention(g, st, True, False, False, like=xd), 8)
  t_agg = bench.timed_replay(lambda: ops.spmm_rhs(g, w, xd, alpha, beta, x0, True, out=out), 8)
  t_lin = bench.timed_replay(lambda: ops.linear(xd, wqk, None, out=qk), 8)
  res = {'order': label, 'attention_us': round(t_att * 1e6, 1), 'aggregation_us': round(t_agg * 1e6, 1), 'projection_us': round(t_lin * 1e6, 1),
         'sum_us': round((t_att + t_agg + t_lin) * 1e6, 1)}
  ops.spmm_rhs(g, w, xd, alpha, beta, x0, True, out=out)
  res['_out'] = out.cpu() if inv is None else out.cpu()[inv]
  ops.tune(_lib.TUNE_XCD_ROWS, 0)
  if extra:
    res.update(extra)
  return res


ref = run('as given (shuffled ids)', None)
ref_out = ref.pop('_out')
print(json.dumps(ref), flush=True)
r = run('by the generator\'s communities', np.argsort(comm_of, kind='stable'))
print(json.dumps(dict(r, equal_to_as_given=bool(torch.equal(r.pop('_out'), ref_out)))), flush=True)
src = ei2[0].numpy()
dst = ei2[1].numpy()
deg = np.bincount(src, minlength=n)
for P in (8, 16, 32):
  t0 = time.perf_counter()
  part = partition_rows((rowptr, colidx), P, refine_links=0).numpy()
  dt = time.perf_counter() - t0
  inside = float((part[src] == part[dst]).mean())
  r = run('label-propagation partitioner, %d parts' % P, np.argsort(part, kind='stable'),
          {'entries_inside_a_part': round(inside, 3), 'partition_seconds': round(dt, 2)})
  print(json.dumps(dict(r, equal_to_as_given=bool(torch.equal(r.pop('_out'), ref_out)))), flush=True)
  if P == 16:
    for label, key in (('rows by ascending length inside a part', deg), ('rows by descending length inside a part', -deg)):
      r = run('%d parts, %s' % (P, label), np.lexsort((key, part)), {'entries_inside_a_part': round(inside, 3)})
      print(json.dumps(dict(r, equal_to_as_given=bool(torch.equal(r.pop('_out'), ref_out)))), flush=True)
r = run('no parts, rows by ascending length', np.argsort(deg, kind='stable'))
print(json.dumps(dict(r, equal_to_as_given=bool(torch.equal(r.pop('_out'), ref_out)))), flush=True)
