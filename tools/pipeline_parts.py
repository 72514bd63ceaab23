"""Can the projection and the attention hide behind the aggregation?  (GPU box only.)  One evaluation of f is the chain
projection -> attention -> aggregation; the aggregation is bound by bytes crossing the fabric, the other two by latency / issue.
Rows cut into P parts of equal entry counts, each part a chain att(p) -> agg(p) -> proj'(p) on its own stream, the next
evaluation's attention waiting for ALL projections: att / proj of one part can run while another part aggregates.
Prints microseconds per evaluation: unsplit serial, split serial, split on P streams (staggered or not)."""
import ctypes
import faulthandler
faulthandler.enable()
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnpde_amd as G
from gnpde_amd import ops, _lib

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'arxiv'
parts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '2,4').split(',')]
cfg = G.synthetic.CONFIGS[name]
ei, n = G.synthetic.make_graph(name)
ei2, _ = G.add_remaining_self_loops(ei, None, 1.0, n)
ei2 = ei2.to(dev)
d, A, h = cfg['d'], 16, 4
L = _lib.lib()
gen = torch.Generator().manual_seed(3)
wqk = (torch.randn(2 * A, d, generator=gen) / d ** 0.5).to(dev)
bqk = torch.zeros(2 * A, device=dev)
x0 = torch.randn(n, d, generator=gen).to(dev)
u = [x0.clone(), torch.empty_like(x0)]
qk = [torch.empty(n, 2 * A, device=dev) for _ in range(2)]
alpha, beta = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
st = [ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False, q=q, k=q[:, A:], ldqk=2 * A) for q in qk]
full = G.CSRGraph(ei2, n)
E = full.e


def att(graph, s, w):
  ws = graph.workspace('att', L.gnpde_attention_workspace_bytes(graph.ref(), ctypes.byref(s)))
  _lib.check(L.gnpde_edge_attention(graph.ref(), ctypes.byref(s), _lib.ptr(w), None, None, _lib.ptr(ws), ws.numel(), _lib.stream_of(w)))


def views(P):
  rowptr = full.t['rowptr'].to(torch.int64)
  cuts = [0] + [int(torch.searchsorted(rowptr, torch.tensor(E * c // P, device=dev)).item()) for c in range(1, P)] + [n]
  out = []
  for p in range(P):
    lo, hi = cuts[p], cuts[p + 1]
    m = (ei2[0] >= lo) & (ei2[0] < hi)
    g = G.CSRGraph(ei2[:, m], n)
    g.set_row_range(lo, hi)
    out.append((lo, hi, g, torch.empty(max(g.e, 1), device=dev)))
  return out


def capture(body, n_evals):
  body(2)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    body(n_evals)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  best = None
  for _ in range(4):
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1)
    best = t if best is None or t < best else best
  return best * 1e3 / n_evals, u[n_evals % 2].clone()


w_full = torch.empty(E, device=dev)


def serial_unsplit(n_evals):
  for e in range(n_evals):
    uin, uout = u[e % 2], u[(e + 1) % 2]
    ops.linear(uin, wqk, bqk, out=qk[e % 2])
    att(full, st[e % 2], w_full)
    ops.spmm_rhs(full, w_full, uin, alpha, beta, x0, True, out=uout)


def reset():
  u[0].copy_(x0)


NE = 16
reset()
t0, ref = capture(serial_unsplit, NE)
print(json.dumps({'what': 'one stream, whole graph: projection, attention, aggregation', 'us_per_evaluation': round(t0, 1)}), flush=True)

for P in parts:
  V = views(P)
  S = [torch.cuda.Stream() for _ in range(P)]

  def serial_split(n_evals):
    for e in range(n_evals):
      uin, uout = u[e % 2], u[(e + 1) % 2]
      ops.linear(uin, wqk, bqk, out=qk[e % 2])
      for (lo, hi, g, w) in V:
        att(g, st[e % 2], w)
      for (lo, hi, g, w) in V:
        ops.spmm_rhs(g, w, uin, alpha, beta, x0, True, out=uout)

  def piped(n_evals, stagger):
    """Part 0 on the launch stream, the others on side streams forked at the start of every evaluation and joined at its end
    (the next evaluation's attention needs every projection anyway)."""
    main = torch.cuda.current_stream()
    ops.linear(u[0], wqk, bqk, out=qk[0])
    for e in range(n_evals):
      uin, uout = u[e % 2], u[(e + 1) % 2]
      prev_att = None
      for p, (lo, hi, g, w) in enumerate(V):
        sp = main if p == 0 else S[p]
        if p:
          sp.wait_stream(main) if not stagger else sp.wait_event(prev_att)
        with torch.cuda.stream(sp):
          att(g, st[e % 2], w)
          if stagger:
            prev_att = torch.cuda.Event()
            prev_att.record(sp)
          ops.spmm_rhs(g, w, uin, alpha, beta, x0, True, out=uout)
          ops.linear(uout[lo:hi], wqk, bqk, out=qk[(e + 1) % 2][lo:hi])
      for sp in S[1:]:
        main.wait_stream(sp)

  def eager(body, n_evals):
    body(2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(4):
      torch.cuda.synchronize()
      e0.record()
      body(n_evals)
      e1.record()
      torch.cuda.synchronize()
      t = e0.elapsed_time(e1)
      best = t if best is None or t < best else best
    return best * 1e3 / n_evals, u[n_evals % 2].clone()

  reset()
  t1, y1 = capture(serial_split, NE)
  res = {'parts': P, 'rows': [(v[0], v[1]) for v in V], 'one_stream_us': round(t1, 1), 'equal_to_unsplit': bool(torch.equal(y1, ref))}
  reset()
  res['one_stream_eager_launches_us'] = round(eager(serial_split, NE)[0], 1)
  for stagger in (False, True):
    reset()
    t2, y2 = eager(lambda k: piped(k, stagger), NE)
    key = 'streams_%s' % ('staggered' if stagger else 'free')
    res[key + '_eager_launches_us'] = round(t2, 1)
    res[key + '_equal'] = bool(torch.equal(y2, ref))
  print(json.dumps(res), flush=True)
  if os.environ.get('CAPTURE_STREAMS', '1') != '0':
    for stagger in (False, True):
      reset()
      t2, y2 = capture(lambda k: piped(k, stagger), NE)
      print(json.dumps({'parts': P, 'captured': 'staggered' if stagger else 'free', 'us': round(t2, 1), 'equal': bool(torch.equal(y2, ref))}), flush=True)
