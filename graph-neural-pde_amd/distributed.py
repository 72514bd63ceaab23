"""Row-partitioned multi-GPU solve with a halo exchange per right-hand-side evaluation
(SURVEY.md section 8e).  One process per GPU, `torch.distributed` ("nccl" == RCCL over xGMI).

The reference is single-device (only `nn.DataParallel` wrappers in its HPO scripts,
ray_tune.py:65-66, which are wrong for a full-graph model).  Here the nodes are split into P parts by
the native k-way partitioner; rank p owns the rows V_p of the operator and of the state.  Its local
graph numbers owned columns first and halo columns after them, grouped by owning peer in ascending
global order, so ONE `all_to_all_single` (a grouped send/recv per peer: every xGMI link busy, no ring)
drops each peer's boundary rows straight into the halo region [n_own, n_own + n_halo) of the stage
buffer -- no unpack kernel.  The exchange runs once per evaluation (4x per rk4 step: each stage
evaluates f at a different state).  Keys of halo rows are recomputed locally (k = W_k x is cheaper than
a second exchange); softmax over rows needs no further collective.

The arithmetic is delegated to a backend object.  The product backend (`NativeBackend`) calls the HIP
library and refuses to run without a GPU; tests on CPU (gloo, world size 2) inject a checker backend
built on the oracle to validate partitioning, index maps and the exchange.
"""
import json
import os
import time

import torch
import torch.distributed as dist

from . import _lib
from .graph import CSRGraph, partition_rows
from .odeint import time_grid


# --------------------------------------------------------------------------------------------------
# partition plan (host, deterministic: every rank computes the same plan from the same inputs)
# --------------------------------------------------------------------------------------------------
def pair_traffic(edge_index, part, world):
  """M[r, q] = rows of part q that the rows of part r reference (distinct nodes): what rank r receives from rank q in every
  evaluation, i.e. the load of the xGMI link q -> r."""
  row, col = edge_index
  pr, pc = part[row], part[col]
  cut = pr != pc
  key = torch.unique(pr[cut] * part.numel() + col[cut])          # distinct (receiving part, node)
  recv, node = key // part.numel(), key % part.numel()
  m = torch.zeros(world * world, dtype=torch.long)
  m.index_add_(0, recv * world + part[node], torch.ones_like(recv))
  return m.view(world, world)


# relative cost of one received row (4 d bytes over one xGMI link) and one unit of aggregation work (an entry gathered at the
# GPU's aggregate bandwidth): both scale with d, their ratio does not
LINK_ROW_COST = 64


class PartitionPlan(object):
  """Rows -> ranks.  The native partitioner is a heuristic (label propagation + packing + refinement) whose outcome moves by
  tens of per cent with its tie-breaking; what a partitioned evaluation pays for is (i) the busiest xGMI link -- the exchange
  ends when the LAST link has delivered, and every pair of ranks has its own link -- and (ii) the busiest rank's aggregation.
  So a few candidates (row weight of the balance constraint x size cap of the packed clusters) are scored by
  LINK_ROW_COST * max link rows + max part work  and the best is taken; every rank computes the same candidates from the
  same inputs and reaches the same choice."""

  # (row_weight, seed, cluster_div); the first is the partitioner's default
  CANDIDATES = ((1, 0, 4), (4, 0, 4), (1, 0, 8), (4, 0, 8), (1, 0, 16), (4, 0, 32), (1, 0, 2), (16, 0, 4))
  MAX_EDGES_FOR_SEARCH = 20_000_000                                      # beyond: one partition (they cost minutes there)

  def __init__(self, edge_index, n, world, part=None, refine_iters=8):
    ei = edge_index.detach().cpu().long()
    self.n, self.world = int(n), int(world)
    self.candidates = None
    if part is None:
      g = CSRGraph(ei, n, device='cpu')
      cands = self.CANDIDATES if (world >= 2 and ei.shape[1] <= self.MAX_EDGES_FOR_SEARCH) else self.CANDIDATES[:1]
      deg = torch.bincount(ei[0], minlength=self.n)
      best, self.candidates = None, []
      for row_weight, seed, cluster_div in cands:
        cand = partition_rows(g, world, refine_iters=refine_iters, seed=seed, row_weight=row_weight,
                              cluster_div=cluster_div).long()
        if len(cands) == 1:
          best = (0, cand)
          break
        links = pair_traffic(ei, cand, world)
        work = torch.zeros(world, dtype=torch.long).index_add_(0, cand, deg + 3)
        cost = LINK_ROW_COST * int(links.max()) + int(work.max())
        self.candidates.append({'row_weight': row_weight, 'seed': seed, 'cluster_div': cluster_div, 'max_link_rows': int(links.max()),
                                'max_halo_rows': int(links.sum(dim=1).max()), 'max_part_work': int(work.max()), 'cost': cost})
        if best is None or cost < best[0]:
          best = (cost, cand)
      part = best[1]
    self.part = part.long()
    key = self.part * self.n + torch.arange(self.n)
    self.order = torch.argsort(key)                      # new position -> old node id
    self.newid = torch.empty(self.n, dtype=torch.long)
    self.newid[self.order] = torch.arange(self.n)        # old node id -> new position
    self.counts = torch.bincount(self.part, minlength=world)
    self.offsets = torch.zeros(world + 1, dtype=torch.long)
    self.offsets[1:] = torch.cumsum(self.counts, 0)
    self.edge_index = ei

  # the settings a search draws from: (row_weight, cluster_div, seed), a fixed order
  SEARCH_SPACE = tuple((w, dv, sd) for sd in (0, 1, 2) for w in (1, 2, 4, 8) for dv in (2, 4, 8, 16, 32, 64))

  @staticmethod
  def score(ei, part, world, deg=None):
    """(cost, busiest link rows, largest halo, busiest rank's work) of a partition: what a partitioned evaluation waits for."""
    if deg is None:
      deg = torch.bincount(ei[0], minlength=part.numel())
    links = pair_traffic(ei, part, world)
    work = torch.zeros(world, dtype=torch.long).index_add_(0, part, deg + 3)
    return LINK_ROW_COST * int(links.max()) + int(work.max()), int(links.max()), int(links.sum(dim=1).max()), int(work.max())

  @classmethod
  def search(cls, edge_index, n, world, rank=0, group_size=1, per_rank=9, refine_iters=8, group=None):
    """The ranks of a job share the search: rank r scores the settings r, r + group_size, ... of SEARCH_SPACE (per_rank of
    them, ~0.5 s each at the ogbn-arxiv shape), the scores are all-gathered (host-side objects, any backend), the cheapest
    setting wins (ties: the earlier one) and every rank recomputes THAT partition itself -- the partitioner is deterministic,
    so nothing but a few numbers travels.  group_size = 1: a plain local search over the first per_rank settings."""
    ei = edge_index.detach().cpu().long()
    world, group_size = int(world), max(int(group_size), 1)
    if world == 1 or ei.shape[1] > cls.MAX_EDGES_FOR_SEARCH:
      return cls(ei, n, world, refine_iters=refine_iters)
    g = CSRGraph(ei, n, device='cpu')
    deg = torch.bincount(ei[0], minlength=int(n))
    mine = []
    for idx in range(int(rank), len(cls.SEARCH_SPACE), group_size)[:max(int(per_rank), 1)]:
      w, dv, sd = cls.SEARCH_SPACE[idx]
      cand = partition_rows(g, world, refine_iters=refine_iters, seed=sd, row_weight=w, cluster_div=dv).long()
      cost, link, halo, work = cls.score(ei, cand, world, deg)
      mine.append({'index': idx, 'row_weight': w, 'cluster_div': dv, 'seed': sd, 'max_link_rows': link, 'max_halo_rows': halo,
                   'max_part_work': work, 'cost': cost})
    if group_size > 1:
      gathered = [None] * group_size
      dist.all_gather_object(gathered, mine, group=group)
      scored = [c for part_list in gathered for c in part_list]
    else:
      scored = mine
    scored.sort(key=lambda c: (c['cost'], c['index']))
    best = scored[0]
    part = partition_rows(g, world, refine_iters=refine_iters, seed=best['seed'], row_weight=best['row_weight'],
                          cluster_div=best['cluster_div'])
    plan = cls(ei, n, world, part=part)
    plan.candidates = scored
    return plan

  def edge_cut(self):
    r, c = self.edge_index
    return float((self.part[r] != self.part[c]).float().mean())

  def shard(self, rank):
    return LocalShard(self, rank)

  def shard_ids(self, rank):
    """own_old_ids of LocalShard(self, rank) -- the original node ids of part `rank` in that shard's local row order
    (interior rows first) -- memoised: every rank needs every part's order to put gathered rows back."""
    memo = self.__dict__.setdefault('_shard_ids', {})
    if rank not in memo:
      memo[rank] = LocalShard(self, rank).own_old_ids
    return memo[rank]


class LocalShard(object):
  """Everything rank `rank` needs: local edge list (rows local, columns local incl. halo), ids of its
  edges in the global edge list, send list and the per-peer row counts of the exchange."""

  def __init__(self, plan, rank):
    n, P = plan.n, plan.world
    row, col = plan.edge_index
    prow, pcol = plan.part[row], plan.part[col]
    off = plan.offsets
    mine = prow == rank
    self.rank, self.world = rank, P
    self.n_own = int(plan.counts[rank])
    self.edge_ids = torch.nonzero(mine).flatten()
    r_part = plan.newid[row[mine]] - off[rank]             # row position inside the part (ascending old id)
    c_new = plan.newid[col[mine]]
    c_own = pcol[mine] == rank
    # interior rows (every neighbour owned) first, boundary rows after them: the interior pass of an evaluation
    # can then run while the halo exchange is in flight
    touches_halo = torch.zeros(self.n_own, dtype=torch.bool)
    touches_halo[r_part[~c_own]] = True
    order_local = torch.argsort(touches_halo.long() * self.n_own + torch.arange(self.n_own))
    self.n_interior = int((~touches_halo).sum())
    pos_local = torch.empty(self.n_own, dtype=torch.long)
    pos_local[order_local] = torch.arange(self.n_own)     # position inside the part -> local row id
    self.halo_new = torch.unique(c_new[~c_own])          # sorted => grouped by owner, ascending inside
    self.n_halo = int(self.halo_new.numel())
    owner = torch.bucketize(self.halo_new, off[1:], right=True)
    self.recv_counts = torch.bincount(owner, minlength=P).tolist()
    c_local = torch.where(c_own, pos_local[(c_new - off[rank]).clamp(0, max(self.n_own - 1, 0))],
                          self.n_own + torch.searchsorted(self.halo_new, c_new))
    self.edge_index = torch.stack([pos_local[r_part], c_local])
    # rows of mine that peers reference: unique (peer, column) pairs over the peers' rows, same order
    need = (pcol == rank) & (prow != rank)
    key = torch.unique(prow[need] * n + plan.newid[col[need]])
    self.send_idx = pos_local[(key % n) - off[rank]]
    self.send_counts = torch.bincount(key // n, minlength=P).tolist()
    self.own_old_ids = plan.order[off[rank]:off[rank + 1]][order_local]

  @property
  def n_local(self):
    return self.n_own + self.n_halo


# --------------------------------------------------------------------------------------------------
# product backend: HIP kernels through the C ABI
# --------------------------------------------------------------------------------------------------
def boundary_cuts(rows, n_interior, n_own, n_chunks):
  """Row boundaries [b_0 = n_interior, ..., b_k = n_own] that cut the boundary rows into k <= n_chunks consecutive, non-empty
  ranges of about equal entry counts (`rows`: the row index of every local entry).  Host arithmetic only."""
  b0, b1 = int(n_interior), int(n_own)
  k = max(1, min(int(n_chunks), b1 - b0))
  bounds = [b0]
  if b1 > b0:
    deg = torch.bincount(rows[rows >= b0] - b0, minlength=b1 - b0).to(torch.float64)
    cum = torch.cumsum(deg + 1e-3, 0)      # (+eps: rows without entries still advance, so no range is empty)
    for c in range(1, k):
      cut = b0 + int(torch.searchsorted(cum, cum[-1] * c / k).item()) + 1
      bounds.append(min(max(cut, bounds[-1] + 1), b1 - (k - c)))
  bounds.append(b1)
  return bounds


class NativeBackend(object):
  """f(u) with fused solver stage on the local shard.  kind = 'laplacian' (params: edge_weight over
  the LOCAL edges), 'transformer' (params: Wq, bq, Wk, bk, heads, att_type, norm_idx, square_plus) or 'gat' (params: W [A, d], a,
  heads, leaky_slope, norm_idx)."""

  def __init__(self, shard, d, device, kind, params, alpha, beta, alpha_sigmoid=True):
    from . import ops
    if torch.device(device).type != 'cuda':
      raise _lib.GnpdeError('the sharded solve runs only on HIP devices; there is no CPU fallback')
    self.ops, self.dev, self.kind, self.d = ops, torch.device(device), kind, d
    self.shard = shard
    self.graph = CSRGraph(shard.edge_index, shard.n_local, device=self.dev)
    self.graph.set_row_range(0, shard.n_own)     # rows to process; columns may address halo rows
    # interior / boundary views for the overlapped evaluation: the same local numbering, each holding only its
    # rows' edges; the boundary view aggregates rows [n_interior, n_own)
    inter = shard.edge_index[0] < shard.n_interior
    self.eid_int = torch.nonzero(inter).flatten()
    self.eid_bnd = torch.nonzero(~inter).flatten()
    self.g_int = CSRGraph(shard.edge_index[:, inter], shard.n_local, device=self.dev)
    self.g_int.set_row_range(0, shard.n_interior)
    self.g_bnd = CSRGraph(shard.edge_index[:, ~inter], shard.n_local, device=self.dev)
    self.g_bnd.set_row_range(shard.n_interior, shard.n_own)
    self.supports_split = True
    self.alpha = alpha.detach().to(self.dev, torch.float32).reshape(-1)
    self.beta = beta.detach().to(self.dev, torch.float32).reshape(-1)
    self.alpha_sigmoid = alpha_sigmoid
    self.send_idx = shard.send_idx.to(torch.int32).to(self.dev)
    self.x0 = torch.zeros(shard.n_own, d, dtype=torch.float32, device=self.dev)   # persistent source term
    self.has_source = False
    if kind == 'laplacian':
      ew = params['edge_weight'].to(self.dev, torch.float32)
      self._edge_weight = ew
      self.w_csr = ops.edge_to_csr_mean(self.graph, ew)
      self.w_int = ops.edge_to_csr_mean(self.g_int, ew[self.eid_int.to(self.dev)])
      self.w_bnd = ops.edge_to_csr_mean(self.g_bnd, ew[self.eid_bnd.to(self.dev)])
      self._kw = dict(kind=_lib.RHS_LAPLACIAN)
    elif kind == 'transformer':
      self.wqk = torch.cat([params['Wq'], params['Wk']]).to(self.dev, torch.float32).contiguous()
      self.bqk = torch.cat([params['bq'], params['bk']]).to(self.dev, torch.float32).contiguous()
      self.heads = int(params['heads'])
      self.A = self.wqk.shape[0] // 2
      self.norm_idx = int(params.get('norm_idx', 0))
      self.square_plus = bool(params.get('square_plus', False))
      self.general = self.norm_idx != 0 or self.square_plus      # needs exchanges BETWEEN the attention passes (rhs_stage_general)
      # score function (reference src/function_transformer_attention.py:193-206): scaled_dot, cosine_sim, pearson or exp_kernel --
      # all are functions of (q_i, k_j) alone, so the partitioned evaluation is the same (keys of the halo rows recomputed locally)
      self.att_type = str(params.get('att_type', 'scaled_dot'))
      self.att_code = _lib.ATT_TYPES[self.att_type]
      self.output_var = self.lengthscale = None
      if self.att_type == 'exp_kernel':
        self.output_var = params['output_var'].detach().to(self.dev, torch.float32).reshape(-1).clone()
        self.lengthscale = params['lengthscale'].detach().to(self.dev, torch.float32).reshape(-1).clone()
      att = ops.attention_struct(self.att_code, self.heads, self.A, 0, False, output_var=self.output_var, lengthscale=self.lengthscale)
      self._kw = dict(kind=_lib.RHS_TRANSFORMER, proj_w=self.wqk, proj_b=self.bqk, att=att)
      # opt['reweight_attention'] (reference src/function_transformer_attention.py:208-209): the scores of an entry times the weight of
      # its edge -- row-local data, every view of the local graph gets the weights of ITS entries in ITS CSR order
      self.reweight = None
      if params.get('edge_weight') is not None:
        ew = params['edge_weight'].detach().to(self.dev, torch.float32)
        self._edge_weight = ew
        self.reweight = {None: ops.edge_to_csr_mean(self.graph, ew),
                         'interior': ops.edge_to_csr_mean(self.g_int, ew[self.eid_int.to(self.dev)]),
                         'boundary': ops.edge_to_csr_mean(self.g_bnd, ew[self.eid_bnd.to(self.dev)])}
      if self.general:
        # the attention passes see ALL local nodes as segments (a halo column is a segment of attention_norm_idx = 1)
        self.g_att = CSRGraph(shard.edge_index, shard.n_local, device=self.dev)
        self.supports_split = False
        if self.reweight is not None:
          self.reweight['att'] = ops.edge_to_csr_mean(self.g_att, self._edge_weight)
    elif kind == 'gat':
      # GAT scores (reference src/function_GAT_attention.py:105-115): leaky_relu(a_src . Wx_i + a_dst . Wx_j) -- functions of the
      # projected rows of the two end points, so the halo rows' projections are recomputed locally like the transformer's keys
      self.wqk = params['W'].detach().to(self.dev, torch.float32).contiguous().clone()     # [A, d] (SpGraphAttentionLayer.proj_weight)
      self.bqk = None
      self.heads = int(params['heads'])
      self.A = self.wqk.shape[0]
      self.norm_idx = int(params.get('norm_idx', 0))
      self.square_plus = False
      self.general = self.norm_idx != 0
      self.att_type, self.att_code = 'GAT', _lib.ATT_GAT
      self.output_var = self.lengthscale = None
      self.gat_a = params['a'].detach().to(self.dev, torch.float32).reshape(-1).clone()
      self.leaky_slope = float(params.get('leaky_slope', 0.2))
      att = ops.attention_struct(_lib.ATT_GAT, self.heads, self.A, 0, False, leaky_slope=self.leaky_slope, gat_a=self.gat_a)
      self._kw = dict(kind=_lib.RHS_GAT, proj_w=self.wqk, proj_b=None, att=att)
      if self.general:
        self.g_att = CSRGraph(shard.edge_index, shard.n_local, device=self.dev)
        self.supports_split = False
    else:
      raise ValueError(kind)
    self._desc = {}
    self._ws = None
    self._src, self._src_version = None, -1
    self.chunk_sets = {}        # boundary rows in k row ranges (split_boundary): k -> [(row_begin, row_end, graph, eid, w_csr)]

  def split_boundary(self, n_chunks):
    """The boundary rows [n_interior, n_own) as up to n_chunks consecutive row ranges of about equal edge counts, each with
    its own view of the local graph (same numbering, only its rows' edges): the passes of the chunked boundary evaluation
    (NativeShardedSolver(boundary_chunks=...)).  Returns [(row_begin, row_end, graph, edge ids, w_csr)], kept per n_chunks."""
    if int(n_chunks) in self.chunk_sets:
      return self.chunk_sets[int(n_chunks)]
    sh = self.shard
    rows = sh.edge_index[0]
    bounds = boundary_cuts(rows, int(sh.n_interior), int(sh.n_own), n_chunks)
    chunks = []
    for c in range(len(bounds) - 1):
      lo, hi = bounds[c], bounds[c + 1]
      m = (rows >= lo) & (rows < hi)
      g = CSRGraph(sh.edge_index[:, m], sh.n_local, device=self.dev)
      g.set_row_range(lo, hi)
      eid = torch.nonzero(m).flatten()
      w = None
      if self.kind == 'laplacian' or getattr(self, 'reweight', None) is not None:
        w = self.ops.edge_to_csr_mean(g, self._edge_weight[eid.to(self.dev)])
      chunks.append((lo, hi, g, eid, w))
    self.chunk_sets[int(n_chunks)] = chunks
    return chunks

  def refresh(self, alpha, beta, params):
    """New values of the scalars / weights IN PLACE (captured graphs and descriptors keep their pointers): what changes
    between two forwards of a block -- trained parameters, attention weights recomputed per forward."""
    self.alpha.copy_(alpha.detach().reshape(-1))
    self.beta.copy_(beta.detach().reshape(-1))
    if self.kind == 'laplacian':
      ew = params['edge_weight'].to(self.dev, torch.float32)
      self.ops.edge_to_csr_mean(self.graph, ew, out=self.w_csr)
      self.ops.edge_to_csr_mean(self.g_int, ew[self.eid_int.to(self.dev)], out=self.w_int)
      self.ops.edge_to_csr_mean(self.g_bnd, ew[self.eid_bnd.to(self.dev)], out=self.w_bnd)
      self._edge_weight = ew
      for chunks in self.chunk_sets.values():
        for (_, _, g, eid, w) in chunks:
          self.ops.edge_to_csr_mean(g, ew[eid.to(self.dev)], out=w)
    elif self.kind == 'gat':
      self.wqk.copy_(params['W'].detach().to(self.dev, torch.float32))
      self.gat_a.copy_(params['a'].detach().reshape(-1))
    else:
      self.wqk.copy_(torch.cat([params['Wq'], params['Wk']]).to(self.dev, torch.float32))
      self.bqk.copy_(torch.cat([params['bq'], params['bk']]).to(self.dev, torch.float32))
      if self.output_var is not None:
        self.output_var.copy_(params['output_var'].detach().reshape(-1))
        self.lengthscale.copy_(params['lengthscale'].detach().reshape(-1))
      if getattr(self, 'reweight', None) is not None and params.get('edge_weight') is not None:
        # opt['reweight_attention']: the edge weights may have changed while the edge_index tensor (the cache key) stayed the same --
        # every view's CSR-ordered copy is rebuilt in place, like the laplacian branch above (round-4 advisor item)
        ew = params['edge_weight'].detach().to(self.dev, torch.float32)
        self._edge_weight = ew
        self.ops.edge_to_csr_mean(self.graph, ew, out=self.reweight[None])
        self.ops.edge_to_csr_mean(self.g_int, ew[self.eid_int.to(self.dev)], out=self.reweight['interior'])
        self.ops.edge_to_csr_mean(self.g_bnd, ew[self.eid_bnd.to(self.dev)], out=self.reweight['boundary'])
        if 'att' in self.reweight:
          self.ops.edge_to_csr_mean(self.g_att, ew, out=self.reweight['att'])
        for chunks in self.chunk_sets.values():
          for (_, _, g, eid, w) in chunks:
            if w is not None:
              self.ops.edge_to_csr_mean(g, ew[eid.to(self.dev)], out=w)

  def _descriptor(self, with_source, part=None):
    """gnpde_rhs_t over the local shard: aggregation on the n_own rows (or the interior / boundary rows),
    projection over all n_local rows (or the own / halo rows).  All variants share one workspace, whose first
    region is the q||k projection (csrc/solver.hip rhs_layout)."""
    d = self._desc.get((with_source, part))
    if d is None:
      kw = dict(self._kw)
      kind = kw.pop('kind')
      sh = self.shard
      # ('chunk', k, c): the c-th of the k row ranges of the boundary rows (split_boundary)
      chunk, chunks = (part[2], self.chunk_sets[part[1]]) if isinstance(part, tuple) else (None, None)
      graph = chunks[chunk][2] if chunk is not None else {None: self.graph, 'interior': self.g_int, 'boundary': self.g_bnd}[part]
      if kind != _lib.RHS_LAPLACIAN and getattr(self, 'reweight', None) is not None:
        rw = chunks[chunk][4] if chunk is not None else self.reweight[part]
        kw['att'] = self.ops.attention_struct(self.att_code, self.heads, self.A, 0, False, output_var=self.output_var,
                                              lengthscale=self.lengthscale, edge_w_csr=rw)
      if kind == _lib.RHS_LAPLACIAN:
        kw['w_csr'] = chunks[chunk][4] if chunk is not None else {None: self.w_csr, 'interior': self.w_int, 'boundary': self.w_bnd}[part]
      elif chunk is not None:
        # the first chunk projects the halo rows that just arrived, the later ones nothing (an empty slice)
        kw['proj_rows'] = (sh.n_own, sh.n_local) if chunk == 0 else (sh.n_local, sh.n_local)
      else:
        # interior pass: keys / queries of the own rows; boundary pass: keys of the halo rows that just arrived
        # (an empty slice when there is no halo)
        kw['proj_rows'] = {None: None, 'interior': (0, sh.n_own), 'boundary': (sh.n_own, sh.n_local)}[part]
      d = self.ops.RhsDescriptor(kind, graph, self.d, self.d, self.alpha, self.beta if with_source else None,
                                 self.x0 if with_source else None, self.alpha_sigmoid, n_state_rows=sh.n_local, **kw)
      self._desc[(with_source, part)] = d
      need = _lib.lib().gnpde_rhs_workspace_bytes(d.ref())
      if self._ws is None or self._ws.numel() < need:
        old_ws = self._ws
        self._ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=self.dev)
        del old_ws
    return d

  def empty(self, rows):
    return torch.empty(rows, self.d, dtype=torch.float32, device=self.dev)

  def pack(self, u, out):
    """out[i] = u[send_idx[i]] (boundary rows into the send buffer)."""
    if out.shape[0] == 0:
      return out
    _lib.check(_lib.lib().gnpde_gather_rows(_lib.ptr(u), u.stride(0), _lib.ptr(self.send_idx), out.shape[0], self.d,
                                            _lib.ptr(out), out.stride(0), _lib.stream_of(u)))
    return out

  def rhs_stage(self, u, x0, stage, part=None, **stage_kw):
    """One evaluation with a fused solver stage: ONE call into the library (projection over own + halo rows,
    attention and aggregation over the own rows), so the host side stays cheap next to the per-GPU work.
    part='interior' / 'boundary' runs the half of the evaluation that does not / does need the halo rows."""
    # new or modified source tensor: refresh the persistent copy.  Keyed on the tensor OBJECT (kept alive here, so its
    # address cannot be handed to another tensor) and its version counter (in-place updates), not on data_ptr():
    # the caching allocator gives the next forward's x0 the address of the freed previous one.
    if x0 is not None and (x0 is not self._src or x0._version != self._src_version):
      self.x0.copy_(x0)
      self._src, self._src_version = x0, x0._version
    desc = self._descriptor(x0 is not None, part)
    self.ops.rhs_stage(desc, u, stage, ws=self._ws, **stage_kw)

  # ---- attention_norm_idx = 1 and / or squareplus: exchanges between the attention passes --------------------------
  def rhs_stage_general(self, u, x0, stage, group=None, **stage_kw):
    """One evaluation (halo rows of u already refreshed) for the normalisers that are not row-local:
       pass 1  scores of the local entries (+ the local maximum)          -> squareplus: MAX all-reduce of one word
       pass 2  statistics per local segment (rows, or own + halo COLUMNS)  -> norm_idx 1: partial statistics of the halo columns
               go to their owners (the reverse of the halo exchange), are merged there peer by peer in rank order
               (gnpde_segment_stats_merge) and come back with the state's exchange pattern
       pass 3  normalise + head mean, then the aggregation with the fused stage on the owned rows.
    Reference: src/function_transformer_attention.py:190-213 (softmax / squareplus over edge[attention_norm_idx]),
    src/utils.py:179-208 (squareplus' global maximum)."""
    import ctypes
    ops, L, sh = self.ops, _lib.lib(), self.shard
    if x0 is not None and (x0 is not self._src or x0._version != self._src_version):
      self.x0.copy_(x0)
      self._src, self._src_version = x0, x0._version
    h, A, n_own, n_loc = self.heads, self.A, sh.n_own, sh.n_local
    qk = ops.linear(u, self.wqk, self.bqk)                               # own AND halo rows (keys of halo rows recomputed locally)
    if self.kind == 'gat':
      st = ops.attention_struct(_lib.ATT_GAT, h, A, self.norm_idx, False, q=qk, k=qk, ldqk=A, leaky_slope=self.leaky_slope,
                                gat_a=self.gat_a)
    else:
      st = ops.attention_struct(self.att_code, h, A, self.norm_idx, self.square_plus, q=qk, k=qk[:, A:], ldqk=2 * A,
                                output_var=self.output_var, lengthscale=self.lengthscale,
                                edge_w_csr=self.reweight['att'] if self.reweight is not None else None)
    g = self.g_att
    ws = g.workspace('att', L.gnpde_attention_workspace_bytes(g.ref(), ctypes.byref(st)))
    offs = (ctypes.c_size_t * 4)()
    _lib.check(L.gnpde_attention_workspace_regions(g.ref(), ctypes.byref(st), offs))
    stream = _lib.stream_of(u)

    def run_pass(k, w=None):
      _lib.check(L.gnpde_edge_attention_pass(g.ref(), ctypes.byref(st), k, _lib.ptr(w), _lib.ptr(ws), ws.numel(), stream))
    run_pass(1)
    world = sh.world
    if self.square_plus and world > 1:
      # the maximum lives as an order-preserving uint32: MAX over the ranks on its unsigned value
      word = ws[offs[3]:offs[3] + 4].view(torch.int32)
      val = word.to(torch.int64) & 0xffffffff
      if dist.get_backend(group) != 'nccl':
        hv = val.cpu()
        dist.all_reduce(hv, op=dist.ReduceOp.MAX, group=group)
        val = hv.to(self.dev)
      else:
        dist.all_reduce(val, op=dist.ReduceOp.MAX, group=group)
      word.copy_(torch.where(val >= 2 ** 31, val - 2 ** 32, val).to(torch.int32))
    run_pass(2)
    if self.norm_idx == 1 and world > 1:
      m = ws[offs[1]:offs[1] + 4 * n_loc * h].view(torch.float32).view(n_loc, h)
      den = ws[offs[2]:offs[2] + 4 * n_loc * h].view(torch.float32).view(n_loc, h)
      n_send = int(sum(sh.send_counts))
      # (i) partial statistics of the halo columns -> their owners
      out_part = torch.cat([m[n_own:], den[n_own:]], dim=1).contiguous()
      in_part = torch.empty(n_send, 2 * h, dtype=torch.float32, device=self.dev)
      _all_to_all_rows(in_part, out_part, sh.send_counts, sh.recv_counts, group)
      # (ii) merge at the owner, peer by peer in rank order (the peers' lists of my rows are unique: one call per peer)
      pos = 0
      for p in range(world):
        cnt = int(sh.send_counts[p])
        if cnt:
          seg = in_part[pos:pos + cnt]
          m_in, den_in, rows = seg[:, :h].contiguous(), seg[:, h:].contiguous(), self.send_idx[pos:pos + cnt]
          # (named: a temporary handed to ptr() is freed at once and the next temporary takes its memory)
          _lib.check(L.gnpde_segment_stats_merge(_lib.ptr(m), _lib.ptr(den), _lib.ptr(rows), cnt, h, _lib.ptr(m_in), _lib.ptr(den_in),
                                                 int(self.square_plus), stream))
        pos += cnt
      # (iii) the totals travel back like the state's boundary rows
      tot = torch.cat([m[:n_own], den[:n_own]], dim=1)[self.send_idx.long()].contiguous()
      back = torch.empty(sh.n_halo, 2 * h, dtype=torch.float32, device=self.dev)
      _all_to_all_rows(back, tot, sh.recv_counts, sh.send_counts, group)
      m[n_own:].copy_(back[:, :h])
      den[n_own:].copy_(back[:, h:])
    if getattr(self, '_w_general', None) is None:
      self._w_general = torch.empty(max(g.e, 1), dtype=torch.float32, device=self.dev)
    run_pass(3, self._w_general)
    beta = self.beta if x0 is not None else None
    ops.spmm_rhs(self.graph, self._w_general, u, self.alpha, beta, self.x0 if x0 is not None else None, self.alpha_sigmoid,
                 stage=stage, **stage_kw)

  def sync(self):
    torch.cuda.synchronize(self.dev)


# --------------------------------------------------------------------------------------------------
# sharded fixed-step solver
# --------------------------------------------------------------------------------------------------
class ShardedSolver(object):
  def __init__(self, shard, backend, group=None):
    self.shard, self.be, self.group = shard, backend, group
    s = shard
    self.y = backend.empty(s.n_local)
    self.ua = backend.empty(s.n_local)
    self.ub = backend.empty(s.n_local)
    self.uc = backend.empty(s.n_local)
    self.send = backend.empty(int(sum(s.send_counts)))
    self.n_exchanges = 0
    self.overlap = os.environ.get('GNPDE_NO_OVERLAP', '0') != '1'   # interior rows overlap the halo exchange

  def exchange(self, u, async_op=False):
    """Refresh the halo rows of `u` (rows [n_own, n_local)) from their owners."""
    s = self.shard
    if s.world == 1 and os.environ.get('GNPDE_FORCE_SHARDED', '0') != '1':
      return None
    self.be.pack(u, self.send)
    recv = u[s.n_own:]
    self.n_exchanges += 1
    if not async_op and dist.get_backend(self.group) != 'nccl' and recv.is_cuda:
      _all_to_all_rows(recv, self.send, s.recv_counts, s.send_counts, self.group)
      return None
    return dist.all_to_all_single(recv, self.send, s.recv_counts, s.send_counts, group=self.group, async_op=async_op)

  def evaluate(self, u, x0, stage, **kw):
    """Exchange + f(u) with the fused stage.  When the backend can split the rows, the interior rows (no halo
    neighbour) are evaluated while the all-to-all is in flight and the boundary rows after it has landed."""
    if getattr(self.be, 'general', False):
      self.exchange(u)
      self.be.rhs_stage_general(u, x0, stage, group=self.group, **kw)
    elif getattr(self.be, 'supports_split', False) and self.overlap and (not u.is_cuda or dist.get_backend(self.group) == 'nccl'):
      # (an asynchronous all-to-all of device tensors over gloo completes on gloo's own stream: wait() does not order it before
      #  the boundary pass on the launch stream -- such groups take the blocking, host-staged exchange below)
      work = self.exchange(u, async_op=True)
      self.be.rhs_stage(u, x0, stage, part='interior', **kw)
      if work is not None:
        work.wait()
      self.be.rhs_stage(u, x0, stage, part='boundary', **kw)
    else:
      self.exchange(u)
      self.be.rhs_stage(u, x0, stage, **kw)

  def integrate(self, y_own, x0_own, T, step_size=1.0, method='rk4'):
    """Integrate the owned rows from t = 0 to T; returns the owned rows of y(T) (a view of an
    internal buffer).  `x0_own` may be None (no source term)."""
    s, be = self.shard, self.be
    grid = time_grid(torch.tensor([0.0, float(T)]), step_size)
    dts = (grid[1:] - grid[:-1]).tolist()
    y, ua, ub, uc = self.y, self.ua, self.ub, self.uc
    n = s.n_own
    y[:n].copy_(y_own)
    for dt in dts:
      if method == 'euler':
        self.evaluate(y, x0_own, _lib.STAGE_EULER, dt=dt, y=y, out_y=ua)
        y, ua = ua, y
      elif method == 'rk4':   # compact stage algebra (gnpde.h): stage states from stage inputs, no k arrays
        self.evaluate(y, x0_own, _lib.STAGE_RK1C, dt=dt, out_y=ua)
        self.evaluate(ua, x0_own, _lib.STAGE_RK2C, dt=dt, y=y, out_y=ub)
        self.evaluate(ub, x0_own, _lib.STAGE_RK3C, dt=dt, k1=ua, out_y=uc)
        self.evaluate(uc, x0_own, _lib.STAGE_RK4C, dt=dt, y=y, k1=ub, out_y=y)
      else:
        raise ValueError(method)
    self.y, self.ua = y, ua
    return y[:n]

  def all_rows_rms(self, n_rows_total):
    """rms over ALL rows of the partitioned state (torchdiffeq's default norm, misc.py _rms_norm), for tensors that hold this
    rank's rows: the squares are summed in double per rank and added over the ranks -- every rank gets the same bits back from
    the all-reduce, so every rank takes the same accept / reject decisions and step sizes."""
    total = float(n_rows_total) * float(self.be.d)
    group = self.group

    def norm(v):
      sq = v.double().pow(2).sum().reshape(1)
      if self.shard.world > 1 or os.environ.get('GNPDE_FORCE_SHARDED', '0') == '1':
        if sq.is_cuda and dist.get_backend(group) != 'nccl':
          host = sq.cpu()
          dist.all_reduce(host, group=group)
          sq = host.to(v.device)
        else:
          dist.all_reduce(sq, group=group)
      return (sq / total).sqrt().to(v.dtype)[0]
    return norm

  def integrate_adaptive(self, y_own, x0_own, t, rtol, atol, n_rows_total, method='dopri5', on_eval=None):
    """torchdiffeq's adaptive embedded pairs (dopri5 / adaptive_heun, reference src/block_constant.py:57-62 with
    opt['method'] = 'dopri5', the default of run_GNN.py) on the partitioned graph: the controller of odeint._solve_dopri5
    (0.2.1's rule, float64 time) on every rank, one halo exchange + one fused evaluation per stage, and the error norm as the
    one extra exchange step the adaptive method has -- an all-reduce of one double per trial step (all_rows_rms).
    Returns the owned rows of y(t[-1])."""
    from .odeint import _solve_dopri5
    n = self.shard.n_own
    u, kbuf = self.ua, self.ub
    self.n_evals = 0

    def f(tq, yo):
      if on_eval is not None:
        on_eval()
      u[:n].copy_(yo)
      self.evaluate(u, x0_own, _lib.STAGE_LINCOMB, out_k=kbuf)
      self.n_evals += 1
      return kbuf[:n].clone()
    out = _solve_dopri5(f, y_own, t, rtol, atol, tableau=method, norm=self.all_rows_rms(n_rows_total))
    return out[-1]


# --------------------------------------------------------------------------------------------------
# native sharded solver: the whole solve enqueued by the library, RCCL inside the hipGraph
# --------------------------------------------------------------------------------------------------
_comm_cache = {}


def _rccl_library_path():
  """The librccl torch.distributed's nccl backend uses: handing the same file to the library makes both share one
  RCCL instance in the process (the copy bundled with torch has no SONAME, so a bare dlopen('librccl.so.1') would
  load a second one from /opt/rocm)."""
  cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
  return cand if os.path.exists(cand) else None


def native_comm(rank, world, group=None):
  """gnpde_comm_t for this process: rank 0 draws the RCCL unique id, torch.distributed ships it (any backend), every
  rank joins.  World 1 needs no process group (self send / recv)."""
  key = (rank, world, id(group))
  if key in _comm_cache:
    return _comm_cache[key]
  import ctypes
  L = _lib.lib()
  path = _rccl_library_path()
  _lib.check(L.gnpde_comm_load_library(path.encode() if path else None))
  buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
  if rank == 0:
    _lib.check(L.gnpde_comm_get_unique_id(buf))
  if world > 1:
    box = [bytes(buf.raw)]
    dist.broadcast_object_list(box, src=0, group=group)
    buf = ctypes.create_string_buffer(box[0], _lib.COMM_ID_BYTES)
  handle = ctypes.c_void_p()
  _lib.check(L.gnpde_comm_create(ctypes.byref(handle), buf, rank, world))
  _comm_cache[key] = handle
  return handle


class P2PContext(object):
  """gnpde_p2p_t of this rank: IPC-shared stage buffers + flag array, connected to every peer.  The 128-byte handles and
  the halo geometry of the peers travel through torch.distributed (any backend: gloo is enough, nothing here needs RCCL)."""

  def __init__(self, shard, d, n_buffers=4, group=None):
    import ctypes
    L = _lib.lib()
    s = shard
    self.shard, self.d = shard, d
    self.buffer_bytes = (max(s.n_local, 1) * d * 4 + 255) // 256 * 256
    handle = ctypes.c_void_p()
    self.handle = None
    buf = ctypes.create_string_buffer(_lib.P2P_HANDLE_BYTES)
    create_failure = None
    try:     # a rank that cannot allocate / export its block must not leave its peers waiting in the gather below
      _lib.check(L.gnpde_p2p_create(ctypes.byref(handle), s.rank, s.world, self.buffer_bytes, n_buffers))
      self.handle = handle
      _lib.check(L.gnpde_p2p_get_handle(handle, buf))
    except _lib.GnpdeError as exc:
      create_failure = str(exc)
    mine = dict(handle=bytes(buf.raw), n_own=s.n_own, recv_counts=[int(v) for v in s.recv_counts],
                buffer_bytes=self.buffer_bytes, failure=create_failure)
    if s.world > 1:
      metas = [None] * s.world
      dist.all_gather_object(metas, mine, group=group)
    else:
      metas = [mine]
    bad = next((m['failure'] for m in metas if m.get('failure')), None)
    if bad is not None:
      self.close()
      raise _lib.GnpdeError('p2p transport unavailable: %s' % bad)
    blob = b''.join(m['handle'] for m in metas)
    failure = None
    try:
      _lib.check(L.gnpde_p2p_connect(handle, ctypes.create_string_buffer(blob, len(blob))))
    except _lib.GnpdeError as exc:     # e.g. the peers' memory cannot be mapped on this system
      failure = str(exc)
    # where THIS rank's rows start inside peer p's stage buffers: after p's own rows and the rows p receives from lower ranks
    self.peer_row0 = [m['n_own'] + sum(m['recv_counts'][:s.rank]) for m in metas]
    self.peer_buffer_bytes = [m['buffer_bytes'] for m in metas]
    self.n_buffers = int(n_buffers)
    self.metas = [dict(n_own=m['n_own'], recv_counts=list(m['recv_counts'])) for m in metas]
    if s.world > 1:
      # every rank has mapped every peer before anyone pushes -- or every rank learns that one of them could not
      outcomes = [None] * s.world
      dist.all_gather_object(outcomes, failure, group=group)
      failure = next((o for o in outcomes if o is not None), None)
    if failure is not None:
      self.close()
      raise _lib.GnpdeError('p2p transport unavailable: %s' % failure)

  def close(self):
    if getattr(self, 'handle', None) is not None and self.handle.value:
      _lib.lib().gnpde_p2p_destroy(self.handle)
      self.handle = None


class NativeShardedSolver(object):
  """gnpde_sharded_solver_t over a NativeBackend's shard: same result as ShardedSolver (the Python-driven loop), but the
  host issues ONE hipGraphLaunch per solve.  transport 'p2p' (default): boundary rows are pushed straight into the peers'
  IPC-mapped halo regions by a kernel inside the graph; 'rccl': grouped ncclSend / ncclRecv, eager launches only."""

  def __init__(self, shard, backend, T, step_size=1.0, method='rk4', with_source=True, transport='p2p', ctx=None,
               comm=None, group=None, boundary_chunks=None):
    """boundary_chunks (P2P only; default GNPDE_BOUNDARY_CHUNKS or 1): k > 1 computes the boundary rows in k row ranges and
    pushes each range's rows of the NEXT stage input to the peers right behind it, so that only the last range's push is
    left to hide behind the next interior pass (same arithmetic in the same order: results are bit-identical to k = 1)."""
    import ctypes
    self.shard, self.be, self.transport = shard, backend, transport
    s = shard
    L = _lib.lib()
    if T is None:       # the exchange engine alone (NativeShardedDopri5 drives it): no time grid
      dts = []
    else:
      grid = time_grid(torch.tensor([0.0, float(T)]), step_size)
      dts = (grid[1:] - grid[:-1]).tolist()
    self.dts = tuple(dts)
    self.d_int = backend._descriptor(with_source, 'interior')
    self.d_bnd = backend._descriptor(with_source, 'boundary')
    self.send_counts = (ctypes.c_int32 * s.world)(*[int(v) for v in s.send_counts])
    self.recv_counts = (ctypes.c_int32 * s.world)(*[int(v) for v in s.recv_counts])
    self.halo = _lib.HaloStruct(world=s.world, rank=s.rank, n_own=s.n_own, n_halo=s.n_halo,
                                send_idx=_lib.ptr(backend.send_idx) if backend.send_idx.numel() else None,
                                send_counts=self.send_counts, recv_counts=self.recv_counts)
    self.method = {'euler': _lib.METHOD_EULER, 'rk4': _lib.METHOD_RK4}[method]
    p2p = transport == 'p2p'
    need = L.gnpde_sharded_solver_workspace_bytes(ctypes.byref(self.halo), self.d_int.ref(), self.d_bnd.ref(), self.method, int(p2p))
    if need == 0:
      raise _lib.GnpdeError('sharded solver: %s' % L.gnpde_last_error().decode(errors='replace'))
    if boundary_chunks is None:
      boundary_chunks = int(os.environ.get('GNPDE_BOUNDARY_CHUNKS', '1'))
    self.boundary_chunks = 1
    self.d_chunks, self.chunks = [], []
    slack = 0
    if p2p and int(boundary_chunks) > 1 and s.world > 1 and s.n_own > s.n_interior:
      self.chunks = backend.split_boundary(int(boundary_chunks))
      self.d_chunks = [backend._descriptor(with_source, ('chunk', int(boundary_chunks), c)) for c in range(len(self.chunks))]
      both = max(L.gnpde_rhs_workspace_bytes(self.d_int.ref()), L.gnpde_rhs_workspace_bytes(self.d_bnd.ref()))
      slack = max(0, max(L.gnpde_rhs_workspace_bytes(dc.ref()) for dc in self.d_chunks) - both) + 256
    self.ws = torch.empty(int(need) + slack, dtype=torch.uint8, device=backend.dev)
    arr = (ctypes.c_float * max(len(dts), 1))(*dts)
    handle = ctypes.c_void_p()
    self.ctx = self.comm = None
    self._own_ctx = False
    if p2p:
      if ctx is None:
        ctx = P2PContext(shard, backend.d, 4, group=group)
        self._own_ctx = True
      self.ctx = ctx
      row0 = (ctypes.c_int64 * s.world)(*ctx.peer_row0)
      pbytes = (ctypes.c_int64 * s.world)(*ctx.peer_buffer_bytes)
      _lib.check(L.gnpde_sharded_solver_create_p2p(ctypes.byref(handle), ctx.handle, ctypes.byref(self.halo), self.d_int.ref(),
                                                   self.d_bnd.ref(), self.method, arr, len(dts), row0, pbytes,
                                                   _lib.ptr(self.ws), self.ws.numel()))
    elif transport == 'rccl':
      exchanges = sum(s.send_counts) + sum(s.recv_counts) > 0
      self.comm = comm if comm is not None else (native_comm(s.rank, s.world, group) if exchanges else None)
      _lib.check(L.gnpde_sharded_solver_create(ctypes.byref(handle), self.comm, ctypes.byref(self.halo), self.d_int.ref(),
                                               self.d_bnd.ref(), self.method, arr, len(dts), _lib.ptr(self.ws), self.ws.numel()))
    else:
      raise ValueError(transport)
    self.handle = handle
    if len(self.d_chunks) > 1:
      order, ptr = self._chunked_push_order(self.chunks)
      refs = (ctypes.POINTER(_lib.RhsStruct) * len(self.d_chunks))(*[ctypes.pointer(dc.struct) for dc in self.d_chunks])
      _lib.check(L.gnpde_sharded_solver_set_boundary_chunks(handle, refs, len(self.d_chunks),
                                                           (ctypes.c_int32 * len(order))(*order), (ctypes.c_int32 * len(ptr))(*ptr)))
      self.boundary_chunks = len(self.d_chunks)
    self.y = backend.empty(s.n_local)
    self.n_rhs_evals = L.gnpde_sharded_solver_num_rhs_evals(handle)
    if getattr(backend, 'general', False):
      try:
        self._attach_general()
      except Exception:
        self.close()
        raise

  @staticmethod
  def stats_layout(n_local, n_send, heads):
    """(byte offset of in_part, bytes of [S | in_part]) inside the shared statistics buffer of a rank (gnpde_general_t)."""
    row = 2 * int(heads) * 4
    off = (int(n_local) * row + 255) // 256 * 256
    return off, off + max(int(n_send), 1) * row

  def _attach_general(self):
    """gnpde_sharded_solver_set_general: the normalisers that are not row-local (attention_norm_idx 1, squareplus) with their exchanges
    between the attention passes inside the per-rank graph -- what NativeBackend.rhs_stage_general drives from Python."""
    import ctypes
    be, s, L, ctx = self.be, self.shard, _lib.lib(), self.ctx
    if self.transport != 'p2p' or ctx is None:
      raise _lib.GnpdeError('normalisers with exchanges between the attention passes: the in-graph form needs the P2P transport')
    h, A = be.heads, be.A
    if be.kind == 'gat':
      att = be.ops.attention_struct(_lib.ATT_GAT, h, A, be.norm_idx, False, leaky_slope=be.leaky_slope, gat_a=be.gat_a)
      proj_m = A
    else:
      att = be.ops.attention_struct(be.att_code, h, A, be.norm_idx, be.square_plus, output_var=be.output_var, lengthscale=be.lengthscale,
                                    edge_w_csr=be.reweight['att'] if be.reweight is not None else None)
      proj_m = 2 * A
    g_att, g_spmm, dev = be.g_att, be.graph, be.dev
    columns = be.norm_idx == 1 and s.world > 1
    in_off, need = self.stats_layout(s.n_local, int(sum(s.send_counts)), h)
    if columns and (ctx.n_buffers < 5 or need > ctx.buffer_bytes):
      raise _lib.GnpdeError('column statistics of %d bytes do not fit a fifth shared buffer of %d bytes' % (need, ctx.buffer_bytes))
    keep = dict(att=att,
                qk=torch.empty(max(s.n_local, 1), proj_m, dtype=torch.float32, device=dev),
                w=torch.empty(max(g_att.e, 1), dtype=torch.float32, device=dev),
                send=torch.empty(max(s.n_halo, 1), 2 * h, dtype=torch.float32, device=dev))
    keep['att_ws'] = torch.empty(max(int(L.gnpde_attention_workspace_bytes(g_att.ref(), ctypes.byref(att))), 256), dtype=torch.uint8, device=dev)
    keep['spmm_ws'] = torch.empty(max(int(L.gnpde_spmm_workspace_bytes(g_spmm.ref(), be.d)), 256), dtype=torch.uint8, device=dev)
    W = s.world
    metas = ctx.metas
    peer_in = [self.stats_layout(m['n_own'] + sum(m['recv_counts']), 0, h)[0] for m in metas]
    # peer p's send list is grouped by destination rank: my rows start behind what p sends to the ranks below me (= what they receive from p)
    rev_row0 = [sum(metas[q]['recv_counts'][p] for q in range(s.rank)) for p in range(W)]
    g = _lib.GeneralStruct()
    g.att_graph, g.spmm_graph, g.att = ctypes.pointer(g_att.struct), ctypes.pointer(g_spmm.struct), ctypes.pointer(att)
    g.qk, g.w, g.stats_send = keep['qk'].data_ptr(), keep['w'].data_ptr(), keep['send'].data_ptr()
    g.att_ws, g.att_ws_bytes = keep['att_ws'].data_ptr(), keep['att_ws'].numel()
    g.spmm_ws, g.spmm_ws_bytes = keep['spmm_ws'].data_ptr(), keep['spmm_ws'].numel()
    g.stats_buffer, g.in_offset = 4, in_off
    keep['tables'] = ((ctypes.c_int64 * W)(*peer_in), (ctypes.c_int64 * W)(*ctx.peer_buffer_bytes), (ctypes.c_int64 * W)(*rev_row0))
    g.peer_in_offset, g.peer_buffer_bytes, g.peer_rev_row0 = keep['tables']
    _lib.check(L.gnpde_sharded_solver_set_general(self.handle, ctypes.byref(g)))
    self._general = keep      # (the library holds raw addresses of all of these)

  def _chunked_push_order(self, chunks):
    """(order, chunk_ptr): the send slots grouped by the boundary chunk that computes their row (interior rows that peers
    read -- directed graphs -- ride with the first chunk), and inside a chunk merged over the destinations in proportion to
    what each is owed, like gnpde_push_order does for the whole list."""
    s = self.shard
    send_rows = s.send_idx.to(torch.int64).cpu()
    counts = [int(v) for v in s.send_counts]
    dest = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts, dtype=torch.int64))
    uppers = torch.tensor([c[1] for c in chunks], dtype=torch.int64)
    chunk_of = torch.searchsorted(uppers, send_rows, right=True).clamp_(max=len(chunks) - 1)
    order, ptr = [], [0]
    for c in range(len(chunks)):
      slots = torch.nonzero(chunk_of == c).flatten()
      if slots.numel():
        dc = dest[slots]
        per = torch.bincount(dc, minlength=len(counts)).to(torch.float64)
        # position (j + 1/2) / count inside the destination's own run of this chunk; stable sort keeps each run's order
        start = torch.zeros(len(counts), dtype=torch.int64)
        start[1:] = torch.cumsum(per.to(torch.int64), 0)[:-1]
        by_dest = torch.argsort(dc, stable=True)
        j = torch.empty(slots.numel(), dtype=torch.float64)
        j[by_dest] = (torch.arange(slots.numel()) - start[dc[by_dest]]).to(torch.float64)
        key = (j + 0.5) / per[dc]
        order += slots[torch.argsort(key, stable=True)].tolist()
      ptr.append(len(order))
    return order, ptr

  def integrate(self, y_own, x0_own=None, use_graph=True):
    """Owned rows of y(T) (a view of an internal buffer).  x0_own refreshes the persistent source term in place."""
    be = self.be
    if getattr(self, '_ran', False) and self.status()[0]:
      # the PREVIOUS solve lost a peer (its remaining waits returned at once, the result used stale halo rows): say so before
      # anything else is computed on it -- callers that need the verdict of the current solve call check() after it
      raise _lib.GnpdeError('sharded solve: a peer never published its boundary rows (exchange timed out); the result of the '
                            'previous integrate() is invalid')
    if x0_own is not None:
      be.x0.copy_(x0_own)
    self.y[:self.shard.n_own].copy_(y_own)
    _lib.check(_lib.lib().gnpde_sharded_solver_run(self.handle, _lib.ptr(self.y), int(bool(use_graph)), _lib.stream_of(self.y)))
    self._ran = True
    return self.y[:self.shard.n_own]

  def check(self):
    """Synchronise and raise if the solve that just ran lost a peer (P2P transport: bounded wait expired)."""
    if self.status()[0]:
      raise _lib.GnpdeError('sharded solve: a peer never published its boundary rows (exchange timed out)')

  def status(self):
    """(timed_out, epochs) -- synchronises; timed_out means a peer never published an evaluation's epoch."""
    import ctypes
    t, e = ctypes.c_int32(0), ctypes.c_int64(0)
    _lib.check(_lib.lib().gnpde_sharded_solver_status(self.handle, ctypes.byref(t), ctypes.byref(e)))
    return bool(t.value), int(e.value)

  def timing(self):
    """Exchange timing of the last run (gnpde_sharded_solver_timing): dict with per-evaluation push durations and waits in
    microseconds (lists), or None when the solver does not exchange / has no stamps.  Synchronises."""
    import ctypes
    n, rate = ctypes.c_int32(0), ctypes.c_int64(0)
    L = _lib.lib()
    _lib.check(L.gnpde_sharded_solver_timing(self.handle, None, 0, ctypes.byref(n), ctypes.byref(rate)))
    if n.value == 0 or rate.value <= 0:
      return None
    buf = (ctypes.c_int64 * (4 * n.value))()
    _lib.check(L.gnpde_sharded_solver_timing(self.handle, buf, n.value, ctypes.byref(n), ctypes.byref(rate)))
    t = torch.tensor(list(buf), dtype=torch.float64).view(n.value, 4)
    us = 1e6 / float(rate.value)
    ok = (t[:, 1] >= t[:, 0]) & (t[:, 3] >= t[:, 2]) & (t[:, 0] > 0)
    if not bool(ok.any()):
      return None
    t = t[ok]
    span = ((t[-1, 3] - t[0, 0]) * us / max(int(ok.sum()) - 0, 1)) if t.shape[0] > 1 else None
    return {'evaluations': int(ok.sum()), 'ticks_per_second': int(rate.value),
            'push_us': ((t[:, 1] - t[:, 0]) * us).tolist(), 'wait_us': ((t[:, 3] - t[:, 2]) * us).tolist(),
            # from the push of this evaluation to the moment every peer's rows are here: the exchange as the boundary pass sees it
            'push_to_landed_us': ((t[:, 3] - t[:, 0]) * us).tolist(),
            'per_evaluation_us': None if span is None else float(span)}

  def set_spin_limit(self, n):
    _lib.check(_lib.lib().gnpde_sharded_solver_set_spin_limit(self.handle, int(n)))

  def close(self):
    if getattr(self, 'handle', None) is not None and self.handle.value:
      _lib.lib().gnpde_sharded_solver_destroy(self.handle)
      self.handle = None
    if self._own_ctx and self.ctx is not None:
      self.ctx.close()
      self.ctx = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


class NativeShardedDopri5(object):
  """gnpde_dopri5_create_sharded over a NativeBackend's shard: dopri5 with the controller ON THE DEVICE of every rank.  A trial step
  is one hipGraph replay per rank (six evaluations, each push + interior rows + wait + boundary rows; the error norm as one double
  summed over the ranks inside the stream, csrc/sharded.hip p2p_sum_kernel); every rank's controller record holds the same bits,
  so every rank reads the same record once per batch of trial steps and queues the same number of them -- no host read and no
  torch.distributed call per trial step (ShardedSolver.integrate_adaptive: one all-to-all per evaluation and one all-reduce per
  trial step from Python).  Reference call: src/block_constant.py:57-62 with opt['method'] = 'dopri5' (run_GNN.py's default)."""

  def __init__(self, shard, backend, rtol, atol, n_rows_total, with_source=True, ctx=None, group=None, pair='dopri5'):
    import ctypes
    self.shard, self.be, self.pair = shard, backend, pair
    self.engine = NativeShardedSolver(shard, backend, None, method='rk4', with_source=with_source, ctx=ctx, group=group, boundary_chunks=1)
    L = _lib.lib()
    self.handle = None
    try:
      nbytes = int(L.gnpde_dopri5_sharded_workspace_bytes(self.engine.handle))
      self.ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=backend.dev)
      handle = ctypes.c_void_p()
      _lib.check(L.gnpde_dopri5_create_sharded(ctypes.byref(handle), self.engine.handle, float(rtol), float(atol), int(n_rows_total),
                                               _lib.ptr(self.ws), self.ws.numel()))
      self.handle = handle
      if pair != 'dopri5':      # torchdiffeq's adaptive_heun: the same controller, one evaluation per trial step (gnpde_dopri5_set_pair)
        _lib.check(L.gnpde_dopri5_set_pair(handle, {'adaptive_heun': 0, 'dopri5': 1}[pair]))
    except Exception:
      if self.handle is not None:
        L.gnpde_dopri5_destroy(self.handle)
        self.handle = None
      self.engine.close()
      raise
    self.out = backend.empty(shard.n_own)

  def integrate(self, y_own, x0_own, t0, t1, trials_per_sync=1, max_evals=0):
    """(owned rows of y(t1) -- a view of an internal buffer --, finished); x0_own refreshes the persistent source term in place."""
    import ctypes
    if x0_own is not None:
      self.be.x0.copy_(x0_own)
    y = _lib.f32rows(y_own, 'y0')
    fin = ctypes.c_int32(0)
    _lib.check(_lib.lib().gnpde_dopri5_run(self.handle, _lib.ptr(y), y.stride(0), float(t0), float(t1), _lib.ptr(self.out), self.out.stride(0),
                                           int(trials_per_sync), int(max_evals), ctypes.byref(fin), _lib.stream_of(y)))
    return self.out, bool(fin.value)

  def stats(self):
    import ctypes
    v = [ctypes.c_int32(0) for _ in range(5)]
    _lib.check(_lib.lib().gnpde_dopri5_stats(self.handle, *[ctypes.byref(x) for x in v]))
    return dict(zip(('evals', 'accepted', 'rejected', 'launches', 'syncs'), [x.value for x in v]))

  def close(self):
    if getattr(self, 'handle', None) is not None and self.handle.value:
      _lib.lib().gnpde_dopri5_destroy(self.handle)
      self.handle = None
    if getattr(self, 'engine', None) is not None:
      self.engine.close()
      self.engine = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


# --------------------------------------------------------------------------------------------------
# the operator surface: ODEblock -> odeint -> here, when torch.distributed is initialised and sharding is requested
# --------------------------------------------------------------------------------------------------
def shard_requested(func):
  """opt['gnpde_shard'] (or GNPDE_SHARD=1) with an initialised process group: the solves of this function run row-partitioned
  over the ranks of the default group (one process per GPU)."""
  if not (dist.is_available() and dist.is_initialized()):
    return False
  v = func.opt.get('gnpde_shard', os.environ.get('GNPDE_SHARD', '0'))
  return str(v).lower() not in ('0', '', 'false', 'none')


def _all_to_all_rows(out, inp, out_splits, in_splits, group=None):
  """all_to_all_single of row blocks over whatever backend the group has (gloo: staged through the host)."""
  if dist.get_backend(group) == 'nccl' or not inp.is_cuda:
    dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=group)
    return out
  o = torch.empty(out.shape, dtype=out.dtype)
  dist.all_to_all_single(o, inp.cpu(), list(out_splits), list(in_splits), group=group)
  out.copy_(o)
  return out


def _host_group_gather(t_dev, group=None):
  """all_gather of equally shaped device tensors over whatever backend the group has (gloo: staged through the host)."""
  world = dist.get_world_size(group)
  if dist.get_backend(group) == 'nccl':
    parts = [torch.empty_like(t_dev) for _ in range(world)]
    dist.all_gather(parts, t_dev, group=group)
    return parts
  host = t_dev.cpu()
  parts = [torch.empty_like(host) for _ in range(world)]
  dist.all_gather(parts, host, group=group)
  return [p.to(t_dev.device) for p in parts]


def _sharded_problem(func):
  """(kind, params) of a function object for NativeBackend, or GnpdeError if the row-partitioned solver does not cover it."""
  from .function_laplacian_diffusion import LaplacianODEFunc
  from .function_transformer_attention import ODEFuncTransformerAtt
  if isinstance(func, LaplacianODEFunc):
    w = func._edge_values()
    if w.dim() == 2:
      w = w.mean(dim=1)
    return 'laplacian', dict(edge_weight=w.detach())
  if isinstance(func, ODEFuncTransformerAtt):
    o, lay = func.opt, func.multihead_att_layer
    if (o['attention_type'] not in _lib.ATT_TYPES or o['mix_features']):
      raise _lib.GnpdeError('the row-partitioned solver covers GRAND-l and GRAND-nl with the scaled_dot / cosine_sim / pearson / exp_kernel '
                            'scores incl. the beltrami split kernel (softmax or squareplus, over rows or columns, re-weighted or not), '
                            'not mix_features; this configuration runs on one GPU only -- unset gnpde_shard')
    rw = lay.edge_weights.detach() if (o['reweight_attention'] and lay.edge_weights is not None) else None
    if getattr(lay, 'split_kernel', False):
      # BLEND's feature x positional kernel (reference src/function_transformer_attention.py:133-171) runs as ONE exp_kernel over the
      # concatenated, length-scaled projections (SpGraphTransAttentionLayer._split_qk_weights): heads of width 2 d_k, output_var =
      # ov_x ov_p, lengthscale 1 -- for the partitioned evaluation just another (W, b) pair
      wqk, bqk = lay.qk_weights()
      A2 = lay.kernel_att_dim
      ent = lay._bufs['qk']
      p = dict(Wq=wqk[:A2].detach(), bq=bqk[:A2].detach(), Wk=wqk[A2:].detach(), bk=bqk[A2:].detach(), heads=lay.h,
               norm_idx=int(o['attention_norm_idx']), square_plus=bool(o['square_plus']), att_type='exp_kernel',
               output_var=ent[3].detach(), lengthscale=ent[4].detach(), edge_weight=rw)
      return 'transformer', p
    p = dict(Wq=lay.Q.weight.detach(), bq=lay.Q.bias.detach(), Wk=lay.K.weight.detach(), bk=lay.K.bias.detach(), heads=lay.h,
             norm_idx=int(o['attention_norm_idx']), square_plus=bool(o['square_plus']), att_type=o['attention_type'])
    if o['attention_type'] == 'exp_kernel':
      p.update(output_var=lay.output_var.detach(), lengthscale=lay.lengthscale.detach())
    p['edge_weight'] = rw
    return 'transformer', p
  from .function_GAT_attention import ODEFuncAtt
  if isinstance(func, ODEFuncAtt):
    if func.opt['mix_features']:
      raise _lib.GnpdeError('the row-partitioned solver does not cover mix_features -- unset gnpde_shard')
    lay = func.multihead_att_layer
    return 'gat', dict(W=lay.proj_weight().detach(), a=lay.a.detach().reshape(-1), heads=lay.h, leaky_slope=float(lay.alpha),
                       norm_idx=int(func.opt['attention_norm_idx']), square_plus=False, att_type='GAT')
  raise _lib.GnpdeError('the row-partitioned solver covers LaplacianODEFunc, ODEFuncTransformerAtt and ODEFuncAtt, not %s'
                        % type(func).__name__)


def solve_sharded(func, y0, t, method, step_size, use_graph=True, group=None, rtol=None, atol=None):
  """torchdiffeq.odeint(func, y0, t, method='euler'|'rk4') of a block of this package on a row-partitioned graph, one
  process per GPU: every rank calls with the SAME replicated y0 / func (as the reference's model is replicated by
  nn.DataParallel, src/ray_tune.py:65-66, which cannot split a full-graph model); the graph is partitioned once per
  edge_index (the ranks share the search, PartitionPlan.search), each rank integrates its rows with the native sharded
  solver (csrc/sharded.hip, P2P transport inside one hipGraph per rank), and the rows are all-gathered so that every rank
  returns the full [2, n, d] result the block expects (reference call: src/block_constant.py:57-62)."""
  rank, world = dist.get_rank(group), dist.get_world_size(group)
  if not (y0.is_cuda and y0.dtype == torch.float32 and y0.dim() == 2):
    raise _lib.GnpdeError('sharded solve: the state must be a float32 [n, d] tensor on a HIP device')
  if func._needs_grad(y0):
    raise _lib.GnpdeError('sharded solve: inference only (no autograd through the partitioned solver); call under torch.no_grad()')
  dev, (n, d) = y0.device, y0.shape
  adaptive = method in ('dopri5', 'adaptive_heun')
  if adaptive:
    dts, n_evals = (), 0          # (evaluations are counted one by one, func._check_nfe, as on one GPU)
  else:
    grid = time_grid(t.detach().to('cpu'), step_size)
    dts = tuple((grid[1:] - grid[:-1]).tolist())
    n_evals = len(dts) * (4 if method == 'rk4' else 1)
    room = func.opt['max_nfe'] + 1 - func.nfe
    if n_evals > room:
      func.nfe += max(room, 0)
      from .utils import MaxNFEException
      raise MaxNFEException
  kind, params = _sharded_problem(func)
  ei = func.edge_index
  st = func.__dict__.setdefault('_shard_state', {})
  key = (id(ei), ei._version, tuple(ei.shape), world, rank, d, kind, str(dev), params.get('norm_idx', 0), params.get('square_plus', False),
         params.get('att_type', ''), params.get('edge_weight') is not None)
  ent = st.get(key)
  if ent is None:
    for old in st.values():
      old['close']()
    st.clear()
    per_rank = max(1, min(12, len(PartitionPlan.SEARCH_SPACE) // max(world, 1)))
    plan = PartitionPlan.search(ei, n, world, rank=rank, group_size=world, per_rank=per_rank, group=group)
    shard = plan.shard(rank)
    local = dict(params)
    if params.get('edge_weight') is not None:
      local['edge_weight'] = params['edge_weight'][shard.edge_ids.to(params['edge_weight'].device)]
    be = NativeBackend(shard, d, dev, kind, local, func.alpha_train, func.beta_train, not func.opt['no_alpha_sigmoid'])
    # normalisers that are not row-local (attention_norm_idx 1, squareplus) exchange BETWEEN the attention passes: they run the
    # Python-driven loop over torch.distributed (ShardedSolver + NativeBackend.rhs_stage_general), not the in-graph P2P solver
    # ... since round 6 inside the per-rank graph too (gnpde_sharded_solver_set_general) when the column statistics fit a fifth shared
    # buffer on EVERY rank (the ranks agree: all or none); GNPDE_SHARDED_HOST_EXCHANGES=1 keeps the Python-driven loop (A/B, tests)
    general = bool(getattr(be, 'general', False))
    in_graph_general = False
    if general and os.environ.get('GNPDE_SHARDED_HOST_EXCHANGES', '0') != '1':
      buf_bytes = (max(shard.n_local, 1) * d * 4 + 255) // 256 * 256
      fits = NativeShardedSolver.stats_layout(shard.n_local, int(sum(shard.send_counts)), be.heads)[1] <= buf_bytes
      votes = [None] * world
      if world > 1:
        dist.all_gather_object(votes, bool(fits), group=group)
      else:
        votes = [bool(fits)]
      in_graph_general = all(votes)
    ctx = None
    if not general:
      ctx = P2PContext(shard, d, 4, group=group)
    elif in_graph_general:
      ctx = P2PContext(shard, d, 5, group=group)
    ent = dict(edge_index=ei, plan=plan, shard=shard, be=be, ctx=ctx, solvers={}, own_ids=shard.own_old_ids.to(dev))

    def close(ent=ent):
      for sol in ent['solvers'].values():
        if hasattr(sol, 'close'):
          sol.close()
      ent['solvers'].clear()
      if ent['ctx'] is not None:
        ent['ctx'].close()
    ent['close'] = close
    st[key] = ent
  plan, shard, be = ent['plan'], ent['shard'], ent['be']
  local = dict(params)
  if params.get('edge_weight') is not None:
    local['edge_weight'] = params['edge_weight'][shard.edge_ids.to(params['edge_weight'].device)]
  be.refresh(func.alpha_train, func.beta_train, local)
  with_source = bool(func.opt['add_source'])
  skey = (method, dts, with_source)
  sol = ent['solvers'].get(skey)
  in_graph = (adaptive and ent['ctx'] is not None and d % 4 == 0 and t.dtype == torch.float32 and
              os.environ.get('GNPDE_SHARDED_HOST_CONTROLLER', '0') != '1')
  if in_graph:
    # dopri5 / adaptive_heun with the controller on every rank's device: trial steps as per-rank hipGraphs, the error norm summed over the ranks
    # inside the stream (NativeShardedDopri5) -- every rank holds the same controller record, reads it once per batch of trial steps
    from .utils import MaxNFEException
    from .odeint import end_points
    room = func.opt['max_nfe'] + 1 - func.nfe
    if room <= 0:
      raise MaxNFEException
    skey = (method, float(rtol), float(atol), with_source)
    sol = ent['solvers'].get(skey)
    if sol is None:
      for old in ent['solvers'].values():     # one live solver per function: its stage buffers are the shared P2P block
        if hasattr(old, 'close'):
          old.close()
      ent['solvers'].clear()
      sol = NativeShardedDopri5(shard, be, rtol, atol, n, with_source=with_source, ctx=ent['ctx'], group=group, pair=method)
      ent['solvers'][skey] = sol
    y_own = y0.detach()[ent['own_ids']]
    x0_own = None
    if with_source:
      if func.x0 is None:
        raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
      x0_own = func.x0.detach()[ent['own_ids']]
    tps = 8 if n * d < (1 << 22) else 1        # (the same rule as on one GPU, over the WHOLE state: the same batches on every rank)
    if os.environ.get('GNPDE_DOPRI5_TRIALS_PER_SYNC'):
      tps = max(1, int(os.environ['GNPDE_DOPRI5_TRIALS_PER_SYNC']))
    t0_, t1_ = end_points(t)
    z_own, finished = sol.integrate(y_own, x0_own, t0_, t1_, trials_per_sync=tps, max_evals=room)
    stats = sol.stats()
    func._dopri5_stats = stats
    spent = stats['evals']
    if not finished or spent > room:
      func.nfe += min(spent, room)
      raise MaxNFEException
    func.nfe += spent
    return _gather_full(z_own, y0, plan, shard, world, n, d, dev, group, func, 0)
  if adaptive:
    # where the in-graph solver does not apply (normalisers with exchanges between the attention passes, rows that are no multiple of
    # 16 bytes): the host controller over sharded evaluations (ShardedSolver.integrate_adaptive), one all-reduce per trial step
    if sol is None:
      for old in ent['solvers'].values():
        if hasattr(old, 'close'):
          old.close()
      ent['solvers'].clear()
      sol = ShardedSolver(shard, be, group)
      ent['solvers'][skey] = sol
    y_own = y0.detach()[ent['own_ids']]
    x0_own = None
    if with_source:
      if func.x0 is None:
        raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
      x0_own = func.x0.detach()[ent['own_ids']].contiguous()
    z_own = sol.integrate_adaptive(y_own, x0_own, t, rtol, atol, n, method=method, on_eval=func._check_nfe).clone()
    return _gather_full(z_own, y0, plan, shard, world, n, d, dev, group, func, 0)
  if getattr(be, 'general', False) and ent['ctx'] is None:
    if sol is None:
      ent['solvers'].clear()
      sol = ShardedSolver(shard, be, group)
      ent['solvers'][skey] = sol
    y_own = y0.detach()[ent['own_ids']]
    x0_own = None
    if with_source:
      if func.x0 is None:
        raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
      x0_own = func.x0.detach()[ent['own_ids']].contiguous()
    # the Python-driven loop builds its own grid from (T, step_size) starting at 0: it must be the grid the caller asked for
    T_loop = float(sum(dts))
    grid_loop = time_grid(torch.tensor([0.0, T_loop]), step_size)
    dts_loop = tuple((grid_loop[1:] - grid_loop[:-1]).tolist())
    if float(t[0]) != 0.0 or len(dts_loop) != len(dts) or any(abs(a - b) > 1e-6 * max(abs(b), 1.0) for a, b in zip(dts_loop, dts)):
      raise _lib.GnpdeError('sharded solve (normaliser with exchanges between the attention passes): the time grid %r from t[0] = %g '
                            'cannot be reproduced by the loop\'s grid %r' % (dts, float(t[0]), dts_loop))
    z_own = sol.integrate(y_own, x0_own, T_loop, step_size, method).clone()
    return _gather_full(z_own, y0, plan, shard, world, n, d, dev, group, func, n_evals)
  if sol is None:
    for old in ent['solvers'].values():     # one live solver per function: its stage buffers are the shared P2P block
      old.close()
    ent['solvers'].clear()
    T = float(sum(dts))
    sol = NativeShardedSolver(shard, be, T, step_size, method, with_source=with_source, ctx=ent['ctx'], group=group)
    if tuple(sol_dts(sol)) != dts:
      sol.close()
      raise _lib.GnpdeError('sharded solve: internal time grid mismatch')
    ent['solvers'][skey] = sol
  y_own = y0.detach()[ent['own_ids']]
  x0_own = None
  if with_source:
    if func.x0 is None:
      raise _lib.GnpdeError('add_source is set but x0 was never assigned (call ODEblock.set_x0)')
    x0_own = func.x0.detach()[ent['own_ids']]
  z_own = sol.integrate(y_own, x0_own, use_graph=use_graph)
  sol.check()                                       # synchronises; a lost peer raises here, on every rank that saw it
  return _gather_full(z_own, y0, plan, shard, world, n, d, dev, group, func, n_evals)


def _gather_full(z_own, y0, plan, shard, world, n, d, dev, group, func, n_evals):
  """Every rank gets the whole state back: pad to the largest part, all-gather the rows, put them at their original ids."""
  m = int(plan.counts.max())
  pad = torch.zeros(m, d, dtype=torch.float32, device=dev)
  pad[:shard.n_own] = z_own
  out = torch.empty(2, n, d, dtype=y0.dtype, device=dev)
  out[0].copy_(y0.detach())
  parts = _host_group_gather(pad, group)
  for p in range(world):
    ids = plan.shard_ids(p).to(dev)
    out[1][ids] = parts[p][:ids.numel()]
  func.nfe += n_evals
  return out


def sol_dts(solver):
  return solver.dts


def scatter_rows(x_global, shard):
  """Owned rows of a replicated global tensor, in the shard's local order."""
  return x_global[shard.own_old_ids.to(x_global.device)]


def gather_rows_all(y_own, plan, shard, group=None):
  """All-gather the owned rows and undo the partition permutation (tests / small graphs).  Every rank orders
  its own rows interior-first, so the original ids travel with the rows."""
  P = plan.world
  m = int(plan.counts.max())
  dev = y_own.device
  pad = torch.zeros(m, y_own.shape[1], dtype=y_own.dtype, device=dev)
  pad[:y_own.shape[0]] = y_own
  ids = torch.full((m,), -1, dtype=torch.long, device=dev)
  ids[:shard.n_own] = shard.own_old_ids.to(dev)
  parts = [torch.empty_like(pad) for _ in range(P)]
  id_parts = [torch.empty_like(ids) for _ in range(P)]
  dist.all_gather(parts, pad, group=group)
  dist.all_gather(id_parts, ids, group=group)
  out = torch.empty(plan.n, y_own.shape[1], dtype=y_own.dtype, device=dev)
  for p in range(P):
    ok = id_parts[p] >= 0
    out[id_parts[p][ok]] = parts[p][ok]
  return out


def negotiate_transport(ladder, phases, agree, notes, on_reject=None):
  """The first rung of `ladder` that EVERY rank gets through: for each candidate the ranks run the phases in lockstep --
  phase(cand) -> (ok, why), exceptions count as a failure -- and vote after each one (`agree(ok)` is True iff all ranks said ok).
  A candidate one rank cannot use is dropped by all of them (its reason goes to `notes[cand]`, `on_reject(cand)` releases what the
  attempt set up) and the next rung is tried; returns the chosen candidate or None.  Host logic only: bench_main hands in the
  device work (one evaluation / a two-step solve against the unpartitioned graph), tests/test_distributed_cpu.py scripted
  failures."""
  for cand in ladder:
    all_ok = True
    for k, phase in enumerate(phases):
      try:
        ok, why = phase(cand)
      except Exception as exc:   # noqa: BLE001 -- any failure of a transport means: try the next one
        ok, why = False, '%s%s: %s' % ('' if k == 0 else 'phase %d: ' % (k + 1), type(exc).__name__, str(exc)[:300])
      if why:
        notes[cand] = why
      all_ok = agree(ok)
      if not all_ok:
        if ok:
          notes[cand] = 'unavailable on another rank' if k == 0 else 'phase %d failed on another rank' % (k + 1)
        break
    if all_ok:
      return cand
    if on_reject is not None:
      on_reject(cand)
  return None


# --------------------------------------------------------------------------------------------------
# bench.py --gpus N  (launched by torch.distributed.run, one rank per GPU)
# --------------------------------------------------------------------------------------------------
def bench_main(args, rank, world, dev, emit=None):
  import gnpde_amd as G
  # GNPDE_RANKS_SHARE_DEVICE=1 (set by the caller, together with a device every rank can see): all ranks on ONE GPU, gloo for
  # the host-side collectives -- the P2P transport does not care which device a peer's memory is on.  Functional runs of the
  # multi-rank bench path on a single-GPU box; the numbers are not a scaling measurement.
  shared = os.environ.get('GNPDE_RANKS_SHARE_DEVICE', '0') == '1'
  kw = {}
  if 'RANK' not in os.environ:      # GNPDE_FORCE_SHARDED=1 python bench.py: the sharded driver as a single rank, no launcher
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    kw = dict(rank=0, world_size=1)
  if shared:
    dist.init_process_group('gloo', **kw)
  else:
    dist.init_process_group('nccl', device_id=dev, **kw)
  red = torch.device('cpu') if shared else dev          # where the small result reductions live
  cfg = G.synthetic.CONFIGS[args.graph]
  ei, n = G.synthetic.make_graph(args.graph, seed=args.seed, scale=args.scale)
  d = cfg['d']
  A, h = args.att_dim or cfg['att_dim'], args.heads or cfg['heads']
  ei_loops, _ = G.add_remaining_self_loops(ei, None, 1.0, n)      # as ODEFuncTransformerAtt.__init__ does
  if args.function == 'laplacian':                                  # the block's rw-normalised edge list (same loops)
    ei_loops, w_loops = G.get_rw_adj(ei, None, norm_dim=1, fill_value=1.0, num_nodes=n, dtype=torch.float32)
  t0 = time.perf_counter()
  # the ranks share the search for the partition (PartitionPlan.search): 9 settings each at 8 ranks = all 72
  per_rank = max(1, min(12, len(PartitionPlan.SEARCH_SPACE) // max(world, 1)))
  plan = PartitionPlan.search(ei_loops, n, world, rank=rank, group_size=world, per_rank=per_rank)
  shard = plan.shard(rank)
  t_plan = time.perf_counter() - t0
  g = torch.Generator().manual_seed(args.seed)
  x = torch.randn(n, d, generator=g)
  g2 = torch.Generator().manual_seed(args.seed + 1)
  params = dict(Wq=torch.randn(A, d, generator=g2) / d ** 0.5, Wk=torch.randn(A, d, generator=g2) / d ** 0.5,
                bq=torch.zeros(A), bk=torch.zeros(A), heads=h)
  kind = args.function
  if kind == 'laplacian':
    params = dict(edge_weight=w_loops[shard.edge_ids])
  be = NativeBackend(shard, d, dev, kind, params, torch.tensor(0.0), torch.tensor(0.1), True)
  x_own = scatter_rows(x, shard).to(dev)
  K, W = args.steps, args.warmup
  use_graph = not args.no_graph
  from . import ops

  def agree(ok):
    """True iff every rank says so (host-side, any backend)."""
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    return all(flags)

  # reference for the self-checks: the SAME problem on the unpartitioned graph, computed natively on this rank's GPU --
  # one evaluation f(x), and (unpartitioned_solve) whole rk4 solves, so that the sharded result is compared after every
  # stage buffer has been pushed into, read and pushed into AGAIN (a stale halo line would not show in one evaluation)
  own_ids = shard.own_old_ids.to(dev)
  with torch.no_grad():
    xg = x.to(dev)
    alpha_d, beta_d = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
    if kind == 'transformer':
      full = CSRGraph(ei_loops.to(dev), n)
      wqk = torch.cat([params['Wq'], params['Wk']]).to(dev).contiguous()
      bqk = torch.zeros(2 * A, device=dev)
      qk = ops.linear(xg, wqk, bqk)
      st = ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False, q=qk, k=qk[:, A:], ldqk=2 * A)
      w_full, _, _ = ops.edge_attention(full, st, True, False, False, like=xg)
      f_full = ops.spmm_rhs(full, w_full, xg, alpha_d, beta_d, xg, True)
      full_kw = dict(proj_w=wqk, proj_b=bqk, att=ops.attention_struct(_lib.ATT_SCALED_DOT, h, A, 0, False))
      full_kind = _lib.RHS_TRANSFORMER
      del qk, w_full, st
    else:
      e_rw, w_rw = G.get_rw_adj(ei, None, norm_dim=1, fill_value=1.0, num_nodes=n, dtype=torch.float32)
      full = CSRGraph(e_rw.to(dev), n)
      w_full_csr = ops.edge_to_csr_mean(full, w_rw.to(dev))
      f_full = ops.spmm_rhs(full, w_full_csr, xg, alpha_d, beta_d, xg, True)
      full_kw = dict(w_csr=w_full_csr)
      full_kind = _lib.RHS_LAPLACIAN
    ref_own = f_full[own_ids].clone()
    del f_full

  def unpartitioned_solve(T):
    """Owned rows of y(T), rk4 with step 1 from y(0) = x: the single-GPU solver (csrc/solver.hip) on the whole graph."""
    grid = time_grid(torch.tensor([0.0, float(T)]), 1.0)
    desc = ops.RhsDescriptor(full_kind, full, d, d, alpha_d, beta_d, xg, True, **full_kw)
    sol = ops.FixedStepSolver(desc, 'rk4', (grid[1:] - grid[:-1]).tolist(), dev)
    try:
      yy = xg.clone()
      sol.run(yy, use_graph=False)
      torch.cuda.synchronize(dev)
      return yy[own_ids].clone()
    finally:
      sol.close()

  def solve_error(y_own, ref):
    return float((y_own - ref).abs().max() / ref.abs().max().clamp_min(1e-30))

  # two rk4 steps = 8 evaluations: each of the four stage buffers is the exchanged one twice
  T_CHECK = 2.0
  ref_check, ref_check_why = None, None
  try:
    with torch.no_grad():
      ref_check = unpartitioned_solve(T_CHECK)
  except Exception as exc:   # noqa: BLE001 -- the transports are then judged on the one-evaluation check alone
    ref_check_why = '%s: %s' % (type(exc).__name__, str(exc)[:200])

  def one_eval_error(f_own):
    return float((f_own - ref_own).abs().max() / ref_own.abs().max().clamp_min(1e-30))

  # Transports, best first; a transport is taken only if EVERY rank could set it up and reproduces the unpartitioned graph
  # through it, exchange included -- first ONE evaluation of f (an euler step of size 1), then, ranks still in lockstep, a
  # two-step rk4 solve in the launch mode the timed run will use (8 evaluations: every stage buffer exchanged twice):
  #   p2p    boundary rows pushed into the peers' IPC-mapped halo regions inside the per-rank hipGraph
  #   rccl   the same native solver with grouped ncclSend / ncclRecv, eager launches (RCCL does not capture on this HIP)
  #   torch  the Python-driven loop over torch.distributed.all_to_all_single (round 1)
  ladder = [t for t in os.environ.get('GNPDE_BENCH_TRANSPORTS', 'p2p,rccl,torch').split(',') if t]
  if os.environ.get('GNPDE_SHARDED_PYTHON_LOOP', '0') == '1':
    ladder = ['torch']
  chosen, notes = None, {}
  ctx = solver = run = None
  err_local = float('inf')
  err_check = None
  have_ref_check = agree(ref_check is not None)   # (every rank takes the same path through the checks)
  if not have_ref_check:
    notes['two_step_check'] = 'skipped: %s' % (ref_check_why or 'reference solve unavailable on another rank')

  def close_quietly(obj):
    try:
      if obj is not None and hasattr(obj, 'close'):
        obj.close()
    except Exception:   # noqa: BLE001
      pass

  state = {'ctx': None, 'err_local': float('inf'), 'err_check': None}

  def one_eval_phase(cand):
    """ONE evaluation of f through the candidate's exchange (an euler step of size 1) against the unpartitioned graph."""
    chk = None
    try:
      if cand == 'torch':
        chk = ShardedSolver(shard, be)
        chk.y[:shard.n_own].copy_(x_own)
        chk.exchange(chk.y)
        f_own = be.empty(shard.n_own)
        be.rhs_stage(chk.y, x_own, _lib.STAGE_RHS, out_k=f_own)
        torch.cuda.synchronize(dev)
        state['err_local'] = one_eval_error(f_own)
      else:
        if cand == 'p2p' and state['ctx'] is None:
          state['ctx'] = P2PContext(shard, d, 4)
        chk = NativeShardedSolver(shard, be, 1.0, 1.0, 'euler', transport=cand, ctx=state['ctx'] if cand == 'p2p' else None)
        f_own = chk.integrate(x_own, x_own, use_graph=False).clone() - x_own
        torch.cuda.synchronize(dev)
        timed_out = chk.status()[0]
        state['err_local'] = one_eval_error(f_own)
        if timed_out:
          return False, 'a peer never published its boundary rows (exchange timed out)'
      if not (state['err_local'] <= 1e-4):
        return False, 'one evaluation differs from the unpartitioned graph by %.3e' % state['err_local']
      return True, None
    finally:
      close_quietly(chk)

  def two_step_phase(cand):
    """Ranks still in lockstep: a two-step rk4 solve in the launch mode the timed run will use (8 evaluations: every stage buffer
    exchanged twice)."""
    chk = None
    try:
      ok, why = True, None
      if cand == 'torch':
        chk = ShardedSolver(shard, be)
        y2 = chk.integrate(x_own, x_own, T_CHECK, 1.0, 'rk4').clone()
        torch.cuda.synchronize(dev)
      else:
        chk = NativeShardedSolver(shard, be, T_CHECK, 1.0, 'rk4', transport=cand, ctx=state['ctx'] if cand == 'p2p' else None)
        y2 = chk.integrate(x_own, x_own, use_graph=bool(use_graph and cand == 'p2p')).clone()
        torch.cuda.synchronize(dev)
        if chk.status()[0]:
          ok, why = False, 'two-step solve: a peer never published its boundary rows (exchange timed out)'
      state['err_check'] = solve_error(y2, ref_check)
      if ok and not (state['err_check'] <= 1e-4):
        ok, why = False, 'two-step rk4 solve differs from the unpartitioned graph by %.3e' % state['err_check']
      return ok, why
    finally:
      close_quietly(chk)

  def reject(cand):
    if cand == 'p2p' and state['ctx'] is not None:
      close_quietly(state['ctx'])
      state['ctx'] = None

  with torch.no_grad():
    phases = [one_eval_phase] + ([two_step_phase] if have_ref_check else [])
    chosen = negotiate_transport(ladder, phases, agree, notes, on_reject=reject)
    ctx, err_local, err_check = state['ctx'], state['err_local'], state['err_check']
    if chosen is None:
      raise _lib.GnpdeError('no halo transport works on this system: %r' % (notes,))
    python_loop = chosen == 'torch'
    graph_mode = use_graph and chosen == 'p2p'
    if python_loop:
      solver = ShardedSolver(shard, be)
      run = lambda T: solver.integrate(x_own, x_own, float(T), 1.0, 'rk4')   # noqa: E731
      if W > 0:
        run(W)
    else:
      if W > 0:
        warm = NativeShardedSolver(shard, be, float(W), 1.0, 'rk4', transport=chosen, ctx=ctx)
        warm.integrate(x_own, x_own, use_graph=graph_mode)
        torch.cuda.synchronize(dev)
        dist.barrier()
        warm.close()
      # P2P: how many row ranges the boundary pass is cut into (each range's rows of the next stage input are pushed right
      # behind it; NativeShardedSolver) is chosen by the clock -- more ranges hide more of the exchange but add launches
      cands = [1]
      if chosen == 'p2p':
        cands = [int(v) for v in os.environ.get('GNPDE_BOUNDARY_CHUNKS', os.environ.get('GNPDE_BENCH_CHUNKS', '1,2,4')).split(',') if v]
      solver, solver_kc, chunk_ms = None, None, {}
      for kc in cands:
        cand_solver, best, why = None, float('inf'), None
        try:
          cand_solver = NativeShardedSolver(shard, be, float(K), 1.0, 'rk4', transport=chosen, ctx=ctx, boundary_chunks=kc)
          cand_solver.integrate(x_own, x_own, use_graph=graph_mode)           # untimed: captures the K-step graph
          torch.cuda.synchronize(dev)
          if cand_solver.status()[0]:
            why = 'a peer never published its boundary rows'
        except Exception as exc:   # noqa: BLE001 -- a range count that cannot be set up is skipped, like a transport
          why = '%s: %s' % (type(exc).__name__, str(exc)[:200])
        if not agree(why is None):
          notes['boundary_chunks=%d' % kc] = why or 'failed on another rank'
          close_quietly(cand_solver)
          continue
        for _ in range(2 if len(cands) > 1 else 0):
          torch.cuda.synchronize(dev)
          dist.barrier()
          t0 = time.perf_counter()
          cand_solver.integrate(x_own, x_own, use_graph=graph_mode)
          torch.cuda.synchronize(dev)
          dist.barrier()
          best = min(best, time.perf_counter() - t0)
        tb = torch.tensor([best if best < float('inf') else 0.0], dtype=torch.float64, device=red)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        chunk_ms[kc] = round(float(tb.item()) * 1e3 / K, 4)
        if solver is None or chunk_ms[kc] < chunk_ms[solver_kc]:    # (the same verdict on every rank: the times are the max over ranks)
          close_quietly(solver)
          solver, solver_kc = cand_solver, kc
        else:
          close_quietly(cand_solver)
      if solver is None:
        raise _lib.GnpdeError('no boundary range count of %r could be set up: %r' % (cands, notes))
      run = lambda T: solver.integrate(x_own, x_own, use_graph=graph_mode)   # noqa: E731
    times = []
    for _ in range(max(getattr(args, 'replays', 1), 1)):
      torch.cuda.synchronize(dev)
      dist.barrier()
      torch.cuda.synchronize(dev)
      t0 = time.perf_counter()
      y = run(K)
      torch.cuda.synchronize(dev)
      dist.barrier()
      torch.cuda.synchronize(dev)
      times.append(time.perf_counter() - t0)
    elapsed = sorted(times)[len(times) // 2]
    y = y.clone()
    # the TIMED solve against the same K steps on the unpartitioned graph (outside the timed region; the single-GPU solver
    # on every rank's own device, compared on the rank's own rows)
    err_solve, solve_why = 0.0, None
    t_single = None
    try:
      torch.cuda.synchronize(dev)
      t0 = time.perf_counter()
      ref_K = unpartitioned_solve(K)
      t_single = time.perf_counter() - t0       # eager launches of the single-GPU solver on this rank's device (incl. set-up)
      err_solve = solve_error(y, ref_K)
    except Exception as exc:   # noqa: BLE001 -- reported, not fatal: the line still carries the one-evaluation check
      solve_why = '%s: %s' % (type(exc).__name__, str(exc)[:200])
    # exchange timing of the last timed replay, written by the kernels inside the graph (P2P transport)
    exch = None if python_loop else solver.timing()
    # this rank's aggregation on its shard (all local rows, one launch per stage variant), by the gather model
    shard_roof = None
    try:
      E_loc = int(shard.edge_index.shape[1])
      ub = [torch.randn(shard.n_local, d, device=dev) for _ in range(4)]
      wl = torch.rand(max(E_loc, 1), device=dev) / 16
      a0, b0 = torch.tensor([0.0], device=dev), torch.tensor([0.1], device=dev)
      x0l = torch.randn(shard.n_own, d, device=dev)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

      def agg():
        ops.spmm_rhs(be.graph, wl, ub[0], a0, b0, x0l, True, out=ub[1][:shard.n_own])
      agg()
      torch.cuda.synchronize(dev)
      e0.record()
      for _ in range(10):
        agg()
      e1.record()
      torch.cuda.synchronize(dev)
      t_agg = e0.elapsed_time(e1) * 1e-3 / 10
      b_agg = E_loc * (8 + 4 * d) + shard.n_own * (4 + 8 * d) + 4 * d * shard.n_own
      shard_roof = {'avg_launch_us': round(t_agg * 1e6, 2), 'algorithmic_bytes_per_launch': b_agg,
                    'gather_model_gbs': round(b_agg / t_agg / 1e9, 1), 'local_rows': shard.n_own, 'local_entries': E_loc,
                    'local_table_mib': round(shard.n_local * d * 4 / 2 ** 20, 1)}
      del ub, wl, x0l
    except Exception as exc:   # noqa: BLE001
      shard_roof = {'error': repr(exc)[:200]}
  have_solve = agree(solve_why is None)
  es = torch.tensor([err_solve if solve_why is None and err_solve == err_solve else 3.0e38], dtype=torch.float32).to(red)
  dist.all_reduce(es, op=dist.ReduceOp.MAX)
  ec = torch.tensor([err_check if err_check is not None and err_check == err_check else -1.0], dtype=torch.float32).to(red)
  dist.all_reduce(ec, op=dist.ReduceOp.MAX)
  err = torch.tensor([err_local], dtype=torch.float32).to(red)
  dist.all_reduce(err, op=dist.ReduceOp.MAX)
  timed_out = False if python_loop else solver.status()[0]
  el = torch.tensor([elapsed], dtype=torch.float64, device=red)
  dist.all_reduce(el, op=dist.ReduceOp.MAX)
  finite = torch.tensor([1.0 if bool(torch.isfinite(y).all()) else 0.0], device=red)
  dist.all_reduce(finite, op=dist.ReduceOp.MIN)
  halo = torch.tensor([float(shard.n_halo), float(shard.n_own), float(shard.edge_index.shape[1]),
                       float(max(shard.recv_counts) if shard.recv_counts else 0), float(shard.n_interior)], device=red)
  halo_max = halo.clone()
  dist.all_reduce(halo_max, op=dist.ReduceOp.MAX)
  # exchange timing: mean over the evaluations of the last replay on every rank, then max / mean over the ranks
  import statistics
  ex_loc = [-1.0, -1.0, -1.0, -1.0]
  if exch:
    ex_loc = [statistics.fmean(exch['push_us']), statistics.fmean(exch['wait_us']), statistics.fmean(exch['push_to_landed_us']),
              exch['per_evaluation_us'] or -1.0]
  ex = torch.tensor(ex_loc, dtype=torch.float64, device=red)
  ex_max = ex.clone()
  dist.all_reduce(ex_max, op=dist.ReduceOp.MAX)
  ex_sum = ex.clone()
  dist.all_reduce(ex_sum, op=dist.ReduceOp.SUM)
  if rank == 0:
    elapsed = float(el.item())
    E = int(ei_loops.shape[1])
    names = {'arxiv': 'ogbn-arxiv', 'cora': 'Cora', 'rmat': 'RMAT-2M'}
    out = {
      'metric': 'ODE steps/sec (full-graph diffusion), %s d=%d rk4' % (names.get(args.graph, args.graph), d),
      'value': round(K / elapsed, 3), 'unit': 'steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
      'ms_per_step': round(1e3 * elapsed / K, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
      'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': 'synthetic %s-shaped graph, GRAND-%s, rk4 3/8-rule, step_size 1, T=%d, rows '
                             'partitioned over %d GPUs, %s'
                             % (names.get(args.graph, args.graph),
                                'nl scaled_dot softmax attention add_source' if kind == 'transformer' else 'l', K, world,
                                {'p2p': 'boundary rows pushed into the peers\' halo regions (IPC-mapped, xGMI stores + epoch '
                                        'flags) once per evaluation inside the per-rank hipGraph',
                                 'rccl': 'boundary rows exchanged by grouped ncclSend / ncclRecv once per evaluation '
                                         '(native solver, eager launches)',
                                 'torch': 'boundary rows exchanged by torch.distributed.all_to_all_single once per '
                                          'evaluation (Python-driven loop)'}[chosen]),
                 'graph': args.graph, 'nodes': n, 'edges_with_self_loops': E, 'd': d, 'attention_dim': A, 'heads': h,
                 'rhs_evals_per_step': 4, 'edge_cut': round(plan.edge_cut(), 4),
                 'max_halo_rows': int(halo_max[0].item()), 'max_owned_rows': int(halo_max[1].item()),
                 'max_local_edges': int(halo_max[2].item()),
                 # what one evaluation moves: every rank receives its halo rows (4 d bytes each) before its boundary rows can be
                 # evaluated; xGMI is point to point, so the rows that come from ONE peer share one link
                 'max_rows_from_one_peer': int(halo_max[3].item()),
                 'max_bytes_on_one_link_per_evaluation': int(halo_max[3].item()) * 4 * d,
                 'max_interior_rows': int(halo_max[4].item()),
                 'partition_seconds': round(t_plan, 2),
                 # settings of the partitioner scored by the ranks (busiest link, busiest rank), cheapest first: the one used, and the spread
                 'partition_search': None if not plan.candidates else {
                   'scored': len(plan.candidates), 'used': plan.candidates[0], 'median_cost': plan.candidates[len(plan.candidates) // 2]['cost'],
                   'worst_cost': plan.candidates[-1]['cost']},

                 'finite': bool(finite.item() == 1.0), 'exchange_timed_out': timed_out, 'ranks_share_one_device': shared,
                 'transport': chosen, 'transports_rejected': notes,
                 'boundary_chunks': None if python_loop else solver_kc,
                 'boundary_chunks_ms_per_step': None if python_loop or len(chunk_ms) < 2 else {str(k): v for k, v in chunk_ms.items()},
                 'driver': 'python loop' if python_loop else 'native, hipGraph %s' % graph_mode,
                 'replays': len(times),
                 'sharded_vs_unpartitioned_one_eval_rel_max': float(err.item()),
                 'sharded_vs_unpartitioned_two_step_rel_max': float(ec.item()) if float(ec.item()) >= 0 else None,
                 'sharded_vs_unpartitioned_timed_solve_rel_max': float(es.item()) if have_solve else None,
                 'timed_solve_check': ('own rows of y(T=%d) of the timed run against the single-GPU solver on the unpartitioned '
                                       'graph (max over ranks)' % K) if have_solve else 'unavailable: %s' % (solve_why or 'failed on another rank')},
      'roofline': None, 'cpu_baseline': None,
    }
    link_rows = int(halo_max[3].item())
    link_bytes = link_rows * 4 * d
    # ---- the exchange as measured by the kernels (P2P transport), next to what DESIGN.md section 6 predicts from the partition
    if exch and float(ex_max[0].item()) >= 0:
      push_max, wait_max, landed_max = float(ex_max[0].item()), float(ex_max[1].item()), float(ex_max[2].item())
      out['exchange'] = {
        'source': 'wall-clock stamps written inside the hipGraph by push_rows_kernel / wait_flags_kernel of the last timed replay '
                  '(gnpde_sharded_solver_timing); means over the evaluations, then max / mean over the ranks',
        'push_us_max_rank': round(push_max, 2), 'push_us_mean_rank': round(float(ex_sum[0].item()) / world, 2),
        'wait_us_max_rank': round(wait_max, 2), 'wait_us_mean_rank': round(float(ex_sum[1].item()) / world, 2),
        'push_to_all_landed_us_max_rank': round(landed_max, 2),
        'per_evaluation_us_max_rank': round(float(ex_max[3].item()), 2) if float(ex_max[3].item()) > 0 else None,
        'busiest_link_bytes_per_evaluation': link_bytes,
        # the busiest link's rows over the slowest rank's push: a LOWER bound of that link's rate (the push kernel serves all
        # links at once and ends when the slowest of them has taken its rows)
        'busiest_link_gbs_lower_bound': round(link_bytes / (push_max * 1e-6) / 1e9, 2) if push_max > 0 else None,
        'ranks_share_one_device': shared}
    else:
      out['exchange'] = None
    per_eval_1 = None if t_single is None else t_single / (4.0 * K)
    out['model'] = {
      'what': 'DESIGN.md section 6: per evaluation a rank waits for max(interior compute, busiest link) and then runs its boundary '
              'rows; compute = the single-GPU evaluation split evenly over the ranks; xGMI link at 50 / 76 GB/s per direction',
      'single_gpu_us_per_evaluation_eager': None if per_eval_1 is None else round(per_eval_1 * 1e6, 2),
      'compute_us_per_evaluation_per_rank': None if per_eval_1 is None else round(per_eval_1 * 1e6 / world, 2),
      'busiest_link_us_at_50_gbs': round(link_bytes / 50e9 * 1e6, 2), 'busiest_link_us_at_76_gbs': round(link_bytes / 76e9 * 1e6, 2),
      'interior_row_share': round(float(halo_max[4].item()) / max(float(halo_max[1].item()), 1.0), 3),
      'predicted_speedup_range': None if per_eval_1 is None else [
        round(per_eval_1 / max(per_eval_1 / world, link_bytes / gbs + (1.0 - float(halo_max[4].item()) / max(float(halo_max[1].item()), 1.0)) * per_eval_1 / world), 2)
        for gbs in (50e9, 76e9)],
      'measured_us_per_evaluation': round(1e6 * elapsed / (4.0 * K), 2)}
    if shard_roof is not None and 'error' not in shard_roof:
      resident = shard_roof['local_table_mib'] < 256
      out['roofline'] = dict(shard_roof, kernel='CSR aggregation + fused epilogue on rank 0\'s shard (all local rows in one launch)',
                             bound='mall' if resident else 'hbm', achieved=shard_roof['gather_model_gbs'], peak=8000.0,
                             unit='GB/s', frac=round(shard_roof['gather_model_gbs'] / 8000.0, 4),
                             frac_is='frac_algorithmic: gather-model bytes of the launch / its time / the 8 TB/s HBM peak (no counter '
                                     'traffic in the multi-rank line; exceeds 1 when the shard\'s table is cache-resident)', traffic=None,
                             note='per-GPU figure of rank 0; the gathered table of a shard (own + halo rows) '
                                  + ('fits the Infinity Cache, so HBM\'s peak is not its ceiling (see the 1-GPU line\'s measured '
                                     'ceiling)' if resident else 'exceeds the Infinity Cache: fraction of the HBM peak'))
    else:
      out['roofline'] = shard_roof
    out['north_star_multi_gpu'] = ('BASELINE configs[4] (R-MAT 2M / 40M, d = 256: `--graph rmat`) is the configuration the north star names for 8 GPUs '
                                   '-- DESIGN.md section 6 predicts ~5x there and 2.3 - 3.0x for the ogbn-arxiv shape, whose 87-MB state leaves '
                                   'every rank launch- and link-latency bound; this line is `--graph %s`, its own prediction is `model`' % args.graph)
    out['cpu_baseline'] = None     # (timed on rank 0 at N = 1 only, by contract)
    out['config']['parallelism'] = 'rows%d' % world
    out['config']['ranks_seen'] = dist.get_world_size()
    if emit is not None:
      emit(out)                      # bench.py: the full object on its own line, the compact headline LAST
    else:
      print(json.dumps(out))
  dist.destroy_process_group()
