"""Prepared graph: the reference hands `edge_index` (int64 COO [2,E], any order) to every op
(reference src/function_transformer_attention.py:35,190-191); here it is converted ONCE per
edge_index tensor into CSR + permutation + CSC view + long-row chunk list by the native host code
(csrc/graph_prep.cpp) and kept resident in HBM as int32 arrays."""
import ctypes
import os
import numpy as np
import torch

from . import _lib


LONG_ROW = 512     # GNPDE_LONG_ROW
L2_BYTES_TOTAL = 32 << 20          # 8 XCDs x 4 MiB
LOCALITY_PART_BYTES = 5.5 * 2 ** 20   # table bytes per part of the locality relabelling (two parts per XCD at the ogbn-arxiv size)
LOCALITY_MIN_GAIN = 1.02           # auto mode: keep the relabelled graph when a timed aggregation on it is at least this much faster
XCD_IMBALANCE_LIMIT = 1.03   # contiguous eighths are kept while the slowest XCD has at most 3 % more than the mean
_LOCALITY_DECISIONS = {}      # (nodes, entries, width, device, edge-set fingerprint) -> (best order, measured gains): one timing probe per graph and process
_LOCALITY_DECISIONS_MAX = 64  # bounded: a block that builds a new edge set every training step must not grow it without end


def contiguous_deal_imbalance(rowptr, row_begin, row_end):
  """max / mean of the work the 8 XCDs get when an aggregation launch over the rows [row_begin, row_end) gives XCD x the
  x-th contiguous eighth (csrc/spmm.hip): entries + 3 per row for rows of up to GNPDE_LONG_ROW entries; longer rows are
  512-entry chunks dealt round robin (an equal share for everyone).  1.0 = perfectly even."""
  rn = int(row_end) - int(row_begin)
  if rn <= 0:
    return 1.0
  rp = rowptr[row_begin:row_end + 1].long()
  lens = rp[1:] - rp[:-1]
  short = lens <= LONG_ROW
  work = torch.where(short, lens + 3, torch.zeros_like(lens)).double()
  per = (rn + 7) // 8
  padded = torch.zeros(8 * per, dtype=torch.float64, device=work.device)
  padded[:rn] = work
  rows = padded.view(8, per).sum(dim=1)
  total = rows + lens[~short].sum().double() / 8.0
  mean = total.mean()
  return float(total.max() / mean) if float(mean) > 0 else 1.0


def build_arrays_on_device(edge_index, n):
  """The arrays of gnpde_graph_build, computed where the edge list already lives (sort / scan / gather ops), so
  that blocks which hand over a NEW edge set every training forward (hard attention, rewiring; reference
  src/block_transformer_hard_attention.py:55-61) pay neither the device->host copy of [2,E] int64 nor the host
  counting sort.  Element for element the same result as csrc/graph_prep.cpp: both sorts are stable."""
  dev = edge_index.device
  e = int(edge_index.shape[1])
  row, col = edge_index[0].long(), edge_index[1].long()
  i32 = dict(dtype=torch.int32, device=dev)
  out = {}
  if e > 0:
    lo = torch.minimum(row.min(), col.min())
    hi = torch.maximum(row.max(), col.max())
    lo, hi = int(lo), int(hi)
    if lo < 0 or hi >= n:
      raise _lib.GnpdeError('libgnpde_hip error -1: graph_build: edge index outside [0,%d)' % n)
  perm = torch.sort(row, stable=True).indices
  colidx = col[perm]
  deg = torch.bincount(row, minlength=n) if e > 0 else torch.zeros(n, dtype=torch.long, device=dev)
  rowptr = torch.zeros(n + 1, dtype=torch.long, device=dev)
  rowptr[1:] = torch.cumsum(deg, 0)
  cscpos = torch.sort(colidx, stable=True).indices
  cdeg = torch.bincount(col, minlength=n) if e > 0 else torch.zeros(n, dtype=torch.long, device=dev)
  cscptr = torch.zeros(n + 1, dtype=torch.long, device=dev)
  cscptr[1:] = torch.cumsum(cdeg, 0)
  long_rows = torch.nonzero(deg > LONG_ROW).flatten()
  long_cols = torch.nonzero(cdeg > LONG_ROW).flatten()
  chunks_per = (deg[long_rows] + LONG_ROW - 1) // LONG_ROW
  chunk_ptr = torch.zeros(long_rows.numel() + 1, dtype=torch.long, device=dev)
  chunk_ptr[1:] = torch.cumsum(chunks_per, 0)
  which = torch.repeat_interleave(torch.arange(long_rows.numel(), device=dev), chunks_per)   # long-row slot of each chunk
  within = torch.arange(which.numel(), device=dev) - chunk_ptr[which]
  chunk_row = long_rows[which]
  chunk_begin = rowptr[chunk_row] + within * LONG_ROW
  chunk_end = torch.minimum(chunk_begin + LONG_ROW, rowptr[chunk_row + 1])
  rows16 = torch.nonzero((deg >= 1) & (deg <= 16)).flatten()
  rows64 = torch.nonzero((deg > 16) & (deg <= LONG_ROW)).flatten()
  # longest first, ties in row order (as csrc/graph_prep.cpp: the long rows of this class start at time 0 of the
  # row-attention launch instead of forming its tail; the records are independent, their order enters no result)
  rows64 = rows64[torch.sort(deg[rows64], descending=True, stable=True).indices]
  listed = torch.cat([rows16, rows64])
  bins = torch.stack([listed, rowptr[listed], deg[listed], torch.zeros_like(listed)], dim=1).reshape(-1)

  def fit(t, size):
    buf = torch.zeros(max(int(size), 1), **i32)
    buf[:t.numel()] = t.to(torch.int32)
    return buf
  out['rowptr'] = rowptr.to(torch.int32)
  out['colidx'] = fit(colidx, e)
  out['perm'] = fit(perm, e)
  out['rowidx'] = fit(row[perm], e)
  out['cscptr'] = cscptr.to(torch.int32)
  out['cscpos'] = fit(cscpos, e)
  nlr, nlc = int(long_rows.numel()), int(which.numel())
  out['long_rows'] = fit(long_rows, nlr)
  out['long_chunk_ptr'] = chunk_ptr.to(torch.int32)
  out['long_chunk_row'] = fit(chunk_row, nlc)
  out['long_chunk_begin'] = fit(chunk_begin, nlc)
  out['long_chunk_end'] = fit(chunk_end, nlc)
  out['long_cols'] = fit(long_cols, long_cols.numel())
  out['bin_rows'] = fit(bins, 4 * max(n, 1))
  out['long_chunk_first'] = fit(chunk_ptr[which], nlc)
  counts = dict(n_long_rows=nlr, n_long_chunks=nlc, n_long_cols=int(long_cols.numel()), n_bin16=int(rows16.numel()),
                n_bin64=int(rows64.numel()), n_bin_le64=int((deg[rows64] <= 64).sum()) if rows64.numel() else 0, max_row_len=int(deg.max()) if n > 0 and e > 0 else 0,
                max_col_len=int(cdeg.max()) if n > 0 and e > 0 else 0)
  return out, counts


_NATIVE_KEYS = ('rowptr', 'colidx', 'perm', 'rowidx', 'cscptr', 'cscpos', 'bin_rows')


def build_arrays_native(edge_index, n):
  """The same arrays and counts as build_arrays_on_device by the library's own device builder (csrc/graph_device.hip: two calls around
  ONE host read of the counts, instead of ~60 torch launches and half a dozen reads).  Element for element equal
  (tests/test_kernels_gpu.py::test_native_graph_builder_equals_the_other_two)."""
  import ctypes
  L = _lib.lib()
  dev = edge_index.device
  e = int(edge_index.shape[1])
  ei = edge_index if (edge_index.dtype == torch.int64 and edge_index.is_contiguous()) else edge_index.to(torch.int64).contiguous()
  i32 = dict(dtype=torch.int32, device=dev)
  out = {'rowptr': torch.empty(n + 1, **i32), 'cscptr': torch.empty(n + 1, **i32), 'bin_rows': torch.empty(4 * max(n, 1), **i32)}
  for k in ('colidx', 'perm', 'rowidx', 'cscpos'):
    out[k] = torch.zeros(max(e, 1), **i32) if e == 0 else torch.empty(e, **i32)
  counts = torch.empty(16, **i32)
  ws = torch.empty(max(int(L.gnpde_graph_build_device_workspace_bytes(e, n)), 256), dtype=torch.uint8, device=dev)
  stream = _lib.stream_of(ei)
  _lib.check(L.gnpde_graph_build_device(_lib.ptr(ei[0]) if e else None, _lib.ptr(ei[1]) if e else None, e, n, _lib.ptr(out['rowptr']), _lib.ptr(out['colidx']),
                                        _lib.ptr(out['perm']), _lib.ptr(out['rowidx']), _lib.ptr(out['cscptr']), _lib.ptr(out['cscpos']),
                                        _lib.ptr(out['bin_rows']), _lib.ptr(counts), _lib.ptr(ws), ws.numel(), stream))
  c = counts.tolist()                       # the one host read
  if c[0]:
    raise _lib.GnpdeError('libgnpde_hip error -1: graph_build: edge index outside [0,%d)' % n)
  n16, n64, nle, nlr, nlc, nch, max_row, max_col = c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]
  for k, size in (('long_rows', nlr), ('long_chunk_row', nch), ('long_chunk_begin', nch), ('long_chunk_end', nch), ('long_cols', nlc),
                  ('long_chunk_first', nch)):
    out[k] = torch.zeros(max(size, 1), **i32)
  out['long_chunk_ptr'] = torch.zeros(nlr + 1, **i32)
  if nlr or nlc:
    _lib.check(L.gnpde_graph_build_device_long(_lib.ptr(out['rowptr']), _lib.ptr(out['cscptr']), n, nlr, nch, nlc, _lib.ptr(out['long_rows']),
                                               _lib.ptr(out['long_chunk_ptr']), _lib.ptr(out['long_chunk_row']), _lib.ptr(out['long_chunk_begin']),
                                               _lib.ptr(out['long_chunk_end']), _lib.ptr(out['long_cols']), _lib.ptr(out['long_chunk_first']),
                                               _lib.ptr(counts), _lib.ptr(ws), ws.numel(), stream))
  counts_d = dict(n_long_rows=nlr, n_long_chunks=nch, n_long_cols=nlc, n_bin16=n16, n_bin64=n64, n_bin_le64=nle, max_row_len=max_row, max_col_len=max_col)
  return out, counts_d


class CSRGraph(object):
  """Device-resident CSR view of `edge_index` for a square N x N operator.

  perm[p] = index (into the caller's edge list) of the entry stored at CSR position p; the sort is
  stable, so duplicates and the caller's within-row order are preserved."""

  def __init__(self, edge_index, num_nodes, device=None):
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
      raise ValueError('edge_index must be [2, E]')
    device = edge_index.device if device is None else torch.device(device)
    self.device = device
    self.n = int(num_nodes)
    self.e = int(edge_index.size(1))
    self._ws = {}
    self._transposed = None
    self._locality = {}
    if edge_index.is_cuda and edge_index.device == device:
      self._init_on_device(edge_index.detach())
      return
    L = _lib.lib()
    ei = edge_index.detach().to('cpu', torch.int64).contiguous()
    row = ei[0].contiguous().numpy()
    col = ei[1].contiguous().numpy()
    nlr, nlc, nlcol = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    _lib.check(L.gnpde_graph_count_long(row.ctypes.data, col.ctypes.data, self.e, self.n, ctypes.byref(nlr),
                                        ctypes.byref(nlc), ctypes.byref(nlcol)))
    self.n_long_rows, self.n_long_chunks, self.n_long_cols = nlr.value, nlc.value, nlcol.value
    host = {
      'rowptr': np.zeros(self.n + 1, np.int32), 'colidx': np.zeros(max(self.e, 1), np.int32),
      'perm': np.zeros(max(self.e, 1), np.int32), 'rowidx': np.zeros(max(self.e, 1), np.int32),
      'cscptr': np.zeros(self.n + 1, np.int32), 'cscpos': np.zeros(max(self.e, 1), np.int32),
      'long_rows': np.zeros(max(self.n_long_rows, 1), np.int32),
      'long_chunk_ptr': np.zeros(self.n_long_rows + 1, np.int32),
      'long_chunk_row': np.zeros(max(self.n_long_chunks, 1), np.int32),
      'long_chunk_begin': np.zeros(max(self.n_long_chunks, 1), np.int32),
      'long_chunk_end': np.zeros(max(self.n_long_chunks, 1), np.int32),
      'long_cols': np.zeros(max(self.n_long_cols, 1), np.int32),
      'bin_rows': np.zeros(4 * max(self.n, 1), np.int32),
      'long_chunk_first': np.zeros(max(self.n_long_chunks, 1), np.int32),
    }
    bin_counts = np.zeros(2, np.int32)
    order = ['rowptr', 'colidx', 'perm', 'rowidx', 'cscptr', 'cscpos', 'long_rows', 'long_chunk_ptr',
             'long_chunk_row', 'long_chunk_begin', 'long_chunk_end', 'long_cols', 'bin_rows']
    _lib.check(L.gnpde_graph_build(row.ctypes.data, col.ctypes.data, self.e, self.n,
                                   *([host[k].ctypes.data for k in order] + [bin_counts.ctypes.data,
                                                                              host['long_chunk_first'].ctypes.data])))
    self.n_bin16, self.n_bin64 = int(bin_counts[0]), int(bin_counts[1])
    self.t = {k: torch.from_numpy(v).to(device) for k, v in host.items()}
    self.perm_long = self.t['perm'][:self.e].long()
    s = _lib.GraphStruct()
    s.n, s.e = self.n, self.e
    s.n_long_rows, s.n_long_chunks = self.n_long_rows, self.n_long_chunks
    s.n_long_cols, s.n_bin16, s.n_bin64 = self.n_long_cols, self.n_bin16, self.n_bin64
    rp, cp = host['rowptr'], host['cscptr']
    self.max_row_len = int((rp[1:] - rp[:-1]).max()) if self.n > 0 else 0
    _lens = rp[1:] - rp[:-1]
    self.n_bin_le64 = int(((_lens > 16) & (_lens <= 64)).sum()) if self.n > 0 else 0
    s.n_bin_le64 = self.n_bin_le64
    self.max_col_len = int((cp[1:] - cp[:-1]).max()) if self.n > 0 else 0
    s.max_row_len, s.max_col_len = self.max_row_len, self.max_col_len
    for k in order + ['long_chunk_first']:
      setattr(s, k, self.t[k].data_ptr())
    self.struct = s
    self._edge_index = ei
    self.set_row_range(0, self.n, rowptr=torch.from_numpy(rp))

  def _init_on_device(self, edge_index):
    # (GNPDE_GRAPH_BUILD=torch: the torch-op builder, kept as the second opinion of the tests)
    native = os.environ.get('GNPDE_GRAPH_BUILD', 'native') != 'torch' and self.e < 2 ** 31
    self.t, c = (build_arrays_native if native else build_arrays_on_device)(edge_index, self.n)
    self.n_long_rows, self.n_long_chunks, self.n_long_cols = c['n_long_rows'], c['n_long_chunks'], c['n_long_cols']
    self.n_bin16, self.n_bin64 = c['n_bin16'], c['n_bin64']
    self.n_bin_le64 = c['n_bin_le64']
    self.max_row_len, self.max_col_len = c['max_row_len'], c['max_col_len']
    self.perm_long = self.t['perm'][:self.e].long()
    s = _lib.GraphStruct()
    s.n, s.e = self.n, self.e
    s.n_long_rows, s.n_long_chunks = self.n_long_rows, self.n_long_chunks
    s.n_long_cols, s.n_bin16, s.n_bin64 = self.n_long_cols, self.n_bin16, self.n_bin64
    s.max_row_len, s.max_col_len = self.max_row_len, self.max_col_len
    s.n_bin_le64 = self.n_bin_le64
    for k, v in self.t.items():
      setattr(s, k, v.data_ptr())
    self.struct = s
    self._edge_index = edge_index
    self.set_row_range(0, self.n)

  def set_row_range(self, row_begin, n_rows, rowptr=None):
    """The aggregation launches of this view cover the rows [row_begin, n_rows) (everything for an ordinary graph; the
    interior / boundary passes of a row-partitioned graph narrow it, distributed.py).  Also decides how those launches deal
    the rows to the 8 XCDs (gnpde_graph_t.xcd_deal): contiguous eighths while that is balanced -- neighbouring rows then
    share an XCD's L2, and it measured 1.8 % faster than hashed blocks at the ogbn-arxiv shape, where both are balanced --
    hashed blocks as soon as the row length depends on the row id (R-MAT: 1.46 by this measure)."""
    row_begin, n_rows = int(row_begin), int(n_rows)
    if not 0 <= row_begin <= n_rows <= self.t['rowptr'].numel() - 1:
      raise _lib.GnpdeError('row range [%d, %d) outside the graph\'s %d rows' % (row_begin, n_rows, self.t['rowptr'].numel() - 1))
    self.n = n_rows
    self.struct.n = n_rows
    self.struct.row_begin = row_begin
    self.xcd_imbalance_contiguous = contiguous_deal_imbalance(self.t['rowptr'] if rowptr is None else rowptr, row_begin, n_rows)
    self.struct.xcd_deal = _lib.XCD_HASHED if self.xcd_imbalance_contiguous > XCD_IMBALANCE_LIMIT else _lib.XCD_CONTIGUOUS

  @property
  def rowptr(self):
    return self.t['rowptr']

  @property
  def colidx(self):
    return self.t['colidx'][:self.e]

  @property
  def perm(self):
    return self.t['perm'][:self.e]

  def ref(self):
    return ctypes.byref(self.struct)

  def transposed(self):
    """CSR of the transposed operator over the SAME edge list (edge e <-> same id), used by the backward
    pass: A^T g is an aggregation over the flipped edges."""
    if self._transposed is None:
      self._transposed = CSRGraph(self._edge_index.flip(0), self.n, self.device)
    return self._transposed

  def transposed_positions(self):
    """(graph_t, t_from_csr): CSR of the transposed operator over the same edge list, and for every position of graph_t the CSR
    position of the same entry in this graph (both are stable sorts of one edge list: composed through the edge ids)."""
    hit = self.__dict__.get('_t_from_csr')
    if hit is None:
      gt = self.transposed()
      if self.e > 0:
        inv = torch.empty(self.e, dtype=torch.int64, device=self.device)
        inv[self.perm_long] = torch.arange(self.e, dtype=torch.int64, device=self.device)
        t_from_csr = inv[gt.perm_long].to(torch.int32).contiguous()
      else:
        t_from_csr = torch.zeros(1, dtype=torch.int32, device=self.device)
      hit = self.__dict__['_t_from_csr'] = (gt, t_from_csr)
    return hit

  def locality_view(self, row_bytes, mode='auto'):
    """LocalityView of this graph (the same operator with its nodes relabelled), or None when it does not apply / does not
    pay.  row_bytes: bytes of one row of the table the aggregation gathers from.  Two orders are candidates:
      'parts'  -- part by part (native label-propagation partitioner, parts of about LOCALITY_PART_BYTES of table): the rows
                  the XCDs work on at the same time reference each other, so their neighbours are lines the L2s already
                  hold.  Pays on graphs with communities (ogbn-arxiv stand-in: aggregation 171 -> 156 us).
      'degree' -- rows by descending length: the most-referenced rows of a power-law graph end up next to each other, and ids
                  whose BITS carry the degree (R-MAT: node 2^k is a hub for every k, so the hot rows sit at power-of-two
                  strides and land on the same memory channels) lose that structure -- R-MAT 2^21, d = 256: 14.7 -> 9.4 ms,
                  a random permutation alone gives 10.3 (profiles/r03_reorder_probe_rmat.txt).
    mode '0': never.  'parts' / 'degree': that order.  '1': the faster of the two by the clock ('parts' without a device).
    'auto' (default): only when the table does not fit the L2s, and then BY THE CLOCK -- a plain aggregation of this width is
    timed on the graph as given and on both candidates, once per graph and width, and the fastest candidate is kept if it is at
    least LOCALITY_MIN_GAIN faster than the graph as given."""
    mode = str(mode).lower()
    if mode in ('0', 'false', 'off', 'none') or self.struct.row_begin != 0 or self.n != self.t['rowptr'].numel() - 1 or self.e == 0:
      return None
    if mode in ('parts', 'degree'):
      return self._locality_candidate(mode, row_bytes)
    force = mode in ('1', 'true', 'on', 'force')
    table = float(self.n) * float(row_bytes)
    if self.device.type != 'cuda':
      return self._locality_candidate('parts', row_bytes) if force else None
    if not force and (table <= L2_BYTES_TOTAL or self.n < 4096):
      return None
    d = max(int(row_bytes) // 4, 1)
    dec = self._locality.setdefault('decision', {})
    if d not in dec:
      # one decision per (nodes, entries, width): a graph object rebuilt for the same edge set (cache eviction, a block that
      # swaps the same edges back in) reuses it instead of timing again; GNPDE_REORDER / opt['gnpde_reorder'] override it
      memo = _LOCALITY_DECISIONS.get(self._locality_memo_key(d))
      if memo is not None:
        dec[d] = memo
    if d not in dec:
      gains = {}
      measured = True
      try:
        t_base = self._aggregation_time(d)
        for kind in ('parts', 'degree'):
          view = self._locality_candidate(kind, row_bytes)
          gains[kind] = t_base / max(view.graph._aggregation_time(d), 1e-9)
          view.graph._ws.pop('spmm%d' % d, None)
      except torch.cuda.OutOfMemoryError:
        # the probe's operands and the candidates' CSRs are transient; when they do not fit next to the live solver buffers
        # the solve simply runs on the graph as given
        self._locality.get('views', {}).clear()
        torch.cuda.empty_cache()
        gains = {'parts': 0.0, 'degree': 0.0}
        measured = False          # a transient shortage: decided for THIS graph object only, never memoised for the shape
      best = max(gains, key=lambda k: gains[k])
      for kind in gains:                                   # the loser's CSR is state-sized at R-MAT scale: drop it
        if kind != best:
          self._locality.get('views', {}).pop(self._locality_key(kind, row_bytes), None)
      dec[d] = (best, gains)
      if measured:
        while len(_LOCALITY_DECISIONS) >= _LOCALITY_DECISIONS_MAX:      # bounded: oldest decision out (dicts keep insertion order)
          _LOCALITY_DECISIONS.pop(next(iter(_LOCALITY_DECISIONS)))
        _LOCALITY_DECISIONS[self._locality_memo_key(d)] = dec[d]
    best, gains = dec[d]
    if not force and gains[best] < LOCALITY_MIN_GAIN:
      return None
    view = self._locality_candidate(best, row_bytes)
    view.stats.setdefault('aggregation_speedup_measured', {})[str(d)] = {k: round(v, 4) for k, v in gains.items()}
    return view

  def _locality_memo_key(self, d):
    """Key of the process-wide decision memo: shape AND a fingerprint of the edge set (a weighted checksum of the row pointer and
    of the column ids: another graph with the same counts does not inherit this one's decision)."""
    fp = self._locality.get('fingerprint')
    if fp is None:
      rp = self.t['rowptr'].to(torch.int64)
      ci = self.t['colidx'][:self.e].to(torch.int64)
      wr = torch.arange(1, rp.numel() + 1, device=rp.device, dtype=torch.int64)
      wc = torch.arange(1, ci.numel() + 1, device=ci.device, dtype=torch.int64)
      fp = self._locality['fingerprint'] = (int((rp * wr).sum().item()) & 0xFFFFFFFFFFFF, int((ci * wc).sum().item()) & 0xFFFFFFFFFFFF)
    return (self.n, self.e, d, self.device.index, fp)

  def _locality_key(self, kind, row_bytes):
    if kind == 'degree':
      return ('degree',)
    table = float(self.n) * float(row_bytes)
    return ('parts', 8 * int(max(1, min(8, round(table / (8 * LOCALITY_PART_BYTES))))))

  def _locality_candidate(self, kind, row_bytes):
    views = self._locality.setdefault('views', {})
    key = self._locality_key(kind, row_bytes)
    view = views.get(key)
    if view is None:
      import time
      t0 = time.perf_counter()
      rowptr = self.t['rowptr'].cpu()
      lens = (rowptr[1:] - rowptr[:-1]).long()
      if kind == 'degree':
        order = torch.sort(lens, descending=True, stable=True).indices
        stats = {'order': 'rows by descending length'}
      else:
        n_parts = key[1]
        colidx = self.t['colidx'][:self.e].cpu()
        part = partition_rows((rowptr, colidx), n_parts, refine_links=0)
        rows = torch.repeat_interleave(torch.arange(self.n), lens)
        inside = float((part[rows] == part[colidx.long()]).double().mean())
        order = torch.sort(part.long(), stable=True).indices     # new position -> old id; the old order is kept inside a part
        stats = {'order': 'part by part', 'n_parts': n_parts, 'entries_inside_a_part': round(inside, 4)}
      view = views[key] = LocalityView(self, order, stats)
      view.stats['preparation_seconds'] = round(time.perf_counter() - t0, 3)
    return view

  def _aggregation_time(self, d, reps=5):
    """Seconds per plain aggregation A u of width d on this graph (HIP events).  The operands are filled by a PRIVATE generator
    -- the probe must not advance the global device RNG stream that dropout draws from, or a run with the probe and a run
    without it would train differently -- and are freed before this returns."""
    from . import ops
    gen = torch.Generator(device=self.device).manual_seed(20240521)
    d = (int(d) + 3) // 4 * 4       # the solvers pad such rows to 16-byte lanes (alloc_state): time the kernel they will run
    u = torch.empty(self.n, d, device=self.device).normal_(generator=gen)
    out = torch.empty_like(u)
    w = torch.empty(max(self.e, 1), device=self.device).uniform_(generator=gen)
    for _ in range(2):
      ops.spmm(self, w, u, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = None
    for _ in range(3):        # the BEST of three timings: one disturbed measurement must not decide the node order of every later solve
      e0.record()             # (seen once in round 5: a run of the headline bench without relabelling, 1000 instead of 1060 steps/s)
      for _ in range(reps):
        ops.spmm(self, w, u, out=out)
      e1.record()
      torch.cuda.synchronize(self.device)
      ti = e0.elapsed_time(e1) * 1e-3 / reps
      t = ti if t is None or ti < t else t
    del u, out, w
    return t

  def workspace(self, tag, nbytes):
    """Persistent scratch keyed by use (stable addresses keep captured hipGraphs valid)."""
    nbytes = max(int(nbytes), 256)
    buf = self._ws.get(tag)
    if buf is None or buf.numel() < nbytes:
      buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
      self._ws[tag] = buf
    return buf


class LocalityView(object):
  """The same operator with its nodes relabelled in a given order (CSRGraph.locality_view chooses it: part by part, or by
  descending row length).  Part by part at the ogbn-arxiv stand-in (40 shuffled communities, 65 % of the edges inside one):
  aggregation 171 -> 156 us per launch with 16 parts, as much as relabelling by the generator's own communities gives
  (profiles/r03_reorder_probe.txt).

  graph: CSRGraph over the relabelled edge list IN THE CALLER'S EDGE ORDER, so `perm`, every per-edge array and -- because the
  entries of a row keep their order -- every row sum are the same as on the original graph: results are bit-identical up to the
  row permutation.  order[i] = original id of the node at position i; inv[v] = position of original node v.  A state x in the
  original order enters as x[order] and leaves as y[inv].  No reference equivalent (torch_sparse.spmm takes the ids as
  given, reference src/function_transformer_attention.py:35)."""

  def __init__(self, base, order, stats):
    dev = base.device
    order = order.to(torch.int64)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), dtype=torch.int64)
    self.order, self.inv = order.to(dev), inv.to(dev)
    # (the relabelled CSR picks its rows -> XCDs deal like any graph: parts differ in density, so contiguous eighths by row
    #  count are usually out of balance and the hashed blocks take over -- all eight XCDs then work through the same part at the
    #  same time.  Work-balanced contiguous ranges, one part per XCD at a time, measured SLOWER: 172.6 vs 157.2 us,
    #  profiles/r03_reorder_probe.txt -- the parts differ in how well they cache, and a launch lasts as long as its slowest XCD)
    ei = base._edge_index.to(dev)
    self.graph = CSRGraph(torch.stack([self.inv[ei[0]], self.inv[ei[1]]]), base.n, device=dev)
    del ei
    self.stats = dict(stats, xcd_imbalance_contiguous=round(float(self.graph.xcd_imbalance_contiguous), 4),
                      xcd_deal='hashed_blocks' if self.graph.struct.xcd_deal == _lib.XCD_HASHED else 'contiguous_eighths')

  def enter(self, x, out=None):
    """x[order] (rows in the relabelled order); written straight into `out` when that is a dense [n, d] buffer."""
    if out is None:
      return x.index_select(0, self.order)
    if out.is_contiguous() and out.dtype == x.dtype and out.shape == x.shape:
      return torch.index_select(x, 0, self.order, out=out)
    out.copy_(x.index_select(0, self.order))
    return out

  def leave(self, y, out=None):
    """y[inv] (rows back in the caller's order)."""
    if out is None:
      return y.index_select(0, self.inv)
    if out.is_contiguous() and out.dtype == y.dtype and out.shape == y.shape:
      return torch.index_select(y, 0, self.inv, out=out)
    out.copy_(y.index_select(0, self.inv))
    return out


class _GraphCache(object):
  """The reference assigns the SAME edge_index tensor to several ODEFunc instances
  (src/block_constant.py:22-24) and some blocks replace it between forwards
  (src/block_transformer_hard_attention.py, run_GNN.py:254): graphs are rebuilt lazily, keyed on
  tensor identity + version, and shared."""

  def __init__(self, capacity=8):
    self.capacity = capacity
    self.entries = []  # (edge_index tensor kept alive, version, n, device, graph)

  def get(self, edge_index, num_nodes, device):
    device = torch.device(device)
    ver = edge_index._version
    for i, (t, v, n, dev, g) in enumerate(self.entries):
      if t is edge_index and v == ver and n == num_nodes and dev == device:
        if i:
          self.entries.insert(0, self.entries.pop(i))
        return g
    g = CSRGraph(edge_index, num_nodes, device)
    self.entries.insert(0, (edge_index, ver, num_nodes, device, g))
    del self.entries[self.capacity:]
    return g

  def clear(self):
    self.entries = []


GRAPHS = _GraphCache()


def graph_of(edge_index, num_nodes, device=None):
  return GRAPHS.get(edge_index, int(num_nodes), edge_index.device if device is None else device)


def partition_rows(graph_or_csr, n_parts, refine_iters=8, seed=0, row_weight=1, cluster_div=4, refine_links=6, stats=None):
  """Balanced k-way row partition (native, csrc/graph_prep.cpp).  Accepts a CSRGraph or a
  (rowptr, colidx) pair of int32 CPU tensors; returns an int32 CPU tensor [n].  The parts are balanced on
  entries + row_weight per row: 1 follows the aggregation time, larger values even out the NODE counts (and with them the
  rows a part has to send to its peers); the label-propagation clusters that are packed into parts are capped at a part's
  work / cluster_div.  refine_links > 0: that many passes per phase of the communication refinement afterwards (rows received
  per rank and the busiest link instead of cut edges; `stats` dict receives before / after)."""
  if isinstance(graph_or_csr, CSRGraph):
    rowptr = graph_or_csr.t['rowptr'].cpu()
    colidx = graph_or_csr.t['colidx'].cpu()
  else:
    rowptr, colidx = graph_or_csr
  rowptr = rowptr.to(torch.int32).contiguous()
  colidx = colidx.to(torch.int32).contiguous()
  n = rowptr.numel() - 1
  part = torch.zeros(n, dtype=torch.int32)
  _lib.check(_lib.lib().gnpde_partition_rows_ex(rowptr.data_ptr(), colidx.data_ptr(), n, int(n_parts), int(refine_iters),
                                                int(seed), int(row_weight), int(cluster_div), part.data_ptr()))
  if refine_links and n_parts > 1 and n * int(n_parts) <= 2 ** 31:
    # communication refinement (gnpde_partition_refine_links): fewer received rows, lower busiest link, same balance window
    st = torch.zeros(5, dtype=torch.int64)
    _lib.check(_lib.lib().gnpde_partition_refine_links(rowptr.data_ptr(), colidx.data_ptr(), n, int(n_parts), int(row_weight),
                                                       int(refine_links), part.data_ptr(), st.data_ptr()))
    if stats is not None:
      stats.update(max_link_before=int(st[0]), max_link_after=int(st[1]), received_rows_before=int(st[2]),
                   received_rows_after=int(st[3]), moves=int(st[4]))
  return part
