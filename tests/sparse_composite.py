"""TEST INFRASTRUCTURE: the rewiring block's two-hop densification as a composite of plain torch ops (what
torch_sparse.spspmm / coalesce compute for the reference, src/block_transformer_rewiring.py:68-86).  The product has no such
path (csrc/twohop.hip is the only implementation); the kernel is tested against this expanded-list form."""
import torch


def _coalesce(index, value, n):
  """Sum duplicates, entries ordered by (row, col) -- torch_sparse.coalesce(op='add')."""
  key = index[0] * n + index[1]
  uniq, inverse = torch.unique(key, sorted=True, return_inverse=True)
  out = torch.zeros(uniq.numel(), dtype=value.dtype, device=value.device).index_add_(0, inverse, value)
  return torch.stack([torch.div(uniq, n, rounding_mode='floor'), uniq % n]), out


def _spspmm(index_a, value_a, index_b, value_b, n):
  """C = A B for COO operands, coalesced (torch_sparse.spspmm(..., coalesced=True)): every entry (i, k) of A is
  paired with row k of B through B's row pointer."""
  order = torch.argsort(index_b[0] * n + index_b[1])
  b_row, b_col, b_val = index_b[0][order], index_b[1][order], value_b[order]
  rowptr = torch.zeros(n + 1, dtype=torch.long, device=b_row.device)
  rowptr[1:] = torch.cumsum(torch.bincount(b_row, minlength=n), 0)
  counts = rowptr[index_a[1] + 1] - rowptr[index_a[1]]
  src = torch.repeat_interleave(torch.arange(index_a.shape[1], device=counts.device), counts)
  first = torch.cumsum(counts, 0) - counts
  pos = torch.arange(src.numel(), device=counts.device) - first[src] + rowptr[index_a[1]][src]
  return _coalesce(torch.stack([index_a[0][src], b_col[pos]]), value_a[src] * b_val[pos], n)


def two_hop(ei, ew, n):
  """(A + (A^2 without its diagonal)) / 2, coalesced."""
  new_edges, new_weights = _spspmm(ei, ew, ei, ew, n)
  keep = new_edges[0] != new_edges[1]
  return _coalesce(torch.cat([ei, new_edges[:, keep]], dim=1), torch.cat([ew, new_weights[keep]], dim=0) / 2, n)
