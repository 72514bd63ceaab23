"""Host-side graph normalisation used when a block is constructed (once per model, not on the hot
path) and the NFE guard exception.  Mirrors the four symbols of the reference's src/utils.py that
the ODE path touches: MaxNFEException (:18), gcn_norm_fill_val (:55-72), get_rw_adj (:105-123) and
the DummyDataset/DummyData holders (:236-246)."""
import torch


class MaxNFEException(Exception):
  """Raised by ODEFunc.forward when opt['max_nfe'] is exceeded (reference src/utils.py:18)."""
  pass


def _num_nodes(edge_index, num_nodes):
  if num_nodes is not None:
    return int(num_nodes)
  return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0


def _segment_sum(values, index, n):
  return torch.zeros(n, dtype=values.dtype, device=values.device).index_add_(0, index, values)


def add_remaining_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
  """PyG 1.7.0 semantics: existing self-loops are dropped from the list, N loops are APPENDED at
  the end, and an existing loop keeps its weight instead of `fill_value`."""
  n = _num_nodes(edge_index, num_nodes)
  row, col = edge_index[0], edge_index[1]
  off_diag = row != col
  loops = torch.arange(n, dtype=row.dtype, device=row.device)
  out_index = torch.cat([edge_index[:, off_diag], torch.stack([loops, loops])], dim=1)
  out_weight = None
  if edge_weight is not None:
    diag_w = torch.full((n,), fill_value, dtype=edge_weight.dtype, device=edge_weight.device)
    on_diag = ~off_diag
    if bool(on_diag.any()):
      diag_w[row[on_diag]] = edge_weight[on_diag]
    out_weight = torch.cat([edge_weight[off_diag], diag_w])
  return out_index, out_weight


def get_rw_adj(edge_index, edge_weight=None, norm_dim=1, fill_value=0., num_nodes=None, dtype=None):
  """Random-walk normalisation w_e / deg[index_e] with deg summed over `row` (norm_dim 0) or `col`
  (norm_dim 1, what the blocks use: column-stochastic weights aggregated into rows)."""
  n = _num_nodes(edge_index, num_nodes)
  if edge_weight is None:
    edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
  if fill_value != 0:
    edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, n)
  idx = edge_index[0] if norm_dim == 0 else edge_index[1]
  inv_deg = _segment_sum(edge_weight, idx, n).pow_(-1)
  return edge_index, (inv_deg[idx] * edge_weight if norm_dim == 0 else edge_weight * inv_deg[idx])


def gcn_norm_fill_val(edge_index, edge_weight=None, fill_value=0., num_nodes=None, dtype=None):
  """Symmetric normalisation deg^-1/2[row] w deg^-1/2[col] with deg summed over `col`; self-loops
  only when int(fill_value) != 0 (the reference's truncation is kept: 0.3 adds no loops)."""
  n = _num_nodes(edge_index, num_nodes)
  if edge_weight is None:
    edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
  if int(fill_value) != 0:
    edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, n)
  row, col = edge_index[0], edge_index[1]
  dis = _segment_sum(edge_weight, col, n).pow_(-0.5)
  dis.masked_fill_(dis == float('inf'), 0)
  return edge_index, dis[row] * edge_weight * dis[col]


class Meter(object):
  """Running tally of function evaluations: run_GNN.py's train() feeds `model.fm` / `model.bm` after every forward /
  backward pass and prints their `.sum` per epoch (reference src/utils.py:212-233, used at src/run_GNN.py:90-95)."""

  def __init__(self):
    self.reset()

  def reset(self):
    self.val = None
    self.sum = 0
    self.cnt = 0

  def update(self, val):
    self.val = val
    self.sum += val
    self.cnt += 1

  def get_average(self):
    return self.sum / self.cnt if self.cnt else 0

  def get_value(self):
    return self.val


class DummyDataset(object):
  def __init__(self, data, num_classes):
    self.data = data
    self.num_classes = num_classes


class DummyData(object):
  def __init__(self, edge_index=None, edge_Attr=None, num_nodes=None, x=None):
    self.edge_index = edge_index
    self.edge_attr = edge_Attr
    self.num_nodes = num_nodes
    self.x = x
