"""CPU restatement (torch, fp32, op-for-op) of the reference's ODE right-hand-side path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this file; the product (graph-neural-pde_amd/) never does and fails loudly without its HIP
library.  Every function cites the reference lines it follows (paths relative to
/root/reference).  Third-party arithmetic that is not in the reference tree is restated from the
pinned packages' published behaviour and marked [3P] (torch-sparse 0.6.9, torch-scatter 2.0.6,
torch-geometric 1.7.0, torchdiffeq 0.2.1; reference README.md:23-29).

Pinning: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, which
oracle/gen_golden.py produced by running the reference's own src/*.py (imported from
/root/reference over oracle/shims) on seeded inputs, and against the known-answer facts of the
reference's unit tests (exact-0.5 attention, rows summing to one, head-mean linearity,
rw-normalisation identities).  dopri5 [3P] has no reference-side golden vector: parity unpinned for
the adaptive solver beyond agreement with the shim.

The op order (index_select -> mul -> scatter_add_, scatter amax -> exp -> scatter_add -> divide by
sum + 1e-16) is what the reference executes on CPU, so timing this file is the "reference CPU
torch-sparse path" baseline (cpu_baseline.kind = "port").
"""
import math
import torch


# ------------------------------------------------------------------------------------------------
# [3P] segment primitives
# ------------------------------------------------------------------------------------------------
def scatter_sum(src, index, n):
  """torch_scatter.scatter_add(src, index, dim=0, dim_size=n) [3P]."""
  out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
  idx = index if src.dim() == 1 else index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
  return out.scatter_add_(0, idx, src)


def scatter_max(src, index, n):
  """torch_scatter.scatter(src, index, dim=0, dim_size=n, reduce='max') [3P] (empty segments -> 0)."""
  out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
  idx = index if src.dim() == 1 else index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
  return out.scatter_reduce_(0, idx, src, 'amax', include_self=False)


def spmm(index, value, m, matrix):
  """torch_sparse.spmm [3P]: out[row_e] += value_e * matrix[col_e]
  (call sites: function_transformer_attention.py:35, function_laplacian_diffusion.py:31-35,
  function_GAT_attention.py:35,41)."""
  row, col = index[0], index[1]
  out = matrix.index_select(0, col) * value.unsqueeze(-1)
  return scatter_sum(out, row, m)


def segment_softmax(src, index, n):
  """torch_geometric.utils.softmax 1.7.0 [3P] (function_transformer_attention.py:213,
  function_GAT_attention.py:114)."""
  out = src - scatter_max(src, index, n)[index]
  out = out.exp()
  out_sum = scatter_sum(out, index, n)[index]
  return out / (out_sum + 1e-16)


def squareplus(src, index, n):
  """utils.py:179-208 -- note the GLOBAL max at :196."""
  out = src - src.max()
  out = (out + torch.sqrt(out ** 2 + 4)) / 2
  out_sum = scatter_sum(out, index, n)[index]
  return out / (out_sum + 1e-16)


# ------------------------------------------------------------------------------------------------
# graph preparation
# ------------------------------------------------------------------------------------------------
def add_remaining_self_loops(edge_index, edge_weight, fill_value, n):
  """torch_geometric.utils.add_remaining_self_loops 1.7.0 [3P]; used at
  function_transformer_attention.py:17-19, utils.py:63,113."""
  row, col = edge_index[0], edge_index[1]
  mask = row != col
  loops = torch.arange(n, dtype=row.dtype).unsqueeze(0).repeat(2, 1)
  new_index = torch.cat([edge_index[:, mask], loops], dim=1)
  if edge_weight is not None:
    loop_w = torch.full((n,), fill_value, dtype=edge_weight.dtype)
    keep = edge_weight[~mask]
    if keep.numel() > 0:
      loop_w[row[~mask]] = keep
    edge_weight = torch.cat([edge_weight[mask], loop_w], dim=0)
  return new_index, edge_weight


def get_rw_adj(edge_index, edge_weight=None, norm_dim=1, fill_value=0., num_nodes=None, dtype=torch.float32):
  """utils.py:105-123."""
  n = num_nodes
  if edge_weight is None:
    edge_weight = torch.ones(edge_index.size(1), dtype=dtype)
  if not fill_value == 0:
    edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, n)
  row, col = edge_index[0], edge_index[1]
  indices = row if norm_dim == 0 else col
  deg = scatter_sum(edge_weight, indices, n)
  deg_inv = deg.pow_(-1)
  edge_weight = deg_inv[indices] * edge_weight if norm_dim == 0 else edge_weight * deg_inv[indices]
  return edge_index, edge_weight


def gcn_norm_fill_val(edge_index, edge_weight=None, fill_value=0., num_nodes=None, dtype=torch.float32):
  """utils.py:55-72."""
  n = num_nodes
  if edge_weight is None:
    edge_weight = torch.ones(edge_index.size(1), dtype=dtype)
  if not int(fill_value) == 0:
    edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, n)
  row, col = edge_index[0], edge_index[1]
  deg = scatter_sum(edge_weight, col, n)
  dis = deg.pow_(-0.5)
  dis.masked_fill_(dis == float('inf'), 0)
  return edge_index, dis[row] * edge_weight * dis[col]


# ------------------------------------------------------------------------------------------------
# attention layers
# ------------------------------------------------------------------------------------------------
def transformer_attention(x, edge, Wq, bq, Wk, bk, heads, attention_type='scaled_dot', norm_idx=0,
                          square_plus=False, edge_weights=None, reweight=False,
                          output_var=None, lengthscale=None):
  """SpGraphTransAttentionLayer.forward, non-beltrami branch
  (function_transformer_attention.py:173-214).  Returns (attention [E,h], prods [E,h])."""
  n = x.shape[0]
  q = torch.nn.functional.linear(x, Wq, bq)
  k = torch.nn.functional.linear(x, Wk, bk)
  d_k = Wq.shape[0] // heads
  q = q.view(-1, heads, d_k).transpose(1, 2)    # [N, d_k, h]   (:180-188)
  k = k.view(-1, heads, d_k).transpose(1, 2)
  src = q[edge[0, :], :, :]                     # (:190)
  dst_k = k[edge[1, :], :, :]                   # (:191)
  if attention_type == 'exp_kernel':            # (:193-194)
    prods = output_var ** 2 * torch.exp(-(torch.sum((src - dst_k) ** 2, dim=1) / (2 * lengthscale ** 2)))
  elif attention_type == 'scaled_dot':          # (:195-196)
    prods = torch.sum(src * dst_k, dim=1) / math.sqrt(d_k)
  elif attention_type == 'cosine_sim':          # (:197-199)
    prods = torch.nn.functional.cosine_similarity(src, dst_k, dim=1, eps=1e-5)
  elif attention_type == 'pearson':             # (:200-206)
    src = src - torch.mean(src, dim=1, keepdim=True)
    dst_k = dst_k - torch.mean(dst_k, dim=1, keepdim=True)
    prods = torch.nn.functional.cosine_similarity(src, dst_k, dim=1, eps=1e-5)
  else:
    raise ValueError(attention_type)
  if reweight and edge_weights is not None:     # (:208-209)
    prods = prods * edge_weights.unsqueeze(dim=1)
  if square_plus:                               # (:210-213)
    attention = squareplus(prods, edge[norm_idx], n)
  else:
    attention = segment_softmax(prods, edge[norm_idx], n)
  return attention, prods


def transformer_attention_split(x, edge, P, heads, feat_dim, pos_dim, norm_idx=0, square_plus=False, edge_weights=None,
                                reweight=False):
  """SpGraphTransAttentionLayer.forward, beltrami + exp_kernel branch (function_transformer_attention.py:133-171):
  one exp kernel on the feature (+ label) columns, one on the positional columns, multiplied.  P maps the layer's
  parameter names (Qx.weight, ..., lengthscale_p) to tensors.  Returns (attention [E,h], prods [E,h])."""
  n = x.shape[0]
  label_index = feat_dim + pos_dim
  p = x[:, feat_dim:label_index]
  xf = torch.cat((x[:, :feat_dim], x[:, label_index:]), dim=1)
  d_k = P['Qx.weight'].shape[0] // heads

  def split_heads(w, b, inp):
    return torch.nn.functional.linear(inp, w, b).view(-1, heads, d_k).transpose(1, 2)
  src_x = split_heads(P['Qx.weight'], P['Qx.bias'], xf)[edge[0, :], :, :]
  dst_x = split_heads(P['Kx.weight'], P['Kx.bias'], xf)[edge[1, :], :, :]
  src_p = split_heads(P['Qp.weight'], P['Qp.bias'], p)[edge[0, :], :, :]
  dst_p = split_heads(P['Kp.weight'], P['Kp.bias'], p)[edge[1, :], :, :]
  prods = P['output_var_x'] ** 2 * torch.exp(-torch.sum((src_x - dst_x) ** 2, dim=1) / (2 * P['lengthscale_x'] ** 2)) \
      * P['output_var_p'] ** 2 * torch.exp(-torch.sum((src_p - dst_p) ** 2, dim=1) / (2 * P['lengthscale_p'] ** 2))
  if reweight and edge_weights is not None:
    prods = prods * edge_weights.unsqueeze(dim=1)
  if square_plus:
    attention = squareplus(prods, edge[norm_idx], n)
  else:
    attention = segment_softmax(prods, edge[norm_idx], n)
  return attention, prods


def gat_attention(x, edge, W, a, heads, leaky_slope=0.2, norm_idx=0):
  """SpGraphAttentionLayer.forward (function_GAT_attention.py:105-115).
  Returns (attention [E,h], wx [N,A])."""
  n = x.shape[0]
  wx = torch.mm(x, W)
  d_k = W.shape[1] // heads
  h = wx.view(-1, heads, d_k).transpose(1, 2)   # [N, d_k, heads]
  edge_h = torch.cat((h[edge[0, :], :, :], h[edge[1, :], :, :]), dim=1).transpose(0, 1)  # [2d_k, E, h]
  edge_e = torch.nn.functional.leaky_relu(torch.sum(a * edge_h, dim=0), leaky_slope)
  attention = segment_softmax(edge_e, edge[norm_idx], n)
  return attention, wx


# ------------------------------------------------------------------------------------------------
# right-hand sides f(t, x)
# ------------------------------------------------------------------------------------------------
def _epilogue(ax, x, alpha_train, beta_train, x0, no_alpha_sigmoid, add_source):
  """function_transformer_attention.py:46-53 == function_laplacian_diffusion.py:43-51 ==
  function_GAT_attention.py:56-64."""
  alpha = alpha_train if no_alpha_sigmoid else torch.sigmoid(alpha_train)
  f = alpha * (ax - x)
  if add_source:
    f = f + beta_train * x0
  return f


def rhs_laplacian(x, edge, weight, alpha_train, beta_train, x0=None, no_alpha_sigmoid=False,
                  add_source=False):
  """LaplacianODEFunc.forward (function_laplacian_diffusion.py:38-51); `weight` is [E], or [E,h]
  when the attention block hands per-head attention (:29-31, mean over heads)."""
  w = weight.mean(dim=1) if weight.dim() == 2 else weight
  ax = spmm(edge, w, x.shape[0], x)
  return _epilogue(ax, x, alpha_train, beta_train, x0, no_alpha_sigmoid, add_source)


def rhs_transformer(x, edge, Wq, bq, Wk, bk, heads, alpha_train, beta_train, x0=None,
                    no_alpha_sigmoid=False, add_source=False, **att_kw):
  """ODEFuncTransformerAtt.forward with mix_features=False (function_transformer_attention.py:38-53,
  :33-35)."""
  attention, _ = transformer_attention(x, edge, Wq, bq, Wk, bk, heads, **att_kw)
  ax = spmm(edge, attention.mean(dim=1), x.shape[0], x)
  return _epilogue(ax, x, alpha_train, beta_train, x0, no_alpha_sigmoid, add_source)


def rhs_from_attention(x, edge, attention, alpha_train, beta_train, x0=None, no_alpha_sigmoid=False, add_source=False):
  """ODEFuncTransformerAtt.forward after the attention layer (function_transformer_attention.py:33-35, :46-53)."""
  ax = spmm(edge, attention.mean(dim=1), x.shape[0], x)
  return _epilogue(ax, x, alpha_train, beta_train, x0, no_alpha_sigmoid, add_source)


def rhs_gat(x, edge, W, a, heads, alpha_train, beta_train, x0=None, no_alpha_sigmoid=False,
            add_source=False, leaky_slope=0.2, norm_idx=0, mix_features=False, Wout=None):
  """ODEFuncAtt.forward (function_GAT_attention.py:45-65, multiply_attention :31-43)."""
  attention, wx = gat_attention(x, edge, W, a, heads, leaky_slope, norm_idx)
  n = x.shape[0]
  if mix_features:
    wx = torch.mean(torch.stack([spmm(edge, attention[:, i], n, wx) for i in range(heads)], dim=0), dim=0)
    ax = torch.mm(wx, Wout)
  else:
    ax = torch.mean(torch.stack([spmm(edge, attention[:, i], n, x) for i in range(heads)], dim=0), dim=0)
  return _epilogue(ax, x, alpha_train, beta_train, x0, no_alpha_sigmoid, add_source)


# ------------------------------------------------------------------------------------------------
# [3P] torchdiffeq 0.2.1 fixed-grid integrators, as called from block_constant.py:57-62
# ------------------------------------------------------------------------------------------------
def two_hop(edge_index, weight, n):
  """S = coalesce(A ++ offdiag(A A)) / 2 of RewireAttODEblock.add_khop_edges (src/block_transformer_rewiring.py:68-86:
  [3P] torch_sparse.spspmm(A, A, coalesced=True) -> torch_geometric remove_self_loops -> cat with A -> / 2 ->
  [3P] torch_sparse.coalesce(op='add')), with dense float64 algebra; the structure comes from a product of the 0/1 patterns,
  so entries whose value happens to be zero stay, as they do in the reference.  Returns (index [2, nnz] ordered by
  (row, col), value float64)."""
  A = torch.zeros(n, n, dtype=torch.float64)
  A.index_put_((edge_index[0], edge_index[1]), weight.double(), accumulate=True)
  P = torch.zeros(n, n, dtype=torch.float64)
  P.index_put_((edge_index[0], edge_index[1]), torch.ones(edge_index.shape[1], dtype=torch.float64), accumulate=True)
  A2, P2 = A @ A, P @ P
  A2.fill_diagonal_(0)
  P2.fill_diagonal_(0)
  idx = ((P + P2) > 0).nonzero().t()
  return idx, ((A + A2) / 2)[idx[0], idx[1]]


def time_grid(T, step_size, dtype=torch.float32):
  """FixedGridODESolver._grid_constructor_from_step_size [3P]: niters = ceil(T/h + 1), last <- T."""
  t = torch.tensor([0, T], dtype=dtype)
  niters = torch.ceil((t[-1] - t[0]) / step_size + 1).item()
  grid = torch.arange(0, niters, dtype=dtype) * step_size + t[0]
  grid[-1] = t[-1]
  return grid


def odeint_fixed(func, y0, T, step_size=1.0, method='rk4'):
  """y(T) of dy/dt = func(t, y) with torchdiffeq's `euler`, `midpoint` or `rk4` (= 3/8-rule
  `rk4_alt_step_func`, confirmed by early_stop_solver.py:10,150-155)."""
  grid = time_grid(T, step_size, y0.dtype)
  y = y0
  third = 1 / 3
  for t0, t1 in zip(grid[:-1], grid[1:]):
    dt = t1 - t0
    if method == 'euler':
      dy = dt * func(t0, y)
    elif method == 'midpoint':     # torchdiffeq fixed_grid.py Midpoint._step_func [3P]
      half = 0.5 * dt
      dy = dt * func(t0 + half, y + func(t0, y) * half)
    elif method == 'rk4':
      k1 = func(t0, y)
      k2 = func(t0 + dt * third, y + dt * k1 * third)
      k3 = func(t0 + dt * 2 * third, y + dt * (k2 - k1 * third))
      k4 = func(t1, y + dt * (k1 - k2 + k3))
      dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
    else:
      raise ValueError(method)
    y = y + dy
  return y


# ------------------------------------------------------------------------------------------------
# early-stopping evaluation of the test-time integrator (early_stop_solver.py:131-225)
# ------------------------------------------------------------------------------------------------
def early_stop_accuracies(z, m2_weight, m2_bias, labels, masks):
  """EarlyStopRK4.evaluate + test (:178-218, :162-168): relu -> linear -> arg-max -> accuracy per mask.
  (The ogbn-arxiv branch inserts a log_softmax, which leaves the arg-max unchanged, and scores with
  ogb's Evaluator, the same ratio.)  The state is cut to the decoder's width when augmented (:180-181)."""
  if m2_weight.shape[1] != z.shape[1]:
    z = z[:, :m2_weight.shape[1]]
  logits = torch.nn.functional.linear(torch.relu(z), m2_weight, m2_bias)
  pred = logits.max(1)[1]
  labels = labels.reshape(-1)
  return [float(pred[m].eq(labels[m]).sum().item()) / float(m.sum().item()) for m in masks]


def odeint_rk4_early_stop(func, y0, T, step_size, m2_weight, m2_bias, labels, masks):
  """EarlyStopRK4.integrate (:163-186) over [0, T]: returns (y(T), best = [train, val, test, time] of the step
  with the strictly best validation accuracy (initial best_val = 0), list of [t1, train, val, test] per step)."""
  grid = time_grid(T, step_size, y0.dtype)
  y = y0
  third = 1 / 3
  best = [0.0, 0.0, 0.0, 0.0]
  steps = []
  for t0, t1 in zip(grid[:-1], grid[1:]):
    dt = t1 - t0
    k1 = func(t0, y)
    k2 = func(t0 + dt * third, y + dt * k1 * third)
    k3 = func(t0 + dt * 2 * third, y + dt * (k2 - k1 * third))
    k4 = func(t1, y + dt * (k1 - k2 + k3))
    y = y + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
    acc = early_stop_accuracies(y, m2_weight, m2_bias, labels, masks)
    if acc[1] > best[1]:
      best = acc + [float(t1)]
    steps.append([float(t1)] + acc)
  return y, best, steps


# ------------------------------------------------------------------------------------------------
# parity metric
# ------------------------------------------------------------------------------------------------
def parity_error(a, b):
  """(max|a-b| / max|b|, ||a-b||_2 / ||b||_2); the bar is 1e-5 on both."""
  a = a.detach().double().cpu()
  b = b.detach().double().cpu()
  den_inf = b.abs().max().clamp_min(1e-30)
  den_2 = b.norm().clamp_min(1e-30)
  return float((a - b).abs().max() / den_inf), float((a - b).norm() / den_2)
