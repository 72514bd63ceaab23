"""GRAND-nl right-hand side: multi-head edge attention recomputed at every evaluation
(reference src/function_transformer_attention.py:12-56 ODEFuncTransformerAtt, :59-217
SpGraphTransAttentionLayer).  Per evaluation: one fp32-MFMA projection q||k, three attention
passes, one aggregation with the fused epilogue -- instead of ~30 PyTorch launches over [E,d_k,h]
and [E,d] temporaries."""
import torch
from torch import nn

from . import _lib, ops
from .base_classes import ODEFunc
from .graph import graph_of
from .utils import add_remaining_self_loops


class SpGraphTransAttentionLayer(nn.Module):
  """Same constructor, parameters (Q, K, V, Wout [+ output_var, lengthscale]) and return value
  `(attention [E,h] in edge order, (v, prods))` as the reference layer."""

  def __init__(self, in_features, out_features, opt, device, concat=True, edge_weights=None):
    super(SpGraphTransAttentionLayer, self).__init__()
    self.in_features = in_features
    self.out_features = out_features
    self.alpha = opt['leaky_relu_slope']
    self.concat = concat
    self.device = device
    self.opt = opt
    self.h = int(opt['heads'])
    self.edge_weights = edge_weights
    self.attention_dim = opt['attention_dim'] if 'attention_dim' in opt else out_features
    assert self.attention_dim % self.h == 0, "Number of heads ({}) must be a factor of the dimension size ({})".format(
      self.h, self.attention_dim)
    self.d_k = self.attention_dim // self.h
    if opt['attention_type'] not in _lib.ATT_TYPES:
      raise ValueError('unknown attention_type %r' % (opt['attention_type'],))
    # BLEND: separate feature / positional kernels, multiplied (reference :83-101, :133-171)
    self.split_kernel = bool(opt['beltrami'] and opt['attention_type'] == "exp_kernel")
    if self.split_kernel:
      self.output_var_x = nn.Parameter(torch.ones(1))
      self.lengthscale_x = nn.Parameter(torch.ones(1))
      self.output_var_p = nn.Parameter(torch.ones(1))
      self.lengthscale_p = nn.Parameter(torch.ones(1))
      fdim, pdim = opt['hidden_dim'] - opt['pos_enc_hidden_dim'], opt['pos_enc_hidden_dim']
      self.Qx, self.Vx, self.Kx = (nn.Linear(fdim, self.attention_dim) for _ in range(3))
      self.Qp, self.Vp, self.Kp = (nn.Linear(pdim, self.attention_dim) for _ in range(3))
      linears = (self.Qx, self.Vx, self.Kx, self.Qp, self.Vp, self.Kp)
    else:
      if opt['attention_type'] == "exp_kernel":
        self.output_var = nn.Parameter(torch.ones(1))
        self.lengthscale = nn.Parameter(torch.ones(1))
      self.Q = nn.Linear(in_features, self.attention_dim)
      self.V = nn.Linear(in_features, self.attention_dim)
      self.K = nn.Linear(in_features, self.attention_dim)
      linears = (self.Q, self.V, self.K)
    self.activation = nn.Sigmoid()
    self.Wout = nn.Linear(self.d_k, in_features)
    for m in linears + (self.Wout,):
      nn.init.constant_(m.weight, 1e-5)  # reference init (:122-126)
    self._bufs = {}

  @property
  def kernel_att_dim(self):
    """Width of q (and of k) as the device kernels see it: the split kernel runs as ONE exp_kernel over the
    concatenation [q_x / l_x ; q_p / l_p] per head (see qk_weights)."""
    return 2 * self.attention_dim if self.split_kernel else self.attention_dim

  def _grad_sources(self):
    if self.split_kernel:
      return (self.Qx.weight, self.Kx.weight, self.Qp.weight, self.Kp.weight)
    return (self.Q.weight, self.K.weight)

  # ---- native descriptor pieces ---------------------------------------------------------------
  def qk_weights(self):
    """[Q.weight; K.weight] ([2A, d]) and [Q.bias; K.bias], refreshed in place when a parameter's
    version changes so that captured solver graphs keep pointing at live data."""
    if self.split_kernel:
      return self._split_qk_weights()
    srcs = (self.Q.weight, self.K.weight, self.Q.bias, self.K.bias)
    sig = tuple((id(p), p._version, str(p.device)) for p in srcs)
    ent = self._bufs.get('qk')
    if ent is None or ent[1].device != self.Q.weight.device:
      ent = [None, torch.empty(2 * self.attention_dim, self.in_features, dtype=torch.float32, device=self.Q.weight.device),
             torch.empty(2 * self.attention_dim, dtype=torch.float32, device=self.Q.weight.device)]
      self._bufs['qk'] = ent
    if ent[0] != sig:
      with torch.no_grad():
        ent[1][:self.attention_dim].copy_(self.Q.weight)
        ent[1][self.attention_dim:].copy_(self.K.weight)
        ent[2][:self.attention_dim].copy_(self.Q.bias)
        ent[2][self.attention_dim:].copy_(self.K.bias)
      ent[0] = sig
    return ent[1], ent[2]

  def _split_qk_weights(self):
    """Split feature / positional kernel as one projection + one exp_kernel.  The reference's score is
        ov_x^2 exp(-|q_x - k_x|^2 / 2 l_x^2) * ov_p^2 exp(-|q_p - k_p|^2 / 2 l_p^2)
      = (ov_x ov_p)^2 exp(-|[q_x / l_x ; q_p / l_p] - [k_x / l_x ; k_p / l_p]|^2 / 2),
    so the projection matrix gets, per head, the d_k rows of Qx (scaled by 1 / l_x, acting on the feature and label
    columns of the state) followed by the d_k rows of Qp (scaled by 1 / l_p, acting on the positional columns), the
    same for K, and the kernel runs with output_var = ov_x ov_p, lengthscale = 1 and heads of width 2 d_k."""
    opt = self.opt
    srcs = (self.Qx.weight, self.Qx.bias, self.Kx.weight, self.Kx.bias, self.Qp.weight, self.Qp.bias, self.Kp.weight,
            self.Kp.bias, self.lengthscale_x, self.lengthscale_p, self.output_var_x, self.output_var_p)
    sig = tuple((id(p), p._version, str(p.device)) for p in srcs)
    dev = self.Qx.weight.device
    A, h, dk = self.attention_dim, self.h, self.d_k
    ent = self._bufs.get('qk')
    if ent is None or ent[1].device != dev:
      ent = [None, torch.zeros(4 * A, self.in_features, dtype=torch.float32, device=dev),
             torch.zeros(4 * A, dtype=torch.float32, device=dev),
             torch.ones(1, dtype=torch.float32, device=dev), torch.ones(1, dtype=torch.float32, device=dev)]
      self._bufs['qk'] = ent
    if ent[0] != sig:
      f0, p0 = opt['feat_hidden_dim'], opt['pos_enc_hidden_dim']
      lab = f0 + p0                                   # label columns (use_labels) follow the positional block
      with torch.no_grad():
        for half, (lx, lp) in enumerate(((self.Qx, self.Qp), (self.Kx, self.Kp))):
          w = ent[1][half * 2 * A:(half + 1) * 2 * A].view(h, 2 * dk, self.in_features)
          b = ent[2][half * 2 * A:(half + 1) * 2 * A].view(h, 2 * dk)
          wx = (lx.weight / self.lengthscale_x).view(h, dk, -1)
          wp = (lp.weight / self.lengthscale_p).view(h, dk, -1)
          w.zero_()
          w[:, :dk, :f0] = wx[:, :, :f0]
          w[:, :dk, lab:] = wx[:, :, f0:]
          w[:, dk:, f0:lab] = wp
          b[:, :dk] = (lx.bias / self.lengthscale_x).view(h, dk)
          b[:, dk:] = (lp.bias / self.lengthscale_p).view(h, dk)
        ent[3].copy_(self.output_var_x * self.output_var_p)
      ent[0] = sig
    return ent[1], ent[2]

  def _reweight_csr(self, graph):
    if not (self.opt['reweight_attention'] and self.edge_weights is not None):
      return None
    ew = self.edge_weights
    # one entry per graph OBJECT (the solver may run on the locality view of the graph the direct calls use; the two newest
    # graphs are kept).  The entry holds graph and weight tensor themselves (compared with `is`): ids of freed objects get reused
    ents = self._bufs.setdefault('rw', [])
    for ent in ents:
      if ent[0] is graph and ent[2] is ew and ent[3] == ew._version:
        return ent[1]
    ents[:] = [e for e in ents if e[0] is not graph][-1:]
    ents.append((graph, ops.edge_to_csr_mean(graph, ew.to(graph.device)), ew, ew._version))
    return ents[-1][1]

  def attention_struct(self, graph, q=None, k=None, ldqk=0):
    dev = graph.device
    kw = {}
    if self.split_kernel:
      self.qk_weights()                       # refreshes the derived output_var
      ent = self._bufs['qk']
      kw = dict(output_var=ent[3], lengthscale=ent[4])
    elif self.opt['attention_type'] == 'exp_kernel':
      kw = dict(output_var=ops._scalar_dev(self.output_var, graph.rowptr),
                lengthscale=ops._scalar_dev(self.lengthscale, graph.rowptr))
    transposed = None
    if self.opt['attention_norm_idx'] == 1 and self.opt['attention_type'] == 'scaled_dot' and graph.device.type == 'cuda' \
        and graph.struct.row_begin == 0 and graph.n == graph.t['rowptr'].numel() - 1:
      transposed = graph.transposed_positions()     # the column normaliser as a fused row pass over the transposed graph
    st = ops.attention_struct(_lib.ATT_TYPES[self.opt['attention_type']], self.h, self.kernel_att_dim,
                              self.opt['attention_norm_idx'], self.opt['square_plus'], q=q, k=k, ldqk=ldqk,
                              edge_w_csr=self._reweight_csr(graph), transposed=transposed, **kw)
    keep = [q, k] + list(kw.values())
    return st, keep

  def forward(self, x, edge):
    """(attention [E,h], (v, prods [E,h])) in the order of `edge` (reference :128-214)."""
    _lib.require_hip(x, edge)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._grad_sources())):
      # training: the block differentiates through this attention (interim composite, see autograd.py)
      from .autograd import layer_attention_with_grad
      att, prods = layer_attention_with_grad(self, x, edge)
      v = None if self.split_kernel else self.V(x).view(-1, self.h, self.d_k).transpose(1, 2)
      return att, (v, prods)
    with torch.no_grad():
      xc = _lib.f32c(x)
      graph = graph_of(edge, xc.shape[0], xc.device)
      wqk, bqk = self.qk_weights()
      qk = ops.linear(xc, wqk, bqk)
      A = self.kernel_att_dim
      st, keep = self.attention_struct(graph, q=qk, k=qk[:, A:], ldqk=2 * A)
      _, att, prods = ops.edge_attention(graph, st, want_w_mean=False, want_att=True, want_prods=True, like=xc)
      if self.split_kernel:
        return att, (None, prods)             # the reference returns v = None on this branch (:171)
      # V is dead on this path (mix_features crashes in the reference) but part of the
      # return value: [N, d_k, h] like the reference's transposed view
      v = ops.linear(xc, _lib.f32c(self.V.weight.detach()), _lib.f32c(self.V.bias.detach()))
      v = v.view(-1, self.h, self.d_k).transpose(1, 2)
    return att, (v, prods)

  def mean_attention(self, x, edge):
    """forward(x, edge)[0].mean(dim=1) for callers without autograd that want nothing else (the attention / hard-attention / rewiring
    blocks in evaluation mode, reference src/block_transformer_attention.py:36-40, src/block_transformer_rewiring.py:232-236): the head
    mean comes straight out of the fused row kernels (no [E,h] attention and products, no generic three-pass path), then goes from the
    CSR order to the order of `edge`."""
    _lib.require_hip(x, edge)
    with torch.no_grad():
      xc = _lib.f32c(x)
      graph = graph_of(edge, xc.shape[0], xc.device)
      wqk, bqk = self.qk_weights()
      qk = ops.linear(xc, wqk, bqk)
      A = self.kernel_att_dim
      st, keep = self.attention_struct(graph, q=qk, k=qk[:, A:], ldqk=2 * A)
      w_csr, _, _ = ops.edge_attention(graph, st, want_w_mean=True, want_att=False, want_prods=False, like=xc)
      out = torch.empty(graph.e, dtype=torch.float32, device=xc.device)
      out[graph.perm_long] = w_csr[:graph.e]
    return out

  def __repr__(self):
    return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class ODEFuncTransformerAtt(ODEFunc):

  def __init__(self, in_features, out_features, opt, data, device):
    super(ODEFuncTransformerAtt, self).__init__(opt, data, device)
    # the reference's __repr__ reads these but never sets them (print(model) raises there)
    self.in_features = in_features
    self.out_features = out_features
    if opt['self_loop_weight'] > 0:
      self.edge_index, self.edge_weight = add_remaining_self_loops(data.edge_index, data.edge_attr,
                                                                   fill_value=opt['self_loop_weight'])
    else:
      self.edge_index, self.edge_weight = data.edge_index, data.edge_attr
    self.multihead_att_layer = SpGraphTransAttentionLayer(in_features, out_features, opt, device,
                                                          edge_weights=self.edge_weight).to(device)

  def multiply_attention(self, x, attention, v=None):
    """mean-over-heads attention times x (reference :25-36; its mix_features branch raises
    AttributeError on `v.shape`, here it is reported as unsupported)."""
    if self.opt['mix_features']:
      raise NotImplementedError('mix_features is not runnable in the reference ODEFuncTransformerAtt either')
    graph = self._graph(x)
    with torch.no_grad():
      return ops.spmm(graph, ops.edge_to_csr_mean(graph, attention), _lib.f32c(x))

  def _descriptor(self, x, x0_override=None, graph=None):
    if self.opt['mix_features']:
      raise NotImplementedError('mix_features is not runnable in the reference ODEFuncTransformerAtt either')
    graph = self._graph(x) if graph is None else graph
    layer = self.multihead_att_layer
    x0 = x0_override if x0_override is not None else self._source(x)
    alpha = ops._scalar_dev(self.alpha_train, x)
    beta = ops._scalar_dev(self.beta_train, x) if x0 is not None else None
    wqk, bqk = layer.qk_weights()
    st, keep = layer.attention_struct(graph)
    desc = ops.RhsDescriptor(_lib.RHS_TRANSFORMER, graph, x.shape[1], x.stride(0), alpha, beta,
                             None if x0 is None else self._match_rows(x0, x), not self.opt['no_alpha_sigmoid'],
                             proj_w=wqk, proj_b=bqk, att=st, padded_rows=_lib.is_padded(x))
    desc.keep += keep
    return desc

  def _descriptor_signature(self, desc):
    s = desc.struct
    a = s.att
    return (id(desc.graph), s.alpha, s.beta, s.x0, s.alpha_sigmoid, s.proj_w, s.proj_b, s.proj_m, s.d, s.ld,
            a.type, a.heads, a.att_dim, a.norm_idx, a.square_plus, a.output_var, a.lengthscale, a.edge_w_csr, a.t_from_csr)

  def __repr__(self):
    return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'
