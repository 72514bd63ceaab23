"""GAT-style right-hand side (reference src/function_GAT_attention.py:11-68 ODEFuncAtt, :71-118
SpGraphAttentionLayer): score = LeakyReLU(a_src . h_i + a_dst . h_j) with h = x W split into heads,
softmax over attention_norm_idx, head-mean aggregation of x (the reference's h separate SpMMs,
stacked and averaged, are one SpMM with mean weights by linearity -- pinned by the reference's own
test_head_aggregation)."""
import torch
from torch import nn

from . import _lib, ops
from .base_classes import ODEFunc
from .graph import graph_of
from .utils import add_remaining_self_loops


class SpGraphAttentionLayer(nn.Module):

  def __init__(self, in_features, out_features, opt, device, concat=True):
    super(SpGraphAttentionLayer, self).__init__()
    self.in_features = in_features
    self.out_features = out_features
    self.alpha = opt['leaky_relu_slope']
    self.concat = concat
    self.device = device
    self.opt = opt
    self.h = opt['heads']
    self.attention_dim = opt['attention_dim'] if 'attention_dim' in opt else out_features
    assert self.attention_dim % opt['heads'] == 0, "Number of heads must be a factor of the dimension size"
    self.d_k = self.attention_dim // opt['heads']
    # proper Parameters (the reference's `nn.Parameter(...).to(device)` silently drops them from
    # model.parameters() on a non-CPU device; names and shapes are the same)
    self.W = nn.Parameter(torch.zeros(size=(in_features, self.attention_dim)))
    nn.init.xavier_normal_(self.W.data, gain=1.414)
    self.Wout = nn.Parameter(torch.zeros(size=(self.attention_dim, self.in_features)))
    nn.init.xavier_normal_(self.Wout.data, gain=1.414)
    self.a = nn.Parameter(torch.zeros(size=(2 * self.d_k, 1, 1)))
    nn.init.xavier_normal_(self.a.data, gain=1.414)
    self._bufs = {}

  def _transposed(self, name, p):
    """Row-major [out, in] copy of a [in, out] parameter for gnpde_linear, refreshed in place."""
    sig = (id(p), p._version, str(p.device))
    ent = self._bufs.get(name)
    if ent is None or ent[1].device != p.device:
      ent = [None, torch.empty(p.shape[1], p.shape[0], dtype=torch.float32, device=p.device)]
      self._bufs[name] = ent
    if ent[0] != sig:
      with torch.no_grad():
        ent[1].copy_(p.t())
      ent[0] = sig
    return ent[1]

  def proj_weight(self):
    return self._transposed('Wt', self.W)

  def out_weight(self):
    return self._transposed('Woutt', self.Wout)

  def attention_struct(self, graph, q=None, ldqk=0):
    a_flat = self.a.detach().reshape(-1)
    st = ops.attention_struct(_lib.ATT_GAT, self.h, self.attention_dim, self.opt['attention_norm_idx'], False,
                              q=q, k=q, ldqk=ldqk, leaky_slope=self.alpha, gat_a=a_flat)
    return st, [q, a_flat]

  def forward(self, x, edge):
    """(attention [E,h] in the order of `edge`, wx [N,A])  (reference :105-115)."""
    _lib.require_hip(x, edge)
    if torch.is_grad_enabled() and (x.requires_grad or self.W.requires_grad or self.a.requires_grad):
      from .autograd import native_gat_attention      # training: node-level terms in PyTorch, per-edge work native
      return native_gat_attention(self, x, edge)
    with torch.no_grad():
      xc = _lib.f32c(x)
      graph = graph_of(edge, xc.shape[0], xc.device)
      wx = ops.linear(xc, self.proj_weight())
      st, keep = self.attention_struct(graph, q=wx, ldqk=self.attention_dim)
      _, att, _ = ops.edge_attention(graph, st, want_w_mean=False, want_att=True, like=xc)
    return att, wx

  def __repr__(self):
    return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class ODEFuncAtt(ODEFunc):

  def __init__(self, in_features, out_features, opt, data, device):
    super(ODEFuncAtt, self).__init__(opt, data, device)
    self.in_features = in_features
    self.out_features = out_features
    if opt['self_loop_weight'] > 0:
      self.edge_index, self.edge_weight = add_remaining_self_loops(data.edge_index, data.edge_attr,
                                                                   fill_value=opt['self_loop_weight'])
    else:
      self.edge_index, self.edge_weight = data.edge_index, data.edge_attr
    self.multihead_att_layer = SpGraphAttentionLayer(in_features, out_features, opt, device).to(device)
    self.attention_dim = opt['attention_dim'] if 'attention_dim' in opt else out_features
    assert self.attention_dim % opt['heads'] == 0, "Number of heads must be a factor of the dimension size"
    self.d_k = self.attention_dim // opt['heads']

  def multiply_attention(self, x, attention, wx):
    """Head-mean aggregation (reference :31-43); with mix_features the aggregate of wx passes
    through Wout."""
    graph = self._graph(x)
    with torch.no_grad():
      w = ops.edge_to_csr_mean(graph, attention)
      if self.opt['mix_features']:
        return ops.linear(ops.spmm(graph, w, _lib.f32c(wx)), self.multihead_att_layer.out_weight())
      return ops.spmm(graph, w, _lib.f32c(x))

  def forward(self, t, x):
    if not self.opt['mix_features']:
      return super(ODEFuncAtt, self).forward(t, x)
    # mix_features: A(x) (xW) Wout replaces A(x) x, so the aggregation cannot carry the epilogue;
    # the elementwise tail is three tiny torch ops on [N,d] (this option is off in every best_params)
    self._check_nfe()
    if self._needs_grad(x):
      from .autograd import rhs_with_grad
      return rhs_with_grad(self, x)
    with torch.no_grad():
      attention, wx = self.multihead_att_layer(x, self.edge_index)
      ax = self.multiply_attention(x, attention, wx)
      alpha = self.alpha_train if self.opt['no_alpha_sigmoid'] else torch.sigmoid(self.alpha_train)
      f = alpha * (ax - x)
      if self.opt['add_source']:
        f = f + self.beta_train * self.x0
    return f

  def _descriptor(self, x, x0_override=None, graph=None):
    if self.opt['mix_features']:
      raise NotImplementedError('mix_features has no fused descriptor')
    graph = self._graph(x) if graph is None else graph
    layer = self.multihead_att_layer
    x0 = x0_override if x0_override is not None else self._source(x)
    alpha = ops._scalar_dev(self.alpha_train, x)
    beta = ops._scalar_dev(self.beta_train, x) if x0 is not None else None
    st, keep = layer.attention_struct(graph)
    desc = ops.RhsDescriptor(_lib.RHS_GAT, graph, x.shape[1], x.stride(0), alpha, beta,
                             None if x0 is None else self._match_rows(x0, x), not self.opt['no_alpha_sigmoid'],
                             proj_w=layer.proj_weight(), proj_b=None, att=st, padded_rows=_lib.is_padded(x))
    desc.keep += keep
    return desc

  def _descriptor_signature(self, desc):
    s = desc.struct
    a = s.att
    return (id(desc.graph), s.alpha, s.beta, s.x0, s.alpha_sigmoid, s.proj_w, s.proj_m, s.d, s.ld,
            a.heads, a.att_dim, a.norm_idx, a.gat_a, a.leaky_slope)

  def __repr__(self):
    return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'
