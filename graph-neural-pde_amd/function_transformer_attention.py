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
    if opt['beltrami'] and opt['attention_type'] == "exp_kernel":
      raise NotImplementedError('the split feature / positional exp_kernel (beltrami) is SURVEY.md 8f row 4 (next)')
    if opt['attention_type'] == "exp_kernel":
      self.output_var = nn.Parameter(torch.ones(1))
      self.lengthscale = nn.Parameter(torch.ones(1))
    self.Q = nn.Linear(in_features, self.attention_dim)
    self.V = nn.Linear(in_features, self.attention_dim)
    self.K = nn.Linear(in_features, self.attention_dim)
    self.activation = nn.Sigmoid()
    self.Wout = nn.Linear(self.d_k, in_features)
    for m in (self.Q, self.V, self.K, self.Wout):
      nn.init.constant_(m.weight, 1e-5)  # reference init (:122-126)
    self._bufs = {}

  # ---- native descriptor pieces ---------------------------------------------------------------
  def qk_weights(self):
    """[Q.weight; K.weight] ([2A, d]) and [Q.bias; K.bias], refreshed in place when a parameter's
    version changes so that captured solver graphs keep pointing at live data."""
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

  def _reweight_csr(self, graph):
    if not (self.opt['reweight_attention'] and self.edge_weights is not None):
      return None
    ew = self.edge_weights
    ent = self._bufs.get('rw')
    sig = (id(graph), id(ew), ew._version)
    if ent is None or ent[0] != sig:
      ent = (sig, ops.edge_to_csr_mean(graph, ew.to(graph.device)), ew)
      self._bufs['rw'] = ent
    return ent[1]

  def attention_struct(self, graph, q=None, k=None, ldqk=0):
    dev = graph.device
    kw = {}
    if self.opt['attention_type'] == 'exp_kernel':
      kw = dict(output_var=ops._scalar_dev(self.output_var, graph.rowptr),
                lengthscale=ops._scalar_dev(self.lengthscale, graph.rowptr))
    st = ops.attention_struct(_lib.ATT_TYPES[self.opt['attention_type']], self.h, self.attention_dim,
                              self.opt['attention_norm_idx'], self.opt['square_plus'], q=q, k=k, ldqk=ldqk,
                              edge_w_csr=self._reweight_csr(graph), **kw)
    keep = [q, k] + list(kw.values())
    return st, keep

  def forward(self, x, edge):
    """(attention [E,h], (v, prods [E,h])) in the order of `edge` (reference :128-214)."""
    _lib.require_hip(x, edge)
    if torch.is_grad_enabled() and (x.requires_grad or self.Q.weight.requires_grad or self.K.weight.requires_grad):
      # training: the block differentiates through this attention (interim composite, see autograd.py)
      from .autograd import layer_attention_with_grad
      att, prods = layer_attention_with_grad(self, x, edge)
      v = self.V(x).view(-1, self.h, self.d_k).transpose(1, 2)
      return att, (v, prods)
    with torch.no_grad():
      xc = _lib.f32c(x)
      graph = graph_of(edge, xc.shape[0], xc.device)
      wqk, bqk = self.qk_weights()
      qk = ops.linear(xc, wqk, bqk)
      A = self.attention_dim
      st, keep = self.attention_struct(graph, q=qk, k=qk[:, A:], ldqk=2 * A)
      _, att, prods = ops.edge_attention(graph, st, want_w_mean=False, want_att=True, want_prods=True, like=xc)
      # V is dead on this path (mix_features crashes in the reference) but part of the
      # return value: [N, d_k, h] like the reference's transposed view
      v = ops.linear(xc, _lib.f32c(self.V.weight.detach()), _lib.f32c(self.V.bias.detach()))
      v = v.view(-1, self.h, self.d_k).transpose(1, 2)
    return att, (v, prods)

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

  def _descriptor(self, x, x0_override=None):
    if self.opt['mix_features']:
      raise NotImplementedError('mix_features is not runnable in the reference ODEFuncTransformerAtt either')
    graph = self._graph(x)
    layer = self.multihead_att_layer
    x0 = x0_override if x0_override is not None else self._source(x)
    alpha = ops._scalar_dev(self.alpha_train, x)
    beta = ops._scalar_dev(self.beta_train, x) if x0 is not None else None
    wqk, bqk = layer.qk_weights()
    st, keep = layer.attention_struct(graph)
    desc = ops.RhsDescriptor(_lib.RHS_TRANSFORMER, graph, x.shape[1], x.stride(0), alpha, beta,
                             None if x0 is None else _lib.f32c(x0), not self.opt['no_alpha_sigmoid'],
                             proj_w=wqk, proj_b=bqk, att=st)
    desc.keep += keep
    return desc

  def _descriptor_signature(self, desc):
    s = desc.struct
    a = s.att
    return (id(desc.graph), s.alpha, s.beta, s.x0, s.alpha_sigmoid, s.proj_w, s.proj_b, s.proj_m, s.d, s.ld,
            a.type, a.heads, a.att_dim, a.norm_idx, a.square_plus, a.output_var, a.lengthscale, a.edge_w_csr)

  def __repr__(self):
    return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'
